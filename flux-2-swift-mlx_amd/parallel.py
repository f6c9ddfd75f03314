"""Batch-level data parallelism across the GPUs of one node — a CALLER of the library's RCCL entry points.

The reference has no distributed code (single process, batch 1).  Images are independent units, so the path shards at image
granularity: image i -> rank i mod world, full model replica per GPU, no collective inside the denoise loop.  Exactly two
exchanges per batch, both inside libflux2mi.so on the context's own stream (csrc/dp.cpp, RCCL over xGMI):
  * flux2mi_dp_broadcast_embeddings: the text embeddings [B,S_txt,J] bf16 from rank 0;
  * flux2mi_dp_all_gather_images:    the decoded u8 images [H,W,3].
What stays here is the control plane of one process per GPU: RANK / WORLD_SIZE / MASTER_* from the environment (the
torchrun contract), a gloo process group on 127.0.0.1 that carries the 128-byte RCCL unique id from rank 0 to the others (and the
CPU-side barrier), and the stream hand-off between torch's current stream (where the caller's tensors are produced / consumed) and
the library's stream (where generate and both collectives run, in stream order): `Context._enter()` before, `Context._exit()` after.

`backend="gloo"` WITHOUT an attached context (tests/test_parallel_gloo.py, `bench.py --dry-run`): the same call sequence on CPU
tensors through torch.distributed — a harness self-test of launcher, rendezvous and index mapping, never a result.
"""
from __future__ import annotations

import ctypes as C
import os
import time
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist

_CTX = None          # api.Context with a communicator (attach_context)
# gloo path (dry runs, CPU tests): wall-clock per exchange, same shape as flux2mi_dp_collective_stats
_GLOO_STATS = {"timing": False, "ms": [0.0, 0.0], "calls": [0, 0], "bytes": [0, 0]}


def init_distributed(backend: Optional[str] = None) -> tuple:
    """Read RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* from the env (torchrun contract) and join the gloo control group.
    `backend` is accepted for symmetry with the earlier interface: only "gloo" exists here — the data path is the library's
    communicator (attach_context), not a torch backend."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if os.environ["MASTER_ADDR"] in ("127.0.0.1", "localhost"):
            # one node, rendezvous on loopback: the container hostname may not resolve, so both bootstraps (gloo's, and RCCL's
            # out-of-band socket for the unique id exchange) are pinned to `lo`; the data path is xGMI either way
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
            os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    return rank, world, local


def attach_context(ctx) -> None:
    """Create the RCCL communicator of this rank on `ctx` (flux2mi_dp_init): rank 0 makes the unique id, the gloo group carries
    it.  With WORLD_SIZE 1 this is a one-rank communicator (the self-test of tests/test_gpu_dp.py); `bench.py` attaches only
    when world > 1."""
    global _CTX
    from ._lib import check, lib
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    ident = (C.c_uint8 * 128)()
    if rank == 0:
        check(lib().flux2mi_dp_unique_id(ident))
    if world > 1:
        box = [bytes(ident)]
        dist.broadcast_object_list(box, src=0)
        ident = (C.c_uint8 * 128).from_buffer_copy(box[0])
    check(lib().flux2mi_dp_init(ctx.handle, rank, world, ident))
    _CTX = ctx


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """image i -> rank i mod world."""
    return [i for i in range(n_items) if i % world == rank]


def _world() -> int:
    return dist.get_world_size() if dist.is_initialized() else 1


def broadcast_text_embeddings(emb: Optional[torch.Tensor], shape: Sequence[int], device, src: int = 0) -> torch.Tensor:
    """Rank `src` holds the text encoder output; everyone gets a copy (exchange 1)."""
    if _CTX is None and _world() == 1:
        return emb.to(device)
    rank = dist.get_rank() if dist.is_initialized() else 0
    if rank == src:
        buf = emb.to(device=device, dtype=torch.bfloat16).contiguous()
    else:
        buf = torch.empty(tuple(shape), dtype=torch.bfloat16, device=device)
    if _CTX is not None:
        from ._lib import check, lib
        _CTX._enter()      # the library's stream waits for the producer of `buf` on torch's current stream
        check(lib().flux2mi_dp_broadcast_embeddings(_CTX.handle, C.c_void_p(buf.data_ptr()), buf.numel(), src))
        _CTX._exit()       # torch's current stream waits for the broadcast
        return buf
    bits = buf.view(torch.uint8)  # gloo on CPU tensors: bf16 as raw bytes
    t0 = time.perf_counter()
    dist.broadcast(bits, src=src)
    _gloo_account(0, time.perf_counter() - t0, bits.numel())
    return bits.view(torch.bfloat16)


def all_gather_images(local_images: torch.Tensor) -> torch.Tensor:
    """local_images u8 [n_local,H,W,3] (same n_local on every rank) -> [world*n_local,H,W,3]
    ordered by (slot, rank) so that image i = slot*world + rank keeps its global index (exchange 2)."""
    if _CTX is None and _world() == 1:
        return local_images
    world = _world()
    local_images = local_images.contiguous()
    if _CTX is not None:
        from ._lib import check, lib
        parts = torch.empty((world,) + tuple(local_images.shape), dtype=torch.uint8, device=local_images.device)
        _CTX._enter()      # behind flux2mi_generate on the library's stream anyway; also behind the caller's own writes
        check(lib().flux2mi_dp_all_gather_images(_CTX.handle, C.c_void_p(local_images.data_ptr()), C.c_void_p(parts.data_ptr()),
                                                local_images.numel()))
        _CTX._exit()
        return parts.transpose(0, 1).reshape(-1, *local_images.shape[1:])
    parts = [torch.empty_like(local_images) for _ in range(world)]
    t0 = time.perf_counter()
    dist.all_gather(parts, local_images)
    _gloo_account(1, time.perf_counter() - t0, local_images.numel() * world)
    stacked = torch.stack(parts, dim=1)  # [n_local, world, H, W, 3]
    return stacked.reshape(-1, *local_images.shape[1:])


def _gloo_account(kind: int, seconds: float, nbytes: int) -> None:
    _GLOO_STATS["calls"][kind] += 1
    _GLOO_STATS["bytes"][kind] += int(nbytes)
    if _GLOO_STATS["timing"]:
        _GLOO_STATS["ms"][kind] += seconds * 1e3


def collective_timing(on: bool) -> None:
    """Bracket every exchange with HIP events on the library's stream (flux2mi_dp_collective_timing); on the gloo path: wall clock."""
    if _CTX is not None:
        from ._lib import check, lib
        check(lib().flux2mi_dp_collective_timing(_CTX.handle, 1 if on else 0))
    _GLOO_STATS["timing"] = bool(on)


def collective_stats(reset: bool = False) -> dict:
    """{"broadcast": {"ms", "calls", "bytes"}, "all_gather": {...}} since the last reset (flux2mi_dp_collective_stats; synchronises)."""
    if _CTX is not None:
        from ._lib import check, lib
        ms, calls, nbytes = (C.c_double * 2)(), (C.c_int64 * 2)(), (C.c_int64 * 2)()
        check(lib().flux2mi_dp_collective_stats(_CTX.handle, ms, calls, nbytes, 1 if reset else 0))
        vals = [list(ms), list(calls), list(nbytes)]
    else:
        vals = [list(_GLOO_STATS["ms"]), list(_GLOO_STATS["calls"]), list(_GLOO_STATS["bytes"])]
        if reset:
            _GLOO_STATS.update(ms=[0.0, 0.0], calls=[0, 0], bytes=[0, 0])
    return {name: {"ms": float(vals[0][k]), "calls": int(vals[1][k]), "bytes": int(vals[2][k])}
            for k, name in enumerate(("broadcast", "all_gather"))}


# ---- analytic scaling model (SURVEY.md §8e "Measurability") -------------------------------------------------------
# One node, image-granular replicas: the only terms that grow with the world size are the two exchanges and rank 0's D2H of the
# gathered batch.  xGMI is point-to-point (7 links per GPU, ~153 GB/s each; PCIe Gen5 x16 63 GB/s spec: /opt/skills/guides/MI355X_MICROARCH.md), RCCL's
# ring forms are bound by ONE link: a broadcast of B bytes streams B through every hop of the chain (pipelined: ~B / link + hops x
# latency); a ring all-gather moves (N - 1) / N of the gathered bytes through each link.
XGMI_LINK_GBS = 153.0          # per link, per direction (data sheet)
XGMI_EFFICIENCY = 0.70         # what a ring protocol sustains of that (assumed; no multi-GPU box was available to measure it)
HOP_LATENCY_US = 12.0          # per ring step, launch + protocol (assumed)
PCIE_D2H_GBS = 50.0            # rank 0's D2H of the gathered images (pinned host memory, PCIe 5 x16; assumed)


def scaling_model(seconds_per_image_1gpu: float, denoise_steps: int, s_txt: int, joint_dim: int, height: int, width: int,
                  worlds=(1, 2, 4, 8)) -> dict:
    """Predicted seconds/image, denoise-steps/s and weak-scaling efficiency at N GPUs from the measured 1-GPU time of one image
    (which already holds one image's D2H) plus the modelled per-batch terms.  One batch = N images, one per GPU."""
    link = XGMI_LINK_GBS * XGMI_EFFICIENCY * 1e9
    img = height * width * 3
    rows = []
    for n in worlds:
        b_bytes = n * s_txt * joint_dim * 2                      # bf16 embeddings of the whole batch, from rank 0
        g_bytes = n * img                                        # u8 images, every rank ends with all of them
        if n == 1:
            t_b = t_g = t_d = 0.0
        else:
            t_b = b_bytes / link + (n - 1) * HOP_LATENCY_US * 1e-6
            t_g = (n - 1) * img / link + (n - 1) * HOP_LATENCY_US * 1e-6
            t_d = (n - 1) * img / (PCIE_D2H_GBS * 1e9)           # the first image's D2H is inside the 1-GPU measurement
        t = seconds_per_image_1gpu + t_b + t_g + t_d
        rows.append({"n_gpus": n, "broadcast_bytes": b_bytes, "all_gather_bytes": g_bytes,
                     "broadcast_ms": round(t_b * 1e3, 4), "all_gather_ms": round(t_g * 1e3, 4), "extra_d2h_ms": round(t_d * 1e3, 4),
                     "seconds_per_image": round(t, 5), "denoise_steps_per_s": round(n * denoise_steps / t, 3),
                     "efficiency": round(seconds_per_image_1gpu / t, 5)})
    return {"model": "t(N) = t(1) + broadcast(N x s_txt x J x 2 B, chain over one xGMI link) + ring all-gather((N-1) x H x W x 3 B per "
                     "link) + rank 0's D2H of the other N-1 images; no term inside the denoise loop (image-granular replicas)",
            "assumptions": {"xgmi_link_GBps": XGMI_LINK_GBS, "ring_efficiency": XGMI_EFFICIENCY, "hop_latency_us": HOP_LATENCY_US,
                            "pcie_d2h_GBps": PCIE_D2H_GBS,
                            "not_modelled": "per-GPU clock differences inside a node (boxes of the pool differ by up to 11 % on one "
                                            "binary; the batch takes the slowest rank's time) and host-side launch jitter"},
            "base_seconds_per_image": round(seconds_per_image_1gpu, 5), "predicted": rows}


def finalize():
    """Tear communicator and control group down (ranks leave together; no-op for a single process)."""
    global _CTX
    if _CTX is not None:
        from ._lib import check, lib
        check(lib().flux2mi_dp_finalize(_CTX.handle))
        _CTX = None
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def all_reduce_max(value: float, device=None) -> float:
    if _CTX is not None:
        from ._lib import check, lib
        v = C.c_double(value)
        check(lib().flux2mi_dp_all_reduce_max(_CTX.handle, C.byref(v)))
        return float(v.value)
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
