"""Batch-level data parallelism across the GPUs of one node.

The reference has no distributed code (single process, batch 1).  Images are
independent units, so the path shards at image granularity: image i -> rank
i mod world, full model replica per GPU, no collective inside the denoise loop.
Exactly two exchanges per batch, over RCCL/xGMI on GPUs (backend "nccl") or gloo
on CPU (tests):
  * broadcast of the text embeddings [B,S_txt,J] bf16 from rank 0;
  * all-gather of the decoded u8 images [H,W,3].
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist


def init_distributed(backend: Optional[str] = None) -> tuple:
    """Read RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* from the env (torchrun contract)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """image i -> rank i mod world."""
    return [i for i in range(n_items) if i % world == rank]


def broadcast_text_embeddings(emb: Optional[torch.Tensor], shape: Sequence[int], device, src: int = 0) -> torch.Tensor:
    """Rank `src` holds the text encoder output; everyone gets a copy.  bf16 is moved as
    raw bytes so that gloo (CPU tests) and RCCL take the same path."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return emb.to(device)
    if dist.get_rank() == src:
        buf = emb.to(device=device, dtype=torch.bfloat16).contiguous()
    else:
        buf = torch.empty(tuple(shape), dtype=torch.bfloat16, device=device)
    bits = buf.view(torch.uint8)  # raw bytes: gloo (CPU tests) and RCCL take the same path
    dist.broadcast(bits, src=src)
    return bits.view(torch.bfloat16)


def all_gather_images(local_images: torch.Tensor) -> torch.Tensor:
    """local_images u8 [n_local,H,W,3] (same n_local on every rank) -> [world*n_local,H,W,3]
    ordered by (slot, rank) so that image i = slot*world + rank keeps its global index."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local_images
    world = dist.get_world_size()
    parts = [torch.empty_like(local_images) for _ in range(world)]
    dist.all_gather(parts, local_images.contiguous())
    stacked = torch.stack(parts, dim=1)  # [n_local, world, H, W, 3]
    return stacked.reshape(-1, *local_images.shape[1:])


def finalize():
    """Tear the process group down (ranks leave together; no-op for a single process)."""
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def all_reduce_max(value: float, device) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
