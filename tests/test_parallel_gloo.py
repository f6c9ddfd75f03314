"""CPU: the N>1 data-parallel path with world_size 2 over gloo (the GPU run uses the same code
over RCCL): sharding, text-embedding broadcast, image all-gather, max-over-ranks timing."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import flux2mi  # noqa: F401
    from flux2mi import parallel as par

    r, w, _ = par.init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    shape = (world, 16, 384)
    emb = None
    if rank == 0:
        emb = torch.randn(shape, generator=torch.Generator().manual_seed(7)).to(torch.bfloat16)
    got = par.broadcast_text_embeddings(emb, shape, "cpu")
    ref = torch.randn(shape, generator=torch.Generator().manual_seed(7)).to(torch.bfloat16)
    assert torch.equal(got.view(torch.int16), ref.view(torch.int16))
    # every rank "decodes" its image(s): image i = slot*world + rank
    n_local = 2
    local = torch.stack([torch.full((4, 6, 3), (s * world + rank), dtype=torch.uint8) for s in range(n_local)])
    allimgs = par.all_gather_images(local)
    assert tuple(allimgs.shape) == (world * n_local, 4, 6, 3)
    assert [int(allimgs[i, 0, 0, 0]) for i in range(world * n_local)] == list(range(world * n_local))
    assert par.shard_indices(5, rank, world) == [i for i in range(5) if i % world == rank]
    par.barrier()
    assert par.all_reduce_max(1.0 + rank, "cpu") == float(world)
    q.put(rank)


def test_dp_world_size_2_gloo():
    world = 2
    port = 29500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert sorted(q.get(timeout=5) for _ in range(world)) == list(range(world))


def _run_bench(args, env_extra=None, timeout=300):
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def test_bench_launches_its_own_ranks_dry_run():
    """`python bench.py --gpus 2` with no WORLD_SIZE starts two fresh ranks itself (the parent touches no GPU), which
    rendezvous on 127.0.0.1, run both collectives and print ONE JSON line from rank 0 with n_gpus = 2.  --dry-run:
    gloo on CPU tensors, the hot path replaced by a placeholder (rank r's image is filled with r + 1)."""
    import zlib
    r, out = _run_bench(["--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1", "--height", "64", "--width", "64"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert sum(1 for l in r.stdout.splitlines() if l.startswith("{")) == 1
    assert out["n_gpus"] == 2 and out["dry_run"] is True and out["value"] is None and out["steps"] == 2
    assert out["config"]["images_per_bench_step"] == 2 and out["config"]["parallelism"] == "dp2"
    expect = bytes([1]) * (64 * 64 * 3) + bytes([2]) * (64 * 64 * 3)      # all-gathered batch in rank order
    assert out["output_check"]["image_crc32"] == f"{zlib.crc32(expect) & 0xffffffff:08x}"
    # SURVEY §8e "Measurability": per-exchange timings of the timed region (here: wall clock around the gloo collectives) with the
    # payload of ONE batch, and the analytic 1 -> 8 model next to them
    cm = out["collective_ms"]
    assert cm["broadcast"] > 0 and cm["all_gather"] > 0
    assert cm["broadcast_bytes"] == 2 * 512 * 64 * 2 and cm["all_gather_bytes"] == 2 * 64 * 64 * 3
    sm = out["scaling_model"]
    assert [r["n_gpus"] for r in sm["predicted"]] == [1, 2, 4, 8]
    assert sm["predicted"][1]["broadcast_bytes"] == cm["broadcast_bytes"] and sm["predicted"][1]["all_gather_bytes"] == cm["all_gather_bytes"]


def test_scaling_model_terms():
    """The model of SURVEY §8e / DESIGN §1e: per-batch bytes of the two exchanges, efficiency 1 at N = 1 and falling with N,
    aggregate throughput growing with N; a longer image time dilutes the exchanges."""
    sys.path.insert(0, ROOT)
    import flux2mi  # noqa: F401
    from flux2mi import parallel as par
    m = par.scaling_model(0.75, 28, 512, 7680, 1024, 1024)
    rows = m["predicted"]
    assert rows[0]["efficiency"] == 1.0 and rows[0]["broadcast_ms"] == 0.0
    for a, b in zip(rows, rows[1:]):
        assert b["efficiency"] < a["efficiency"] and b["denoise_steps_per_s"] > a["denoise_steps_per_s"]
    r8 = rows[-1]
    assert r8["broadcast_bytes"] == 8 * 512 * 7680 * 2 and r8["all_gather_bytes"] == 8 * 1024 * 1024 * 3
    link = par.XGMI_LINK_GBS * par.XGMI_EFFICIENCY * 1e9
    assert abs(r8["broadcast_ms"] - (r8["broadcast_bytes"] / link + 7 * par.HOP_LATENCY_US * 1e-6) * 1e3) < 1e-3
    assert abs(r8["all_gather_ms"] - (7 * 1024 * 1024 * 3 / link + 7 * par.HOP_LATENCY_US * 1e-6) * 1e3) < 1e-3
    assert r8["efficiency"] > 0.99                       # two exchanges per IMAGE, not per step: the path is compute-bound at any N
    slow = par.scaling_model(7.5, 50, 512, 15360, 1024, 1024)["predicted"][-1]
    assert slow["efficiency"] > r8["efficiency"]


def test_bench_refuses_world_size_mismatch():
    """--gpus 8 inside a 1-rank environment (or any WORLD_SIZE != --gpus) must fail instead of reporting n_gpus = 1."""
    r, out = _run_bench(["--gpus", "8", "--dry-run"], env_extra={"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and out is None
    assert "WORLD_SIZE" in r.stderr
    r, out = _run_bench(["--gpus", "1", "--dry-run"], env_extra={"WORLD_SIZE": "2", "RANK": "0"})
    assert r.returncode != 0 and out is None


@pytest.mark.parametrize("config, gpus, expect", [
    ("D", 2, {"model": "klein-9b", "dtype_starts": "fp8", "steps_eff": 28, "hook": False}),
    ("E", 2, {"model": "dev", "dtype_starts": "bf16", "steps_eff": 35, "hook": True}),
    ("C", 1, {"model": "dev", "dtype_starts": "bf16", "steps_eff": 50, "hook": False}),
])
def test_bench_config_entry_points_dry_run(config, gpus, expect):
    """`bench.py --gpus N --config {C,D,E}` (VERDICT r3 #6b): the multi-GPU entry points of BASELINE.json configs[2..4] emit the same
    JSON contract, `config.workload` names what runs and `config.baseline_config` the BASELINE entry; E's effective step count comes
    from the host scheduler (50 steps at strength 0.7 -> 35).  --dry-run: gloo, no compute."""
    r, out = _run_bench(["--gpus", str(gpus), "--dry-run", "--config", config, "--steps", "1", "--warmup", "0", "--height", "64", "--width", "64"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert out["n_gpus"] == gpus and out["scaling"] == "weak" and out["unit"] == "denoise-steps/s"
    c = out["config"]
    assert expect["model"] in c["workload"] and c["baseline_config"].startswith(f"{config} = BASELINE.json configs[")
    assert c["effective_denoise_steps"] == expect["steps_eff"] and c["parallelism"] == f"dp{gpus}"
    assert ("RePaint step hook" in c["workload"]) == expect["hook"]
    assert out["dtype"].startswith(expect["dtype_starts"])
    assert c["matches_baseline_config"] is False          # 64x64 here: the line says so itself
    for k in ("roofline", "roofline_secondary", "roofline_hbm", "cpu_baseline"):
        assert k in out
