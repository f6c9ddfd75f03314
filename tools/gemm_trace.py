#!/usr/bin/env python3
"""Per-workgroup timeline of one warm single_blocks.linear1 launch of the production 256x256 GEMM
(FLUX2MI_GEMM_TRACE stamps: entry, first K-tile ready, K loop done, stores issued, stores retired).
Answers: how long is the prologue / epilogue of a tile, and how long does a CU sit between two tiles.  Dev tool."""
import os, sys, subprocess, collections
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.path.join(ROOT, "gpurun_out", "gemm_trace.bin")
if "--analyse" not in sys.argv:
    os.makedirs(os.path.dirname(path), exist_ok=True)
    env = dict(os.environ, FLUX2MI_GEMM_TRACE=path)
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "one_forward.py"), "klein-4b", "3"], check=True, env=env)
a = np.fromfile(path, dtype=np.uint64).reshape(-1, 8, 8).astype(np.int64)   # [wg][wave][field]
nwg = a.shape[0]
t0, t1, t2, t3, t4 = (a[:, :, i] for i in range(5))
base = t0.min()
print(f"workgroups {nwg}; launch span {(t4.max() - base)} cycles")
def stat(name, x):
    print(f"  {name:34s} mean {x.mean():9.0f}  p10 {np.percentile(x, 10):9.0f}  p50 {np.percentile(x, 50):9.0f}  p90 {np.percentile(x, 90):9.0f}")
stat("prologue (entry -> first K-tile)", (t1 - t0).reshape(-1))
stat("K loop", (t2 - t1).reshape(-1))
stat("epilogue issue (loop end -> issued)", (t3 - t2).reshape(-1))
stat("store drain (issued -> vmcnt 0)", (t4 - t3).reshape(-1))
wg_start, wg_end = t0.min(axis=1), t4.max(axis=1)
stat("workgroup total", wg_end - wg_start)
hw = a[:, 0, 5]
key = ((hw >> 32) << 16) | ((hw & 0xffffffff) >> 8 & 0xff)
by = collections.defaultdict(list)
for i in range(nwg):
    by[int(key[i])].append((int(wg_start[i]), int(wg_end[i]), int(a[i, 0, 7])))
gaps, busy = [], []
for k, lst in by.items():
    lst.sort()
    for (s0, e0, _), (s1, e1, _) in zip(lst, lst[1:]):
        gaps.append(s1 - e0)
    busy.append(sum(e - s for s, e, _ in lst))
print(f"CUs seen {len(by)}; workgroups per CU {min(len(v) for v in by.values())}..{max(len(v) for v in by.values())}")
stat("gap: WG end -> next WG entry (same CU)", np.array(gaps))
# same-round lockstep: spread of the K-loop end times of the first round
first = np.sort(t2.max(axis=1))[: min(256, nwg)]
print(f"  spread of the first 256 loop-end times: p10-p90 {np.percentile(first, 90) - np.percentile(first, 10):.0f} cycles")
epi = a[:, 0, 7]
for e in np.unique(epi):
    m = epi == e
    print(f"  epi {e}: tiles {m.sum()}, epilogue issue mean {(t3 - t2)[m].mean():.0f}, drain mean {(t4 - t3)[m].mean():.0f}")
