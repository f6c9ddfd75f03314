"""What the GPU box's host gives the CPU oracle: cores, cgroup quota, memory, and f32 matmul rate by thread count."""
import os
import time

import torch

print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads())
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/memory.max"):
    try:
        print(p, open(p).read().strip())
    except OSError as e:
        print(p, e)
print(open("/proc/meminfo").read().split("\n")[0])
a = torch.randn(4608, 3072)
b = torch.randn(9216, 3072)
for n in (8, 16, 32, 64, 128):
    if n > (os.cpu_count() or 1):
        break
    torch.set_num_threads(n)
    torch.nn.functional.linear(a, b)
    t0 = time.perf_counter()
    for _ in range(3):
        torch.nn.functional.linear(a, b)
    dt = (time.perf_counter() - t0) / 3
    print(f"threads {n}: {2 * 4608 * 3072 * 9216 / dt / 1e12:.2f} TFLOP/s f32")
