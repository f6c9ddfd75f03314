#!/usr/bin/env python3
"""Which hipBLASLt (Tensile) kernels torch.matmul picks for the DiT's long-K N = d GEMMs: run under
`rocprofv3 --kernel-trace --stats` and read the kernel names (macro tile, stream-K, ...).  Dev tool."""
import sys, torch
shapes = [(4608, 3072, 12288), (4608, 3072, 9216), (4608, 3072, 3072), (4608, 27648, 3072), (4608, 4096, 16384)]
for M, N, K in shapes:
    a = torch.randn((M, K), device="cuda").to(torch.bfloat16)
    w = (torch.randn((N, K), device="cuda") * 0.05).to(torch.bfloat16)
    for _ in range(5):
        c = torch.matmul(a, w.t())
    torch.cuda.synchronize()
print("ok")
