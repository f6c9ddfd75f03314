#!/usr/bin/env python3
"""Power / clock / time of the 256x256 GEMM at 8192^3 (and linear1's shape) under ONE timing-only ablation of the dev
library (FLUX2MI_LIB=libflux2mi_dev.so FLUX2MI_GEMM_ABLATE=n): which part of the kernel the 1400 W go to.  Dev tool."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch

from power_probe import run
import flux2mi
from flux2mi.api import Context

ctx = Context(0)
tag = f"ABL={os.environ.get('FLUX2MI_GEMM_ABLATE', '0')} V={os.environ.get('FLUX2MI_GEMM_V', '2')}"
a = torch.randn((8192, 8192), device="cuda").to(torch.bfloat16)
w = (torch.randn((8192, 8192), device="cuda") * 0.05).to(torch.bfloat16)
shape = os.environ.get("SHAPE")
if shape:
    M, N, K = [int(x) for x in shape.split("x")]
    a = torch.randn((M, K), device="cuda").to(torch.bfloat16)
    w = (torch.randn((N, K), device="cuda") * 0.05).to(torch.bfloat16)
    tag += " " + shape
if "--blaslt" in sys.argv:
    run("hipBLASLt 8192^3", lambda: torch.matmul(a, w.t()), 4.0)
else:
    run(f"gemm 8192^3 {tag}", lambda: ctx.op_gemm_bf16(a, w), 4.0)
