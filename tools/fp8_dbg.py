import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flux2mi
from flux2mi import synthetic as syn
from flux2mi.api import Context
from tests.test_gpu_dit import _fp8_case
from tests.util import rel_l2
ctx = Context(0)
for std in (0.05, 0.08):
    outs, ref32, ref8 = _fp8_case(ctx, syn.DIT_PRESETS["tiny"], 19, 80, 112, std=std)
    d_hip = (outs["fp8"] - outs["bf16"]).flatten(); d_ref = (ref8 - ref32).flatten()
    print(std, "bf16-vs-32", rel_l2(outs["bf16"], ref32), "fp8-vs-ref8", rel_l2(outs["fp8"], ref8), "fp8-vs-32", rel_l2(outs["fp8"], ref32),
          "ref8-vs-32", rel_l2(ref8, ref32), "cos", float(torch.dot(d_hip, d_ref) / (d_hip.norm() * d_ref.norm())))
