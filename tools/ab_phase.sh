#!/bin/bash
# Same-box A/B of two product libraries, libflux2mi_old.so (a copy of an earlier build) and libflux2mi.so: attention and GEMM
# microbenchmarks, two interleaved rounds.  Used for the code-placement question (profiles/r02_q_code_placement_ab.log): the
# steady loops of attn64b_kernel / gemm256s_kernel<8> / <7> at 4 vs 0 bytes modulo 8 — no difference beyond noise.
for r in 1 2; do
  for lib in libflux2mi_old.so libflux2mi.so; do
    echo "== $lib round $r"
    FLUX2MI_LIB=$lib python tools/attn_offset_bench.py 2>/dev/null | head -3 | cut -c1-160
    FLUX2MI_LIB=$lib python tools/gemm_bench.py 2>/dev/null | grep -v "^FLUX\|amdgpu" | cut -c1-110
  done
done
