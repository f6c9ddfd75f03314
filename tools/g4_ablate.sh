#!/bin/bash
# cycles per K-tile of the 4-wave GEMM's loop under timing-only ablations (dev library)
export FLUX2MI_LIB=libflux2mi_dev.so FLUX2MI_GEMM=256 FLUX2MI_GEMM_V=6 FLUX2MI_GEMM_TRACE=1
for a in 0 7 6 4 2 1 3; do
  echo "== ABL=$a (1 no reads, 2 no writes, 4 no loads)"
  FLUX2MI_GEMM_ABLATE=$a python tools/gemm_g4_check.py --variant v6-g4 2>&1 | grep -E "trace.*K (12288|8192)|linear2|square8k" | awk '!seen[$0]++' | head -6
done
