#!/bin/bash
export FLUX2MI_LIB=libflux2mi_dev.so FLUX2MI_GEMM=256 FLUX2MI_GEMM_V=6 FLUX2MI_GEMM_TRACE=1
for a in 0 100 104; do
  echo "== ABL=$a (0 register staging, 100 LDS-DMA, 104 LDS-DMA without the DMA)"
  FLUX2MI_GEMM_ABLATE=$a python tools/gemm_g4_check.py --variant v6-g4 $1 2>&1 | grep -E "check|trace.*K (12288|8192)|v6-g4" | awk '!seen[$0]++' | head -40
done
FLUX2MI_GEMM_V=2 python tools/gemm_g4_check.py --variant v2-8wave 2>&1 | grep v2
