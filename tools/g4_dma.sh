#!/bin/bash
# 4-wave experiment (dev library): register staging vs global_load_lds vs buffer_load..lds, then the 8-wave kernel
export FLUX2MI_LIB=libflux2mi_dev.so FLUX2MI_GEMM=256 FLUX2MI_GEMM_V=6 FLUX2MI_GEMM_TRACE=1
for a in ${G4_ABLS:-0 100 200}; do
  echo "== ABL=$a (0 register staging, 100 global_load_lds, 200 buffer_load lds; +4 = without the DMA)"
  FLUX2MI_GEMM_ABLATE=$a timeout -k 10 300 python tools/gemm_g4_check.py --variant v6-g4 $1 $2 2>&1 | grep -E "check|trace.*K (12288|8192)|v6-g4|Error|error" | awk '!seen[$0]++' | head -40
done
FLUX2MI_GEMM_V=2 python tools/gemm_g4_check.py --variant v2-8wave $2 2>&1 | grep v2
python tools/gemm_g4_check.py --variant blaslt $2 2>&1 | grep blaslt
