#!/bin/bash
# A/B of the bf16 256x256 GEMM with the builtin LDS-DMA (VAR=11) vs the scalar-base asm form (VAR=3, the default), dev library, NI forced to 8
export FLUX2MI_LIB=libflux2mi_dev.so FLUX2MI_GEMM=256 FLUX2MI_GEMM_NI=8
for r in 1 2; do for v in 3 19; do echo "== VAR=$v"; FLUX2MI_GEMM_VAR=$v FLUX2MI_GEMM_V=2 python tools/gemm_g4_check.py --variant v2-8wave $1 2>&1 | grep -E "check.*(exact|K3072)|v2-8wave"; done; done
