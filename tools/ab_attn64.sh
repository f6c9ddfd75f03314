#!/bin/bash
# same-box A/B of the model step: attention64b.hip (default, MFMA 16x16x32) vs attention64.hip (FLUX2MI_ATTN64=1, 32x32x16) vs the
# 8-wave attention kernel (FLUX2MI_ATTN64=0); dev library
export FLUX2MI_LIB=libflux2mi_dev.so
for r in 1 2; do
  for v in 2 1 0; do
    export FLUX2MI_ATTN64=$v
    python bench.py --steps 3 --warmup 1 --no-cpu-baseline $1 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('ATTN64=$v', 'ms/step', d['denoise_ms_per_denoise_step'], 's/image', d['seconds_per_image'], 'linear1 TF', d['roofline']['achieved'], 'crc', d['output_check']['image_crc32'], flush=True)"
  done
done
