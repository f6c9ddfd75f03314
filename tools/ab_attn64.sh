#!/bin/bash
# same-box A/B of the model step: attention64.hip (default) vs the 8-wave attention kernel (dev library, FLUX2MI_ATTN64=0)
export FLUX2MI_LIB=libflux2mi_dev.so
for r in 1 2; do
  for v in new old; do
    if [ $v = old ]; then export FLUX2MI_ATTN64=0; else unset FLUX2MI_ATTN64; fi
    python bench.py --steps 3 --warmup 1 --no-cpu-baseline $1 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$v', 'ms/step', d['denoise_ms_per_denoise_step'], 's/image', d['seconds_per_image'], 'linear1 TF', d['roofline']['achieved'], 'crc', d['output_check']['image_crc32'], flush=True)"
  done
done
