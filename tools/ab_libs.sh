#!/bin/bash
# Same-box A/B of two builds of the library (FLUX2MI_LIB), interleaved: tools/ab_libs.sh OUT LIB_A LIB_B [bench.py args...]
# e.g. tools/ab_libs.sh gpurun_out/ab libflux2mi_r04.so libflux2mi.so --steps 4 --warmup 1   (round-4 build against this round's)
set -o pipefail
O=$1; A=$2; B=$3; shift 3
mkdir -p $O
for rep in 1 2; do
  for L in $A $B; do
    FLUX2MI_LIB=$L python bench.py --no-cpu-baseline --no-secondary "$@" > $O/run_${L%.so}_$rep.json.log 2>>$O/err.txt || exit 1
    python3 - "$O/run_${L%.so}_$rep.json.log" "$L" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], {k: d.get(k) for k in ["value", "seconds_per_image", "denoise_ms_per_denoise_step", "decode_ms_per_image"]},
      "mfma_stream", d["roofline"].get("mfma_stream"), "linear1 ms", d["roofline"]["avg_launch_ms"], "crc", d["output_check"]["image_crc32"], flush=True)
PY
  done
done
