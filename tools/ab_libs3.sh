#!/bin/bash
# Same-box A/B of several builds of the library, interleaved, 3 rounds: tools/ab_libs3.sh OUT "LIB_A LIB_B ..." [bench.py args...]
set -o pipefail
O=$1; LIBS=$2; shift 2
mkdir -p $O
for rep in 1 2 3; do
  for L in $LIBS; do
    FLUX2MI_LIB=$L python bench.py --no-cpu-baseline --no-secondary "$@" > $O/run_${L%.so}_$rep.json.log 2>>$O/err.txt || exit 1
    python3 - "$O/run_${L%.so}_$rep.json.log" "$L" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], {k: d.get(k) for k in ["value", "denoise_ms_per_denoise_step", "decode_ms_per_image"]},
      "mfma_stream", d["roofline"].get("mfma_stream"), "linear1 ms", d["roofline"]["avg_launch_ms"], "crc", d["output_check"]["image_crc32"], flush=True)
PY
  done
done
