#!/bin/bash
O=gpurun_out/r06b; mkdir -p $O
run() { local t=$1; shift; timeout -k 10 "$t" "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILLED: $*"; exit 1; fi; return 0; }
run 200 python tools/vae_attn_bench.py > $O/vae_attn_bench.txt 2>&1
run 400 python -m pytest tests/test_gpu_resolutions.py -x -q -s -k "vae_attention" > $O/attn_tests.log 2>&1
for v in standard small; do run 200 python tools/vae_bench.py $v 10 >> $O/vae_bench.txt 2>&1; done
run 200 python tools/vae_bench.py standard 3 256 256 >> $O/vae_bench.txt 2>&1
tail -3 $O/attn_tests.log; grep -v amdgpu.ids $O/vae_attn_bench.txt $O/vae_bench.txt
