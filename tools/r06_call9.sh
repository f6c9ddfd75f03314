#!/bin/bash
O=gpurun_out/r06j; mkdir -p $O
run() { local t=$1; shift; timeout -k 10 "$t" "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILLED: $*"; exit 1; fi; return 0; }
run 300 python -m pytest tests/test_gpu_vae.py -x -q -s > $O/vae_tests.log 2>&1
for v in small standard; do run 200 python tools/vae_bench.py $v 10 >> $O/vae_bench.txt 2>&1; done
run 200 python tools/vae_bench.py small 3 256 256 >> $O/vae_bench.txt 2>&1
run 400 python -m pytest tests/test_gpu_resolutions.py -q -s -k "vae_decode" > $O/res_tests.log 2>&1
tail -n 4 $O/vae_tests.log $O/res_tests.log; grep -v amdgpu $O/vae_bench.txt
