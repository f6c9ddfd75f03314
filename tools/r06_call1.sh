#!/bin/bash
# round 6, GPU call 1: the streaming VAE attention (op bench, parity), decode timings, the resolution tests (writes the new oracle fixture)
O=gpurun_out/r06a; mkdir -p $O
run() { local t=$1; shift; timeout -k 10 "$t" "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILLED: $*"; exit 1; fi; return 0; }
run 200 python tools/vae_attn_bench.py > $O/vae_attn_bench.txt 2>&1
run 400 python -m pytest tests/test_gpu_resolutions.py -x -q -s -k "vae_attention" > $O/attn_tests.log 2>&1
for v in standard small; do run 200 python tools/vae_bench.py $v 10 >> $O/vae_bench.txt 2>&1; done
run 200 python tools/vae_bench.py standard 3 128 192 >> $O/vae_bench.txt 2>&1
run 200 python tools/vae_bench.py standard 3 256 256 >> $O/vae_bench.txt 2>&1
run 200 python tools/vae_bench.py small 3 256 256 >> $O/vae_bench.txt 2>&1
FLUX2MI_WRITE_ORACLE_CACHE=$O/cache run 900 python -m pytest tests/test_gpu_resolutions.py -q -s -k "not vae_attention" --durations=0 > $O/resolution_tests.log 2>&1
run 600 python -m pytest tests/test_gpu_vae.py tests/test_gpu_dp.py "tests/test_gpu_ops.py::test_gemm_gate_resid_beyond_32bit_offsets" "tests/test_gpu_formats.py::test_trim_from_another_thread_while_calls_run" "tests/test_gpu_configs.py::test_vae_standard_large_latents" -q -s > $O/regression_tests.log 2>&1
tail -3 $O/*.log; cat $O/vae_attn_bench.txt $O/vae_bench.txt
