#!/bin/bash
O=gpurun_out/r06d; mkdir -p $O
run() { local t=$1; shift; timeout -k 10 "$t" "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILLED: $*"; exit 1; fi; return 0; }
run 200 python tools/fwd_crc.py > $O/fwd_crc.txt 2>&1
run 600 python -m pytest tests/test_gpu_ops.py -x -q -k "gemm" > $O/gemm_tests.log 2>&1
run 300 python bench.py --config A --steps 10 --warmup 2 --no-cpu-baseline --no-secondary >> $O/small_s.jsonl 2>>$O/bench.err
run 300 python bench.py --height 512 --width 512 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary >> $O/small_s.jsonl 2>>$O/bench.err
run 900 python -m pytest tests/test_gpu_dit.py tests/test_gpu_pipeline.py tests/test_gpu_golden.py tests/test_gpu_textenc.py "tests/test_gpu_configs.py::test_config_a_klein4b_256_4steps" "tests/test_gpu_configs.py::test_klein4b_bf16_teacher_forced_blocks_vs_store_point_oracle" -x -q > $O/model_tests.log 2>&1
grep -v amdgpu $O/fwd_crc.txt; tail -3 $O/gemm_tests.log $O/model_tests.log; python3 -c "
import json
for l in open('$O/small_s.jsonl'):
    d=json.loads(l); print(d['config']['workload'][:40], d['seconds_per_image'], d['denoise_ms_per_denoise_step'], d['dit_step_mfma_frac'])"
