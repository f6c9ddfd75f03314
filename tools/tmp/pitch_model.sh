set -o pipefail
O=gpurun_out/pm; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_formats.py tests/test_gpu_dit.py tests/test_gpu_configs.py tests/test_gpu_pipeline.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bf16.log 2>&1 && tail -1 $O/bf16.log | cut -c1-200
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --weights fp8 > $O/fp8.log 2>&1 && tail -1 $O/fp8.log | cut -c1-200
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --weights fp8 --model klein-9b > $O/fp8_9b.log 2>&1 && tail -1 $O/fp8_9b.log | cut -c1-200
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --model klein-9b > $O/bf16_9b.log 2>&1 && tail -1 $O/bf16_9b.log | cut -c1-200
