set -e
mkdir -p gpurun_out/r01_g
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r01_g/gpu_tests.log 2>&1 || { tail -40 gpurun_out/r01_g/gpu_tests.log; exit 1; }
tail -1 gpurun_out/r01_g/gpu_tests.log
timeout -k 10 300 python bench.py --steps 3 --warmup 1 > gpurun_out/r01_g/bench.json.log 2>gpurun_out/r01_g/bench.err
timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --weights fp8 > gpurun_out/r01_g/bench_fp8.json.log 2>>gpurun_out/r01_g/bench.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01_g/prof -o b -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r01_g/prof.log 2>&1
timeout -k 10 900 python tools/run_configs.py A D D8 C E K T > gpurun_out/r01_g/configs.jsonl 2>gpurun_out/r01_g/configs.err
timeout -k 10 200 python tools/gemm_trace.py > gpurun_out/r01_g/gemm_trace.txt 2>&1
tail -3 gpurun_out/r01_g/configs.jsonl
