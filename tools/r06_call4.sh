#!/bin/bash
# round 6, GPU call 4: vendor comparison on this build, resolution bench lines, small-S kernel profile
O=gpurun_out/r06c; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { local t=$1; shift; timeout -k 10 "$t" "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILLED: $*"; exit 1; fi; return 0; }
run 300 python tools/gemm_bench.py --check > $O/gemm_vs_hipblaslt.log 2>&1
run 300 python tools/gemm_bench.py --check --9b >> $O/gemm_vs_hipblaslt.log 2>&1
for hw in "512 512" "1024 1536" "2048 2048"; do set -- $hw; run 400 python bench.py --height $1 --width $2 --steps 3 --warmup 1 --no-cpu-baseline >> $O/resolutions.jsonl 2>>$O/bench.err; done
run 300 python bench.py --config A --steps 10 --warmup 2 --no-cpu-baseline >> $O/resolutions.jsonl 2>>$O/bench.err
run 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_a -- python3 bench.py --config A --steps 20 --warmup 2 --no-cpu-baseline --no-mfma-stream --no-secondary > $O/prof_a.log 2>&1
find $O/prof_a -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/config_a_kernel_stats.csv
run 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_512 -- python3 bench.py --height 512 --width 512 --steps 5 --warmup 1 --no-cpu-baseline --no-mfma-stream --no-secondary > $O/prof_512.log 2>&1
find $O/prof_512 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r512_kernel_stats.csv
rm -rf $O/prof_a $O/prof_512
tail -20 $O/gemm_vs_hipblaslt.log; cut -c1-600 $O/resolutions.jsonl
