#!/bin/bash
O=gpurun_out/r06e; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { local t=$1; shift; timeout -k 10 "$t" "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILLED: $*"; exit 1; fi; return 0; }
run 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_a -- python3 bench.py --config A --steps 20 --warmup 2 --no-cpu-baseline --no-mfma-stream --no-secondary > $O/prof_a.log 2>&1
find $O/prof_a -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/config_a_kernel_stats.csv
run 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_512 -- python3 bench.py --height 512 --width 512 --steps 5 --warmup 1 --no-cpu-baseline --no-mfma-stream --no-secondary > $O/prof_512.log 2>&1
find $O/prof_512 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r512_kernel_stats.csv
rm -rf $O/prof_a $O/prof_512
