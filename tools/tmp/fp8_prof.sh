set -o pipefail
O=gpurun_out/fp8prof2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-mfma-stream --weights fp8 > $O/prof.log 2>&1 || exit 1
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/stats.csv
python3 tools/kernel_timeline.py $O/prof sinusoid > $O/timeline.txt
rm -rf $O/prof
head -8 $O/stats.csv | cut -c1-150
