#!/bin/bash
# VAE decode at 1024x1024: per-kernel time only (rocprofv3 --kernel-trace --stats) + the per-launch timeline of the last decode.
# usage: tools/vae_stats.sh [standard|small] [tag]
set -o pipefail
V=${1:-standard}; T=${2:-s}; O=gpurun_out/vae_$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 tools/vae_bench.py $V 10 | tee $O/vae_bench.txt || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 tools/vae_bench.py $V 3 > $O/prof.log 2>&1 || exit 1
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/vae_kernel_stats.csv
python3 tools/kernel_timeline.py $O/prof nchw_to_nhwc > $O/vae_timeline.txt
rm -rf $O/prof
head -14 $O/vae_kernel_stats.csv | cut -c1-150
