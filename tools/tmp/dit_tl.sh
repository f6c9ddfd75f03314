set -o pipefail
O=gpurun_out/dit_tl; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 tools/one_forward.py klein-4b 3 > $O/log.txt 2>&1 || exit 1
python3 tools/kernel_timeline.py $O/t sinusoid > $O/timeline_bf16.txt
rm -rf $O/t
tail -3 $O/timeline_bf16.txt
