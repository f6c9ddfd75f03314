set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in on off; do
O=gpurun_out/fp8prof_$v; mkdir -p $O
if [ $v = off ]; then export FLUX2MI_LIB=libflux2mi_dev.so FLUX2MI_FP8_BLOCKS=0; fi
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-mfma-stream --weights fp8 > $O/prof.log 2>&1 || exit 1
python3 tools/kernel_timeline.py $O/prof sinusoid > $O/timeline.txt
rm -rf $O/prof
done
echo ok
