#!/bin/bash
# VAE decode at 1024x1024 on ONE box: per-kernel time (rocprofv3 --kernel-trace --stats), the per-launch timeline, and the PMC passes
# (FETCH_SIZE / WRITE_SIZE / SQ) of every VAE kernel.  usage: tools/vae_measure.sh [standard|small] [tag]; outputs under gpurun_out/vae_<tag>/
set -o pipefail
V=${1:-standard}; T=${2:-a}; O=gpurun_out/vae_$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 tools/vae_bench.py $V 10 | tee $O/vae_bench.txt || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 tools/vae_bench.py $V 3 > $O/prof.log 2>&1 || exit 1
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/vae_kernel_stats.csv
python3 tools/kernel_timeline.py $O/prof latents_to_vae > $O/vae_timeline.txt 2>/dev/null || true
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc_$C -- python3 tools/vae_bench.py $V 1 > $O/pmc_$C.log 2>&1 || exit 1
  python3 tools/pmc_summary.py $O/pmc_$C > $O/vae_pmc_$(echo $C | tr A-Z a-z).txt
done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $O/sq1 -- python3 tools/vae_bench.py $V 1 > $O/sq1.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq2 -- python3 tools/vae_bench.py $V 1 > $O/sq2.log 2>&1 || exit 1
python3 tools/pmc_sq.py $O/sq1 $O/sq2 > $O/vae_pmc_sq.txt
rm -rf $O/prof/*/*kernel_trace.csv $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/sq1 $O/sq2
head -30 $O/vae_kernel_stats.csv | cut -c1-160
echo done
