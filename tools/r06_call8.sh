#!/bin/bash
# round 6: round-5 library against this round's on one box (headline + config D), smoke(), VAE kernel profiles (standard / small)
O=gpurun_out/r06i; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { local t=$1; shift; timeout -k 10 "$t" "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILLED: $*"; exit 1; fi; return 0; }
run 600 bash tools/ab_libs.sh $O/ab_b libflux2mi_r05.so libflux2mi.so --steps 4 --warmup 1 > $O/r05_vs_r06_config_b.txt 2>&1
run 600 bash tools/ab_libs.sh $O/ab_d libflux2mi_r05.so libflux2mi.so --config D --steps 2 --warmup 1 > $O/r05_vs_r06_config_d.txt 2>&1
run 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
for v in standard small; do
  run 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -- python3 tools/vae_bench.py $v 5 > $O/prof_$v.log 2>&1
  find $O/prof_$v -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/vae_${v}_kernel_stats.csv
  rm -rf $O/prof_$v
done
cat $O/r05_vs_r06_config_b.txt $O/r05_vs_r06_config_d.txt; tail -2 $O/smoke.txt
