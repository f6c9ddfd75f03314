#!/bin/bash
# end-of-round measurement batch on ONE box: bench (bf16 with cpu_baseline, fp8), the other configurations, kernel stats, PMC traffic
# passes for roofline.traffic.  Outputs under gpurun_out/rm/ (copied into profiles/ by hand).
set -o pipefail
O=gpurun_out/rm; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --steps 5 --warmup 1 > $O/bench.json.log 2>$O/bench.err || exit 1
tail -1 $O/bench.json.log | cut -c1-400
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --weights fp8 > $O/bench_fp8.json.log 2>>$O/bench.err || exit 1
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --attn-softmax online > $O/bench_online_softmax.json.log 2>>$O/bench.err || exit 1
python tools/run_configs.py A D D8 C E K T > $O/configs.jsonl 2>>$O/bench.err || exit 1
tail -8 $O/configs.jsonl | cut -c1-200
# the BASELINE.json configurations D and E through bench.py's own contract (one GPU here; `--gpus 8 --config D` / `--gpus 4 --config E` on a node)
python bench.py --config D --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_config_d.json.log 2>>$O/bench.err || exit 1
python bench.py --config E --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_config_e.json.log 2>>$O/bench.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-mfma-stream --no-secondary > $O/prof.log 2>&1 || exit 1
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv
python3 tools/linear1_from_trace.py $O/prof $O/linear1_rocprof.json
grep '^{"metric' $O/prof.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench roofline.avg_launch_ms (HIP events) of the same run:', d['roofline']['avg_launch_ms'], 'over', d['roofline']['launches'], 'launches')" | tee $O/linear1_bench_events.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 tools/one_forward.py klein-4b 2 > $O/fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 tools/one_forward.py klein-4b 2 > $O/write.log 2>&1 || exit 1
python3 tools/pmc_summary.py $O/fetch > $O/pmc_fetch_size.txt
python3 tools/pmc_summary.py $O/write > $O/pmc_write_size.txt
python3 tools/linear1_traffic.py $O/fetch $O/write $O/linear1_traffic.json | tail -5
python3 tools/secondary_traffic.py $O/pmc_fetch_size.txt $O/pmc_write_size.txt $O/secondary_traffic.json | tail -3
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $O/sq1 -- python3 tools/one_forward.py klein-4b 2 > $O/sq1.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq2 -- python3 tools/one_forward.py klein-4b 2 > $O/sq2.log 2>&1 || exit 1
python3 tools/pmc_sq.py $O/sq1 $O/sq2 > $O/pmc_sq.txt
head -12 $O/pmc_sq.txt | cut -c1-220
rm -rf $O/prof/*/*kernel_trace.csv $O/fetch $O/write $O/sq1 $O/sq2
echo done
