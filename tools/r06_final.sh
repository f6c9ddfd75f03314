#!/bin/bash
# round 6, the final measurement batch on ONE box: round_measure.sh (-> gpurun_out/rm), the resolution lines, the stated tolerance on this build,
# the same-box A/B of the VAE mid attention (dev library) and of the split-K launches
O=gpurun_out/r06g; mkdir -p $O
run() { local t=$1; shift; timeout -k 10 "$t" "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILLED: $*"; exit 1; fi; return 0; }
run 1000 bash tools/round_measure.sh > $O/round_measure.log 2>&1
cd $GRAFT_REPO_ROOT
for hw in "512 512" "1024 1536" "2048 2048"; do set -- $hw; run 400 python bench.py --height $1 --width $2 --steps 3 --warmup 1 --no-cpu-baseline >> $O/resolutions.jsonl 2>>$O/bench.err; done
run 300 python bench.py --config A --steps 10 --warmup 2 --no-cpu-baseline >> $O/resolutions.jsonl 2>>$O/bench.err
run 400 python tools/trajectory_tolerance.py 28 1024 --against profiles/_traj/oracle_1024.pt > $O/trajectory_tolerance_1024.json 2>>$O/bench.err
for rep in 1 2; do for f in 0 1; do
  echo "FLUX2MI_VAE_FLASH=$f" >> $O/vae_flash_ab.txt
  FLUX2MI_LIB=libflux2mi_dev.so FLUX2MI_VAE_FLASH=$f run 200 python tools/vae_bench.py standard 10 >> $O/vae_flash_ab.txt 2>&1
  FLUX2MI_LIB=libflux2mi_dev.so FLUX2MI_VAE_FLASH=$f run 200 python tools/vae_bench.py small 10 >> $O/vae_flash_ab.txt 2>&1
done; done
for f in 0 1; do
  echo "FLUX2MI_VAE_FLASH=$f" >> $O/vae_flash_ab.txt
  FLUX2MI_LIB=libflux2mi_dev.so FLUX2MI_VAE_FLASH=$f run 300 python tools/vae_bench.py standard 3 256 256 >> $O/vae_flash_ab.txt 2>&1
done
for rep in 1 2; do for f in 0 1; do
  echo "FLUX2MI_GEMM_SPLITK=$f" >> $O/split_k_ab.txt
  FLUX2MI_LIB=libflux2mi_dev.so FLUX2MI_GEMM_SPLITK=$f run 300 python bench.py --config A --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --no-mfma-stream 2>>$O/bench.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['workload'][:36], d['seconds_per_image'], d['denoise_ms_per_denoise_step'], d['output_check']['image_crc32'])" >> $O/split_k_ab.txt
  FLUX2MI_LIB=libflux2mi_dev.so FLUX2MI_GEMM_SPLITK=$f run 300 python bench.py --height 512 --width 512 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-mfma-stream 2>>$O/bench.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['workload'][:36], d['seconds_per_image'], d['denoise_ms_per_denoise_step'], d['output_check']['image_crc32'])" >> $O/split_k_ab.txt
done; done
grep -v amdgpu $O/vae_flash_ab.txt; cat $O/split_k_ab.txt; cat $O/trajectory_tolerance_1024.json | cut -c1-300; tail -5 $O/round_measure.log
