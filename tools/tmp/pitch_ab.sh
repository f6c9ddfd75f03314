set -o pipefail
O=gpurun_out/pm; mkdir -p $O
export FLUX2MI_LIB=libflux2mi_dev.so
for i in 1 2; do
python bench.py --steps 4 --warmup 2 --no-cpu-baseline > $O/bf16_pad_$i.log 2>&1 && tail -1 $O/bf16_pad_$i.log | cut -c100-190
FLUX2MI_W_PAD=0 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > $O/bf16_nopad_$i.log 2>&1 && tail -1 $O/bf16_nopad_$i.log | cut -c100-190
done
python bench.py --steps 4 --warmup 2 --no-cpu-baseline --weights fp8 > $O/fp8_pad.log 2>&1 && tail -1 $O/fp8_pad.log | cut -c100-190
FLUX2MI_W_PAD=0 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --weights fp8 > $O/fp8_nopad.log 2>&1 && tail -1 $O/fp8_nopad.log | cut -c100-190
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --weights fp8 --model klein-9b > $O/fp8_9b_pad.log 2>&1 && tail -1 $O/fp8_9b_pad.log | cut -c100-190
FLUX2MI_W_PAD=0 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --weights fp8 --model klein-9b > $O/fp8_9b_nopad.log 2>&1 && tail -1 $O/fp8_9b_nopad.log | cut -c100-190
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --model klein-9b > $O/bf16_9b_pad.log 2>&1 && tail -1 $O/bf16_9b_pad.log | cut -c100-190
FLUX2MI_W_PAD=0 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --model klein-9b > $O/bf16_9b_nopad.log 2>&1 && tail -1 $O/bf16_9b_nopad.log | cut -c100-190
