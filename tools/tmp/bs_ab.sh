set -o pipefail
O=gpurun_out/bsab; mkdir -p $O
for i in 1 2; do
python bench.py --steps 4 --warmup 2 --no-cpu-baseline --weights fp8 > $O/on_$i.log 2>&1 && tail -1 $O/on_$i.log | cut -c100-180
FLUX2MI_LIB=libflux2mi_dev.so FLUX2MI_FP8_BLOCKS=0 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --weights fp8 > $O/off_$i.log 2>&1 && tail -1 $O/off_$i.log | cut -c100-180
done
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --weights fp8 --model klein-9b > $O/on9.log 2>&1 && tail -1 $O/on9.log | cut -c100-180
FLUX2MI_LIB=libflux2mi_dev.so FLUX2MI_FP8_BLOCKS=0 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --weights fp8 --model klein-9b > $O/off9.log 2>&1 && tail -1 $O/off9.log | cut -c100-180
