for a in 0 1 2 3 5 6 7 8; do echo "ABL=$a"; BLISSGPU_ABL=$a BLISSGPU_SERIAL=1 python bench.py --songs 128 --steps 2 --warmup 1 --no-cpu-baseline --no-pairwise 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('   stft8192', d['roofline']['kernels_ms_per_step']['stft8192_kernel'])"; done
