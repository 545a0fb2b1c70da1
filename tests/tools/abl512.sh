for a in 0 2 5 6 7; do echo "ABL512=$a"; BLISSGPU_ABL512=$a BLISSGPU_SERIAL=1 python bench.py --songs 128 --steps 2 --warmup 1 --no-cpu-baseline --no-pairwise 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('   fft512', d['roofline']['kernels_ms_per_step']['fft512_kernel'])"; done
