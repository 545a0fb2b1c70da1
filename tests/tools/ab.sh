#!/bin/bash
# A/B timing of two builds on the same GPU box: bliss-rs_amd/libblissgpu.so (A) vs bliss-rs_amd/libblissgpu_b.so (B)
# usage: KERNELS="stft8192_kernel fft512_kernel" SONGS=256 REPS=3 bash tests/tools/ab.sh
R=$PWD
for rep in $(seq 1 ${REPS:-3}); do
  for v in A B; do
    lib=$R/bliss-rs_amd/libblissgpu.so; [ $v = B ] && lib=$R/bliss-rs_amd/libblissgpu_b.so
    BLISSGPU_LIB=$lib python bench.py $([ "${SERIAL:-1}" = 1 ] && echo --serial) --songs ${SONGS:-256} --steps 3 --warmup 1 --no-cpu-baseline --no-pairwise --no-host-feed 2>/dev/null | tail -1 | V=$v KERNELS="${KERNELS:-stft8192_kernel}" python -c "
import sys,json,os
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_ms_per_step']
print(os.environ['V'], 'ms/step', d['ms_per_step'], ' '.join(f'{n}={k[n]}' for n in os.environ['KERNELS'].split()))"
  done
done
