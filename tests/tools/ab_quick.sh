#!/bin/bash
# parity tests of the new build, then A (libblissgpu.so) vs B (libblissgpu_b.so) on the same box, serial and overlapped
R=$PWD; O=$R/gpurun_out/abq; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log | cut -c1-300
KERNELS="${KERNELS:-stft8192_kernel fft512_kernel chroma_kernel}" SONGS=512 REPS=2 SERIAL=1 bash tests/tools/ab.sh
for v in A B A B; do
  lib=$R/bliss-rs_amd/libblissgpu.so; [ $v = B ] && lib=$R/bliss-rs_amd/libblissgpu_b.so
  BLISSGPU_LIB=$lib timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-feed --no-pairwise --no-small-calls 2>/dev/null | tail -1 | python -c "
import sys,json
r=json.loads(sys.stdin.read()); print('$v', r['value'],'songs/s',r['ms_per_step'],'ms', {k:round(v,2) for k,v in r['roofline']['kernels_ms_per_step'].items()})"
done
