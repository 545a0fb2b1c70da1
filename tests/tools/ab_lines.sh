#!/bin/bash
# A (libblissgpu.so) vs B (libblissgpu_b.so) default bench lines on the same box, alternating; then the kernel timeline of A
R=$PWD
for v in A B A B; do
  lib=$R/bliss-rs_amd/libblissgpu.so; [ $v = B ] && lib=$R/bliss-rs_amd/libblissgpu_b.so
  BLISSGPU_LIB=$lib timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-feed --no-pairwise --no-small-calls --no-playlist "$@" 2>/dev/null | tail -1 | python -c "
import sys,json
r=json.loads(sys.stdin.read()); print('$v', r['value'],'songs/s',r['ms_per_step'],'ms')"
done
bash tests/tools/timeline.sh "$@" | tail -14
