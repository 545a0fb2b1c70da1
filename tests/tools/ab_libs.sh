#!/bin/bash
# bench.py's step on the same box for a list of builds, alternating, white noise then musical content:
#   bash tests/tools/ab_libs.sh <out> <rounds> <tag>...     (tag "default" = libblissgpu.so, else libblissgpu_<tag>.so)
out=$1; rounds=$2; shift 2
cd $(dirname $0)/../..
for cfg in batch musical; do
  for r in $(seq $rounds); do
    for tag in "$@"; do
      lib=$PWD/bliss-rs_amd/libblissgpu_$tag.so; [ "$tag" = "default" ] && lib=$PWD/bliss-rs_amd/libblissgpu.so
      BLISSGPU_LIB=$lib timeout 600 python bench.py --config $cfg --steps 6 --warmup 2 --no-cpu-baseline --no-host-feed --no-pairwise --no-small-calls 2>/dev/null | tail -1 | python -c "
import sys,json
r=json.loads(sys.stdin.read()); k=r['roofline']['kernels_ms_per_step']
print('$cfg $tag', r['value'],'songs/s',r['ms_per_step'],'ms', {n:round(k[n],3) for n in ('fft512_kernel','rolloff_fix_kernel','stft8192_kernel','chroma_kernel','summary_kernel')})" >> $out
    done
  done
done
