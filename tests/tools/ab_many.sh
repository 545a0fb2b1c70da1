#!/bin/bash
# A/B of several builds on the same box: bash tests/tools/ab_many.sh <out file> <tag> [<tag> ...]
# (tag "" = bliss-rs_amd/libblissgpu.so, otherwise libblissgpu_<tag>.so); per build: one 1024-song step as scheduled,
# then every kernel alone on 512 songs (KBENCH_SERIAL=1).  Equal hashes = bit-identical rows.
out=$1; shift
cd $(dirname $0)/../..
for tag in "$@"; do
  lib=bliss-rs_amd/libblissgpu${tag:+_$tag}.so; [ "$tag" = "default" ] && lib=bliss-rs_amd/libblissgpu.so
  [ -f $lib ] || { echo "missing $lib" >> $out; continue; }
  timeout 120 tests/tools/kbench $lib 1024 180 3 >> $out 2>&1
  KBENCH_SERIAL=1 timeout 120 tests/tools/kbench $lib 512 180 2 >> $out 2>&1
done
