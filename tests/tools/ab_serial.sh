#!/bin/bash
# every kernel alone (KBENCH_SERIAL=1) on 512 songs for a list of build tags: bash tests/tools/ab_serial.sh <out> <tag>...
out=$1; shift
cd $(dirname $0)/../..
for tag in "$@"; do
  lib=bliss-rs_amd/libblissgpu_$tag.so; [ "$tag" = "default" ] && lib=bliss-rs_amd/libblissgpu.so
  KBENCH_SERIAL=1 timeout 90 tests/tools/kbench $lib 512 180 2 2>&1 | sed -E 's/ (onset|beat|tune_select|tune_final|summary|assemble|rolloff_fix)_kernel=[0-9.]+//g; s/ row0=.*//' >> $out
done
