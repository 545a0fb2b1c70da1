#!/bin/bash
# BLISSGPU_OPT_TAIL_MODE = N >= 2 (beat state machines on a stream confined to N CUs, beside the FFT-8192 kernel) against the
# default schedule, 1024 three-minute songs, alternating:  bash tests/tools/ab_tail_mask.sh <out> <rounds> <mode>...
out=$1; rounds=$2; shift 2
cd $(dirname $0)/../..
for r in $(seq $rounds); do
  for m in "$@"; do
    echo "== tail_mode $m" >> $out
    KBENCH_TAIL_MODE=$m timeout 120 tests/tools/kbench bliss-rs_amd/libblissgpu.so 1024 180 4 2>&1 | sed -E 's/ (tune_select|tune_final|assemble|onset)_kernel=[0-9.]+//g' >> $out
  done
done
