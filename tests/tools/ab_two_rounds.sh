#!/bin/bash
# A/B of build variants: bash tests/tools/r5_ab.sh <out> <tag>...   (scheduled 1024-song step x3, then alone on 512)
out=gpurun_out/$1; shift; mkdir -p gpurun_out; rm -f $out
for rep in 1 2; do bash tests/tools/ab_many.sh $out "$@"; done
sed -E 's/ (onset|tune_select|tune_final|summary|assemble|rolloff_fix)_kernel=[0-9.]+//g; s/ row0=.*//' $out
