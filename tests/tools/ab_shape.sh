#!/bin/bash
# A/B of the two shapes of stft8192_kernel in ONE binary (BLISSGPU_OPT_STFT_SHAPE): scheduled 1024-song step and every kernel
# alone on 512 songs, two rounds; equal hashes = bit-identical rows
cd $(dirname $0)/../..; out=gpurun_out/${1:-stft_shape_ab.txt}; mkdir -p gpurun_out; rm -f $out
for rep in 1 2; do for shape in ${SHAPES:-0 1}; do
  echo "# shape $shape" >> $out
  KBENCH_STFT_SHAPE=$shape timeout 120 tests/tools/kbench bliss-rs_amd/libblissgpu.so 1024 180 3 >> $out 2>&1
  KBENCH_STFT_SHAPE=$shape KBENCH_SERIAL=1 timeout 120 tests/tools/kbench bliss-rs_amd/libblissgpu.so 512 180 2 >> $out 2>&1
done; done
sed -E 's/ (onset|tune_select|tune_final|summary|assemble|rolloff_fix)_kernel=[0-9.]+//g; s/ row0=.*//' $out
