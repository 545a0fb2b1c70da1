#!/bin/bash
# SQ counters and shader clock of stft8192_kernel in both shapes (BLISSGPU_OPT_STFT_SHAPE through KBENCH_STFT_SHAPE)
R=$PWD; O=$R/gpurun_out/shape; rm -rf $O; mkdir -p $O
for shape in 0 1; do
  export KBENCH_STFT_SHAPE=$shape
  KRE=stft8192 SONGS=256 bash tests/tools/pmc_one.sh 2>&1 | grep -v amdgpu.ids > $O/r05_pmc_stft8192_shape$shape.txt; tail -18 $O/r05_pmc_stft8192_shape$shape.txt
  SONGS=512 bash tests/tools/kernel_clocks.sh 2>&1 | grep -E "kernel |stft8192|fft512|chroma_kernel" > $O/r05_kernel_clocks_shape$shape.txt; cat $O/r05_kernel_clocks_shape$shape.txt
done
