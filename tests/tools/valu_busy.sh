#!/bin/bash
# VALU-busy fraction of the three big analysis kernels: one rocprofv3 --pmc pass (kernel-trace only) with SQ_ACTIVE_INST_VALU,
# SQ_BUSY_CYCLES, GRBM_GUI_ACTIVE and SQ_WAVE_CYCLES over a serial 128-song step of tests/tools/kbench.
#   usage (GPU box): bash tests/tools/valu_busy.sh        -> gpurun_out/valu/valu_busy.{txt,json}
R=$PWD; cd /tmp; export TMPDIR=/tmp; export KBENCH_SERIAL=1
[ -x $R/tests/tools/kbench ] || g++ -std=c++17 -O1 -o $R/tests/tools/kbench $R/tests/tools/kbench.cpp -ldl
B="$R/tests/tools/kbench ${LIB:-$R/bliss-rs_amd/libblissgpu.so} ${SONGS:-128} 180 1"
OUT=$R/gpurun_out/valu; rm -rf $OUT; mkdir -p $OUT
timeout 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --kernel-trace --kernel-include-regex "stft8192|fft512_kernel|chroma_kernel" --output-format csv -d $OUT/1 -o p -- $B > $OUT/pass.log 2>&1
echo "pass rc=$?"
cd $R; python tests/tools/valu_busy.py $OUT
