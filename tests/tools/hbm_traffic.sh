#!/bin/bash
# HBM traffic of every analysis kernel: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (they do not fit
# one pass on gfx950), kernel-trace only.  usage (GPU box): SONGS=256 bash tests/tools/hbm_traffic.sh
R=$PWD; cd /tmp; export TMPDIR=/tmp; export KBENCH_SERIAL=1
S=${SONGS:-256}
# the batch is driven by tests/tools/kbench (C ABI, no Python: nothing but the library's kernels in the trace)
[ -x $R/tests/tools/kbench ] || g++ -std=c++17 -O1 -o $R/tests/tools/kbench $R/tests/tools/kbench.cpp -ldl
B="$R/tests/tools/kbench ${LIB:-$R/bliss-rs_amd/libblissgpu.so} $S 180 1"
OUT=${TRAFFIC_OUT:-$R/gpurun_out/hbm}; rm -rf $OUT
i=0
for set in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout ${PMC_TIMEOUT:-200} rocprofv3 --pmc $set --kernel-trace --kernel-include-regex "bg::" --output-format csv -d $OUT/$i -o p -- $B > $R/gpurun_out/hbm_$i.log 2>&1
  echo "pass $i ($set) rc=$?"
done
cd $R; python tests/tools/hbm_traffic.py $OUT $S
