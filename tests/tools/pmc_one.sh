#!/bin/bash
# usage: KRE=regex [PMC_CMD="command"] bash tests/tools/pmc_one.sh   (two cheap SQ passes of the kernels matching KRE)
R=$PWD; cd /tmp; export TMPDIR=/tmp; export KBENCH_SERIAL=1
[ -x $R/tests/tools/kbench ] || g++ -std=c++17 -O1 -o $R/tests/tools/kbench $R/tests/tools/kbench.cpp -ldl
B=${PMC_CMD:-"$R/tests/tools/kbench $R/bliss-rs_amd/libblissgpu.so ${SONGS:-128} 180 1"}
rm -rf $R/gpurun_out/pmco
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_WAVES"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --kernel-trace --kernel-include-regex "${KRE:-stft8192}" --output-format csv -d $R/gpurun_out/pmco/$i -o p -- $B > $R/gpurun_out/pmco_$i.log 2>&1
  echo "set $i rc=$?"
done
cd $R; python tests/tools/pmc_table.py gpurun_out/pmco
