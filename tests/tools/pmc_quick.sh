#!/bin/bash
# PMC passes (counters only + kernel trace) for the analysis kernels on a 128-song batch
R=$PWD; cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --songs ${SONGS:-128} --steps 1 --warmup 1 --no-cpu-baseline --no-pairwise"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT" \
           "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAVES" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F64"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc/$i -o p -- $B > $R/gpurun_out/pmc_$i.log 2>&1
  echo "set $i rc=$?"
done
cd $R
python tests/tools/pmc_table.py gpurun_out/pmc
