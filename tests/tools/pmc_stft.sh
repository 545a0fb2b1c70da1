#!/bin/bash
R=$PWD; cd /tmp; export TMPDIR=/tmp; export BLISSGPU_SERIAL=1
B="python $R/bench.py --songs 128 --steps 1 --warmup 1 --no-cpu-baseline --no-pairwise"
i=0
for set in "TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum TA_BUFFER_COALESCED_READ_CYCLES_sum" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_MISC" \
           "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --kernel-include-regex "${KRE:-stft8192|fft512}" --output-format csv -d $R/gpurun_out/pmcs/$i -o p -- $B > $R/gpurun_out/pmcs_$i.log 2>&1
  echo "set $i rc=$?"
done
cd $R
python tests/tools/pmc_table.py gpurun_out/pmcs
