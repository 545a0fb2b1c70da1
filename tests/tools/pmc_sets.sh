#!/bin/bash
# rocprofv3 --pmc passes over arbitrary counter sets (one pass per ';'-separated set) of the kernels matching KRE:
#   KRE=stft8192 SETS="GRBM_GUI_ACTIVE SQ_BUSY_CYCLES;SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA" [PMC_CMD="..."] bash tests/tools/pmc_sets.sh
# (sets of TA_* / TCP_* counters did not finish within the 120 s limit of a pass on this pool: keep to SQ_* / GRBM_* here)
R=$PWD; cd /tmp; export TMPDIR=/tmp; export KBENCH_SERIAL=1
[ -x $R/tests/tools/kbench ] || g++ -std=c++17 -O1 -o $R/tests/tools/kbench $R/tests/tools/kbench.cpp -ldl
B=${PMC_CMD:-"$R/tests/tools/kbench $R/bliss-rs_amd/libblissgpu.so ${SONGS:-128} 180 1"}
rm -rf $R/gpurun_out/pmcs; i=0
IFS=';' read -ra ARR <<< "$SETS"
for set in "${ARR[@]}"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --kernel-trace --kernel-include-regex "${KRE:-stft8192}" --output-format csv -d $R/gpurun_out/pmcs/$i -o p -- $B > $R/gpurun_out/pmcs_$i.log 2>&1
  echo "set $i ($set) rc=$?"
done
cd $R; python tests/tools/pmc_table.py gpurun_out/pmcs
