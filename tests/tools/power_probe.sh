#!/bin/bash
# Is the chip at its power cap under the analysis kernels?  Samples rocm-smi (power, sclk, mclk, temperature) every 0.2 s while
# kbench runs 1024-song steps back to back; also each big kernel alone (KBENCH_SERIAL=1 runs them one after another).
#   bash tests/tools/power_probe.sh > gpurun_out/power_probe.txt
R=$PWD
[ -x $R/tests/tools/kbench ] || g++ -std=c++17 -O1 -o $R/tests/tools/kbench $R/tests/tools/kbench.cpp -ldl
rocm-smi --showmaxpower --showclocks --showpower 2>&1 | grep -v "^=\|^$" | head -30
echo "---- idle sample"
rocm-smi --showpower --showclocks -t 2>&1 | grep -i "power\|sclk\|mclk\|Temperature (Sensor junction)" | head -8
(tests/tools/kbench bliss-rs_amd/libblissgpu.so 1024 180 150 > /tmp/kb.out 2>&1) &
KB=$!
sleep 3   # synthesis + warm-up
echo "---- under load (scheduled steps)"
for i in $(seq 1 16); do
  rocm-smi --showpower --showclocks 2>&1 | grep -i "Package Power\|sclk" | sed 's/^GPU\[0\]\s*: //' | tr '\n' '|'; echo
  sleep 0.4
done
wait $KB; tail -1 /tmp/kb.out | cut -c1-200
