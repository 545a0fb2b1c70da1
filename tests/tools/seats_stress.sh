#!/bin/bash
# The single-song front with many seats on ONE GPU (an 8-GPU node has eight seats): tests/cpp/test_threads with 32 threads
# x 16 calls against 2 / 4 / 8 default contexts on device 0, REPS times each; every threaded row must be bit-identical to
# the serial run and no run may stall.  (The CPU form of this test, tests/cpp/test_front.cpp, is in the not-gpu suite; the
# two-seat GPU form is in the -m gpu suite.)       usage (GPU box): [SEATS="0,0,0,0 0,0,0,0,0,0,0,0"] REPS=4 bash tests/tools/seats_stress.sh
R=$PWD; mkdir -p /tmp/seats
g++ -std=c++17 -O1 -pthread $R/tests/cpp/test_threads.cpp -o /tmp/seats/test_threads -ldl -L$R/bliss-rs_amd -lblissgpu -Wl,-rpath,$R/bliss-rs_amd || exit 1
for d in ${SEATS:-0,0 0,0,0,0 0,0,0,0,0,0,0,0}; do
  for i in $(seq 1 ${REPS:-4}); do
    BLISSGPU_DEFAULT_DEVICES=$d timeout 60 /tmp/seats/test_threads 32 16 > /tmp/seats/out.txt 2>&1
    echo "seats $d run $i rc=$? $(grep -c 'all checks passed' /tmp/seats/out.txt) $(grep default_devices /tmp/seats/out.txt | head -1)"
  done
done
