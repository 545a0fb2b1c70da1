#!/bin/bash
R=$PWD; O=$R/gpurun_out/r5c4; rm -rf $O; mkdir -p $O
timeout 300 python tests/tools/resample_bench.py > $O/r05_resample_kernel.txt 2>&1; cat $O/r05_resample_kernel.txt
TAG=r05 bash tests/tools/profile_round.sh > $O/profile_round.log 2>&1; tail -150 $O/profile_round.log
