#!/bin/bash
# One GPU call: the -m gpu suite, the three bench configurations and (optionally) the SQ counter passes of one kernel.
#   usage (GPU box): [KRE=stft8192] [SKIP_TESTS=1] bash tests/tools/gpu_check.sh
R=$PWD; O=$R/gpurun_out/check; rm -rf $O; mkdir -p $O
if [ -z "$SKIP_TESTS" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
fi
timeout 600 python bench.py > $O/bench_batch.json 2> $O/bench_batch.err; echo "bench batch rc=$?"; head -c 1500 $O/bench_batch.json; echo
timeout 900 python bench.py --config mixed --steps 2 --warmup 1 > $O/bench_mixed.json 2> $O/bench_mixed.err; echo "bench mixed rc=$?"; head -c 2500 $O/bench_mixed.json; echo; tail -3 $O/bench_mixed.err
timeout 900 python bench.py --config library --steps 2 --warmup 1 > $O/bench_library.json 2> $O/bench_library.err; echo "bench library rc=$?"; head -c 2000 $O/bench_library.json; echo; tail -3 $O/bench_library.err
if [ -n "$KRE" ]; then
  KRE=$KRE bash tests/tools/pmc_one.sh > $O/pmc_$KRE.txt 2>&1; tail -30 $O/pmc_$KRE.txt
fi
