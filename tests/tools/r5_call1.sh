#!/bin/bash
# round-5 GPU call: the new resampler tests first (fast feedback), then the whole -m gpu suite
R=$PWD; O=$R/gpurun_out/r5c1; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_resample.py -m gpu -q -s --durations=5 > $O/pytest_resample.log 2>&1; echo "resample rc=$?"; tail -12 $O/pytest_resample.log
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
