#!/bin/bash
# soak of the final binary: -m gpu suite, ragged random batches, random musical songs, the 1024 bench songs against the
# oracle with the noise floor, run-to-run determinism (kbench, 60 runs x 768 songs through five chunks)
R=$PWD; O=$R/gpurun_out/soak; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 900 python tests/tools/random_check.py --songs 768 --seeds 1 2 > $O/r05_random_check.json 2> $O/random.err; echo "random rc=$?"; cut -c1-600 $O/r05_random_check.json
timeout 900 python tests/tools/musical_check.py --songs 300 --seed 1 > $O/r05_musical_check.json 2> $O/musical.err; echo "musical rc=$?"; cut -c1-700 $O/r05_musical_check.json
timeout 900 python tests/tools/musical_check.py --songs 300 --seed 2 --mods > $O/r05_musical_check_mods.json 2>> $O/musical.err; echo "musical mods rc=$?"; cut -c1-700 $O/r05_musical_check_mods.json
timeout 1500 python tests/tools/full_check.py --noise-floor > $O/r05_full_check_1024songs.json 2> $O/full.err; echo "full rc=$?"; cut -c1-1500 $O/r05_full_check_1024songs.json
KBENCH_WS_LIMIT_MB=2048 KBENCH_DETERMINISM=60 timeout 600 tests/tools/kbench bliss-rs_amd/libblissgpu.so 768 60 1 > $O/r05_determinism_60runs_768songs.txt 2>&1; echo "determinism rc=$?"; tail -3 $O/r05_determinism_60runs_768songs.txt | cut -c1-400
