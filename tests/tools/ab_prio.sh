#!/bin/bash
R=$PWD; O=$R/gpurun_out/prio; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for P in 1 0 1 0; do
  BLISSGPU_SIDE_PRIORITY=$P timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-feed --no-pairwise --no-small-calls > $O/b_$P.json 2> $O/b_$P.err
  python - "$O/b_$P.json" $P <<'PY'
import json,sys
r=json.load(open(sys.argv[1])); print("prio",sys.argv[2],r["value"],"songs/s",r["ms_per_step"],"ms", {k:round(v,2) for k,v in r["roofline"]["kernels_ms_per_step"].items()})
PY
done
for P in 1 0; do
BLISSGPU_SIDE_PRIORITY=$P timeout 600 python bench.py --config mixed --steps 2 --warmup 1 --no-cpu-baseline > $O/mixed_$P.json 2>$O/mixed_$P.err; python -c "import json;r=json.load(open('$O/mixed_$P.json'));print('mixed prio $P',r['value'],r['three_minute_song_equivalents_per_sec'],r['ms_per_step'],r['config']['chunks_per_step'], r['roofline']['kernels_ms_per_step'])"
done
