#!/bin/bash
# tests + A/B of the pipeline depth (BLISSGPU_PIPELINE_CHUNKS: 1 = one chunk per batch as in round 1)
R=$PWD; O=$R/gpurun_out/pipe; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log
for P in 1 2 4 8 1 2 4 8; do
  BLISSGPU_PIPELINE_CHUNKS=$P timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-feed --no-pairwise --no-small-calls > $O/b_$P.json 2> $O/b_$P.err
  python - "$O/b_$P.json" $P <<'PY'
import json,sys
r=json.load(open(sys.argv[1])); print("chunks",sys.argv[2],r["value"],"songs/s",r["ms_per_step"],"ms", {k:round(v,2) for k,v in r["roofline"]["kernels_ms_per_step"].items()})
PY
done
timeout 600 python bench.py --config mixed --steps 2 --warmup 1 --no-cpu-baseline > $O/mixed.json 2>$O/mixed.err; python -c "import json;r=json.load(open('$O/mixed.json'));print('mixed',r['value'],r['three_minute_song_equivalents_per_sec'],r['ms_per_step'],r['config']['chunks_per_step'])"
timeout 600 python bench.py --steps 3 > $O/bench_full.json 2>$O/bench_full.err; python -c "import json;r=json.load(open('$O/bench_full.json'));print('full',r['value'],r.get('cpu_baseline'),r.get('small_calls'))"
