#!/bin/bash
# A/B of the chunk stream schedules (BLISSGPU_SCHED = 0 round-1 order, 1 tuning beside FFT-512, 2 tuning + chroma beside FFT-512)
R=$PWD; O=$R/gpurun_out/sched; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
for S in 0 1 2 0 1 2; do
  BLISSGPU_SCHED=$S timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-feed --no-pairwise --no-small-calls > $O/b_$S.json 2> $O/b_$S.err
  python - "$O/b_$S.json" $S <<'P'
import json,sys
r=json.load(open(sys.argv[1])); print("sched",sys.argv[2],r["value"],"songs/s",r["ms_per_step"],"ms", {k:round(v,2) for k,v in r["roofline"]["kernels_ms_per_step"].items()})
P
done
BLISSGPU_SCHED=2 timeout 600 python bench.py --config mixed --steps 2 --warmup 1 --no-cpu-baseline > $O/mixed_2.json 2>$O/mixed_2.err; python -c "import json;r=json.load(open('$O/mixed_2.json'));print('mixed sched2',r['value'],r['three_minute_song_equivalents_per_sec'],r['ms_per_step'])"
BLISSGPU_SCHED=0 timeout 600 python bench.py --config mixed --steps 2 --warmup 1 --no-cpu-baseline > $O/mixed_0.json 2>$O/mixed_0.err; python -c "import json;r=json.load(open('$O/mixed_0.json'));print('mixed sched0',r['value'],r['three_minute_song_equivalents_per_sec'],r['ms_per_step'])"
