#!/bin/bash
# one scheduling option (a bench.py flag), values alternated on the same box: FLAG=--tail-mode VALS="0 1" bash tests/tools/ab_env.sh [bench args]
R=$PWD
for rep in 1 2; do for v in $VALS; do
  timeout 600 python bench.py $FLAG $v --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline --no-host-feed --no-pairwise --no-small-calls --no-playlist "$@" 2>/dev/null | tail -1 | python -c "
import sys,json
r=json.loads(sys.stdin.read()); print('$FLAG $v', r['value'],'songs/s',r['ms_per_step'],'ms', r['config'].get('chunks_per_step'))"
done; done
