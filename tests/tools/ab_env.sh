#!/bin/bash
# A/B of one environment switch on the same box: VAR=BLISSGPU_TAIL_MODE VALS="0 1" bash tests/tools/ab_env.sh
for rep in 1 2 3; do for v in $VALS; do
  env $VAR=$v timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-host-feed --no-pairwise --no-small-calls ${EXTRA} 2>/dev/null | tail -1 | python -c "
import sys,json
r=json.loads(sys.stdin.read()); print('$VAR=$v', r['value'],'songs/s',r['ms_per_step'],'ms', {k:round(v,2) for k,v in r['roofline']['kernels_ms_per_step'].items()})"
done; done
