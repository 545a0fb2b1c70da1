#!/bin/bash
# quick perf+parity iteration on the GPU box: 256 songs, kernel table
python bench.py --songs ${SONGS:-256} --steps 2 --warmup 1 --cpu-songs 8 --no-pairwise --no-host-feed > gpurun_out/perf.log 2>&1; echo rc=$?
tail -1 gpurun_out/perf.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('songs/s', d['value'], 'ms/step', d['ms_per_step'])
for k,v in sorted(d['roofline']['kernels_ms_per_step'].items(), key=lambda kv:-kv[1]): print(f'  {k:22s} {v:9.3f}')
print(d.get('cpu_baseline'))
"
