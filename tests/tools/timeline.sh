#!/bin/bash
# Kernel timeline of the default bench step (GPU box): bash tests/tools/timeline.sh [bench args]
R=$PWD; O=$R/gpurun_out/tl; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $O/trace -o tl -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-host-feed --no-playlist --no-small-calls --no-pairwise "$@" > $O/trace.log 2>&1; echo "trace rc=$?"
cd $R
DB=$(find $O/trace -name "*.db" | head -1)
python tests/tools/timeline.py $DB | tee $O/timeline.txt
rm -rf $O/trace
