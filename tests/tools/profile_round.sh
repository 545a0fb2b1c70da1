#!/bin/bash
# Round-end evidence on the GPU box: (1) the default bench line, (2) rocprofv3 --kernel-trace --stats of the
# same command, (3) HBM traffic (separate FETCH_SIZE / WRITE_SIZE --pmc passes).  Outputs under gpurun_out/round/;
# copy the summaries into profiles/ afterwards.   usage: TAG=r01 bash tests/tools/profile_round.sh
R=$PWD; TAG=${TAG:-r01}; O=$R/gpurun_out/round; rm -rf $O; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 600 $O/bench.json
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o bench -- python $R/bench.py --no-cpu-baseline --no-host-feed --no-playlist > $O/trace.log 2>&1; echo "trace rc=$?"
cd $R
DB=$(find $O/trace -name "*.db" | head -1)
python tests/tools/rocpd_stats.py $DB > $O/${TAG}_bench_1024songs.kernel_stats.txt; head -16 $O/${TAG}_bench_1024songs.kernel_stats.txt
rm -rf $O/trace
SONGS=${HBM_SONGS:-256} bash tests/tools/hbm_traffic.sh > $O/hbm.log 2>&1; tail -16 $O/hbm.log
cp gpurun_out/hbm/hbm_traffic.json gpurun_out/hbm/hbm_traffic.txt $O/ 2>/dev/null
