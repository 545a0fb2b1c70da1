#!/bin/bash
# Round-end evidence on the GPU box, in one call: (1) the default bench line and the mixed / library configurations and
# the node form, (2) rocprofv3 --kernel-trace --stats of the default bench command, (3) HBM traffic (separate FETCH_SIZE /
# WRITE_SIZE --pmc passes), (4) SQ counter passes of the two FFT kernels and of the pairwise kernel (self, general),
# (5) the kernel timeline of one step.  Outputs under gpurun_out/round/; copy the summaries into profiles/.
#   usage: TAG=r04 bash tests/tools/profile_round.sh        (STRESS_REPS=100: the seat stress right behind the counter passes)
R=$PWD; TAG=${TAG:-r04}; O=$R/gpurun_out/round; rm -rf $O; mkdir -p $O
[ -x $R/tests/tools/kbench ] || g++ -std=c++17 -O1 -o $R/tests/tools/kbench $R/tests/tools/kbench.cpp -ldl
# traffic first: bench.py reads the table it leaves under profiles/
SONGS=${HBM_SONGS:-256} bash tests/tools/hbm_traffic.sh > $O/hbm.log 2>&1; tail -16 $O/hbm.log
cp gpurun_out/hbm/hbm_traffic.json $O/hbm_traffic.json; cp gpurun_out/hbm/hbm_traffic.json profiles/hbm_traffic.json
cp gpurun_out/hbm/hbm_traffic.txt $O/${TAG}_hbm_traffic_256songs.txt
python bench.py > $O/${TAG}_bench_batch.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 400 $O/${TAG}_bench_batch.json; echo
python bench.py --config mixed --steps 2 --warmup 1 > $O/${TAG}_bench_mixed.json 2> $O/bench_mixed.err; echo "mixed rc=$?"; head -c 300 $O/${TAG}_bench_mixed.json; echo
python bench.py --config library --steps 2 --warmup 1 > $O/${TAG}_bench_library.json 2> $O/bench_library.err; echo "library rc=$?"; head -c 300 $O/${TAG}_bench_library.json; echo
python bench.py --node 2 --node-devices 0,0 --songs 512 > $O/${TAG}_bench_node_2_loopback_ranks.json 2> $O/bench_node.err; echo "node rc=$?"; head -c 200 $O/${TAG}_bench_node_2_loopback_ranks.json; echo
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o bench -- python $R/bench.py --no-cpu-baseline --no-host-feed --no-playlist --no-small-calls > $O/trace.log 2>&1; echo "trace rc=$?"
cd $R
DB=$(find $O/trace -name "*.db" | head -1)
python tests/tools/rocpd_stats.py $DB > $O/${TAG}_bench_1024songs.kernel_stats.txt; head -16 $O/${TAG}_bench_1024songs.kernel_stats.txt
rm -rf $O/trace
for K in stft8192 fft512 chroma_kernel tune_pass2 beat_acf; do
  KRE=$K bash tests/tools/pmc_one.sh 2>&1 | grep -v amdgpu.ids > $O/${TAG}_pmc_${K}.txt; tail -18 $O/${TAG}_pmc_${K}.txt
done
KRE=pairwise PMC_CMD="$R/tests/tools/kbench $R/bliss-rs_amd/libblissgpu.so pairwise 100000 1" bash tests/tools/pmc_one.sh 2>&1 | grep -v amdgpu.ids > $O/${TAG}_pmc_pairwise.txt; tail -40 $O/${TAG}_pmc_pairwise.txt
SONGS=1024 bash tests/tools/timeline_kbench.sh > $O/${TAG}_timeline_1024songs.txt 2>&1; tail -14 $O/${TAG}_timeline_1024songs.txt
# the single-song front with 4 and 8 seats DIRECTLY behind the counter passes (the one stall on record, round 3, followed them)
if [ "${STRESS_REPS:-0}" -gt 0 ]; then
  SEATS="0,0,0,0 0,0,0,0,0,0,0,0" REPS=$STRESS_REPS bash tests/tools/seats_stress.sh > $O/${TAG}_seats_stress.txt 2>&1
  echo "seat stress: $(grep -c 'rc=0 1' $O/${TAG}_seats_stress.txt) clean runs of $(grep -c '^seats' $O/${TAG}_seats_stress.txt)"; grep -v 'rc=0 1' $O/${TAG}_seats_stress.txt | head
fi
