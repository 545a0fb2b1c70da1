#!/bin/bash
# Kernel timeline of one analysis step driven by tests/tools/kbench (no Python in the trace):
#   [KBENCH_* options] SONGS=1024 bash tests/tools/timeline_kbench.sh [lib]
R=$PWD; O=$R/gpurun_out/tl; rm -rf $O; mkdir -p $O
L=${1:-$R/bliss-rs_amd/libblissgpu.so}
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o tl -- $R/tests/tools/kbench $L ${SONGS:-1024} 180 1 > $O/trace.log 2>&1; echo "trace rc=$?"
cd $R
python - $O/trace <<'PY' | tee $O/timeline.txt
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        # "void bg::fft512_kernel<false>(float const*, ...)" -> "fft512_kernel"
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("bg::", "").split("<")[0]
        rows.append((name[:26], int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort(key=lambda r: r[1])
# the last step = from the last launch of the first big kernel of a step
firsts = [i for i, r in enumerate(rows) if r[0] in ("fft512_kernel", "stft8192_kernel")]
step_starts = [i for k, i in enumerate(firsts) if k == 0 or i - firsts[k - 1] > 3]
i0 = step_starts[-2] if len(step_starts) >= 2 else step_starts[-1]   # the plain (unprofiled) step
i1 = step_starts[-1] if len(step_starts) >= 2 else len(rows)
t0 = rows[i0][1]
for n, s, e in rows[i0:i1]:
    print(f"{n:26s} {(s - t0) / 1e6:9.3f} -> {(e - t0) / 1e6:9.3f}   ({(e - s) / 1e6:7.3f} ms)")
PY
rm -rf $O/trace
