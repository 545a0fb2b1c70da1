#!/bin/bash
# Average shader clock DURING each kernel: GRBM_GUI_ACTIVE (busy cycles of the graphics engine) / the kernel's duration, from one
# rocprofv3 --pmc pass of kbench with every kernel alone (KBENCH_SERIAL=1).  Which kernels does the power management slow down?
#   SONGS=512 bash tests/tools/kernel_clocks.sh
R=$PWD; cd /tmp; export TMPDIR=/tmp; export KBENCH_SERIAL=1
[ -x $R/tests/tools/kbench ] || g++ -std=c++17 -O1 -o $R/tests/tools/kbench $R/tests/tools/kbench.cpp -ldl
O=$R/gpurun_out/clk; rm -rf $O
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/a -o p -- $R/tests/tools/kbench $R/bliss-rs_amd/libblissgpu.so ${SONGS:-512} 180 2 > $O.log 2>&1; echo "analysis rc=$?"
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/b -o p -- $R/tests/tools/kbench $R/bliss-rs_amd/libblissgpu.so pairwise 100000 2 >> $O.log 2>&1; echo "pairwise rc=$?"
cd $R
python - $O <<'PY'
import csv, glob, sys, collections
dur = {}
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[(f.split("/")[-3], r["Dispatch_Id"])] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
acc = collections.defaultdict(lambda: [0.0, 0.0, 0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
            continue
        key = (f.split("/")[-3], r["Dispatch_Id"])
        if key not in dur:
            continue
        name, ns = dur[key]
        name = name.split("(")[0].replace("void ", "").replace("bg::", "")
        a = acc[name]
        a[0] += float(r["Counter_Value"]); a[1] += ns; a[2] += 1
print(f"{'kernel':44s} {'launches':>8s} {'ms':>9s} {'MHz (busy cycles / duration)':>30s}")
for name, (cyc, ns, n) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    if ns > 2e5:
        print(f"{name[:44]:44s} {n:8d} {ns / 1e6:9.3f} {cyc / ns * 1e3:30.0f}")
PY
