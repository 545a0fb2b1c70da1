"""VALU-busy fraction per kernel from the pass of tests/tools/valu_busy.sh.

SQ_ACTIVE_INST_VALU counts, summed over the chip's 1024 SIMDs, the quad-cycles in which a SIMD has a VALU instruction in flight
(the SQ counters of gfx950 tick once per four shader cycles: SQ_WAVE_CYCLES of a kernel with W resident wavefronts per SIMD is
W x 1024 x cycles / 4 -- checked below).  Busy fraction = 4 x SQ_ACTIVE_INST_VALU / (1024 x shader cycles of the dispatch); the
shader cycles come from GRBM_GUI_ACTIVE (summed over the 8 XCDs by rocprofv3: / 8) and, as a cross-check, from the dispatch
duration x the clock that implies.  Writes valu_busy.txt and valu_busy.json (kernel_sources_sha256 = what it is valid for;
bench.py puts `frac` of the dominant kernel into the line as roofline_valu and refuses the file when the sources changed)."""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

root = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(float)
seen = set()
for f in sorted(glob.glob(f"{root}/*/p_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0].replace("bg::", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (k, r["Dispatch_Id"]) not in seen:
            seen.add((k, r["Dispatch_Id"]))
            dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e9
N_SIMD, N_XCD = 1024, 8
rows, lines = {}, ["# VALU-busy = 4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x shader cycles); cycles = GRBM_GUI_ACTIVE / 8 XCDs",
                   f"{'kernel':20s} {'ms (profiled)':>14s} {'cycles':>14s} {'GHz':>6s} {'VALU busy':>10s} {'waves/SIMD':>11s} {'SQ busy':>8s}"]
for k, v in sorted(agg.items(), key=lambda kv: -dur[kv[0]]):
    cycles = v.get("GRBM_GUI_ACTIVE", 0.0) / N_XCD
    if cycles <= 0 or dur[k] <= 0:
        continue
    ghz = cycles / dur[k] / 1e9
    busy = 4.0 * v.get("SQ_ACTIVE_INST_VALU", 0.0) / (N_SIMD * cycles)
    resident = 4.0 * v.get("SQ_WAVE_CYCLES", 0.0) / (N_SIMD * cycles)
    sq_busy = v.get("SQ_BUSY_CYCLES", 0.0)
    rows[k] = {"valu_busy": round(busy, 4), "implied_clock_GHz": round(ghz, 3), "mean_resident_waves_per_simd": round(resident, 2),
               "profiled_ms": round(dur[k] * 1e3, 3)}
    lines.append(f"{k:20s} {dur[k] * 1e3:14.3f} {cycles:14.0f} {ghz:6.2f} {busy:10.3f} {resident:11.2f} {sq_busy / max(1.0, cycles * N_XCD):8.2f}")
txt = "\n".join(lines)
print(txt)
open(os.path.join(root, "valu_busy.txt"), "w").write(txt + "\n")
h = hashlib.sha256()
for name in ("kernels_chroma.hip", "kernels_fft512.hip", "kernels_tempo.hip", "kernels_finalize.hip", "fft_r16.hpp", "device_utils.hpp", "internal.hpp"):
    h.update(open(os.path.join(ROOT, "bliss-rs_amd", "csrc", name), "rb").read())
json.dump({"kernels": rows, "kernel_sources_sha256": h.hexdigest(),
           "note": "4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE / 8), serial 128-song step of tests/tools/kbench, rocprofv3 --pmc (kernel-trace only)"},
          open(os.path.join(root, "valu_busy.json"), "w"), indent=1)
