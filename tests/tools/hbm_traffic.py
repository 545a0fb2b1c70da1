"""Per-kernel HBM traffic from the FETCH_SIZE / WRITE_SIZE passes of tests/tools/hbm_traffic.sh.

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): rocprofv3 reports both
counters in KiB; on gfx950 FETCH_SIZE tallies 128-byte read requests at 64 bytes, so it is doubled.
WRITE_SIZE is used as reported (calibrated below against the spectrogram the STFT kernel must write).
Writes gpurun_out/hbm/hbm_traffic.txt and profiles/hbm_traffic.json (read by bench.py for roofline.traffic).
"""
import collections
import csv
import glob
import json
import os
import sys

root, songs = sys.argv[1], int(sys.argv[2])
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for f in sorted(glob.glob(f"{root}/*/p_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[(f, k)].add(r["Dispatch_Id"])
launches = {}
for (f, k), ids in disp.items():
    launches[k] = max(launches.get(k, 0), len(ids))
rows = {}
lines = [f"# HBM traffic per launch, {songs} songs x 3 969 000 samples; FETCH_SIZE x2 (gfx950), KiB -> bytes",
         f"{'kernel':28s} {'launches':>8s} {'read GB':>10s} {'write GB':>10s} {'total GB':>10s} {'L2 hit':>7s}"]
for k in sorted(agg):
    if not k.startswith("bg::"):
        continue
    n = max(launches.get(k, 1), 1)
    v = agg[k]
    rd = 2.0 * v.get("FETCH_SIZE", 0.0) * 1024.0 / n
    wr = v.get("WRITE_SIZE", 0.0) * 1024.0 / n
    hit, miss = v.get("TCC_HIT_sum", 0.0), v.get("TCC_MISS_sum", 0.0)
    rows[k] = {"launches": n, "read_bytes": rd, "write_bytes": wr, "l2_hit": hit / (hit + miss) if hit + miss else None}
    lines.append(f"{k:28s} {n:8d} {rd / 1e9:10.3f} {wr / 1e9:10.3f} {(rd + wr) / 1e9:10.3f} "
                 f"{(rows[k]['l2_hit'] if rows[k]['l2_hit'] is not None else float('nan')):7.3f}")
txt = "\n".join(lines)
print(txt)
open(os.path.join(root, "hbm_traffic.txt"), "w").write(txt + "\n")
tot = {"kernels": rows, "songs_per_launch": songs}
dom = max((k for k in rows if "synth" not in k and "pairwise" not in k),
          key=lambda k: rows[k]["read_bytes"] + rows[k]["write_bytes"], default=None)
tot["note"] = "kernel = the kernel bench.py reports as dominant by time; bytes_per_launch = 2*FETCH_SIZE + WRITE_SIZE"
name = os.environ.get("DOMINANT", "bg::stft8192_kernel")
if name in rows:
    tot["kernel"] = name.replace("bg::", "")
    tot["bytes_per_launch"] = rows[name]["read_bytes"] + rows[name]["write_bytes"]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
try:  # what the measurement is valid for: bench.py refuses the file when the kernel sources have changed since
    import hashlib

    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "bliss-rs_amd", "csrc")
    h = hashlib.sha256()
    for name in ("kernels_chroma.hip", "kernels_fft512.hip", "kernels_tempo.hip", "kernels_finalize.hip", "fft_r16.hpp",
                 "device_utils.hpp", "internal.hpp"):
        h.update(open(os.path.join(csrc, name), "rb").read())
    tot["kernel_sources_sha256"] = h.hexdigest()
except OSError:
    pass
json.dump(tot, open(os.path.join(root, "hbm_traffic.json"), "w"), indent=1)
