#!/usr/bin/env python3
"""Per-kernel times of one 1024-song step with the kernels one after another (BLISSGPU_OPT_SERIAL = 1: nothing overlaps, so a
kernel's time is its own): the minimum and the median over the steps, HIP events on the stream each kernel runs on.

    python tests/tools/kernel_times.py [songs=1024] [steps=6] [--write profiles/kernel_times_serial.json]

tests/test_gpu_round6.py::test_the_three_big_kernels_are_not_slower_than_recorded reads the file: a kernel more than 6 % above
its recorded minimum fails the suite (the per-stage table the reference's criterion harness prints,
benches/analysis_pipeline.rs:8-126, as a regression guard)."""
import json
import os
import statistics
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]

BIG = ("stft8192_kernel", "fft512_kernel", "chroma_kernel")


def measure(songs=1024, steps=6, N=3969000):
    import torch

    import bliss_rs_amd as bliss

    ctx = bliss.Context(0)
    ctx.set_option("serial", 1)
    lens = np.full(songs, N, np.uint64)
    offs = np.arange(songs, dtype=np.uint64) * np.uint64((N + 63) // 64 * 64)
    pcm = torch.empty(int(offs[-1]) + N + 64, dtype=torch.float32, device="cuda")
    ctx.synth_white_noise(pcm, offs, lens, first_song_index=0)
    out = torch.empty((songs, 23), dtype=torch.float32, device="cuda")
    status = torch.empty((songs,), dtype=torch.int32, device="cuda")
    ctx.analyze(pcm, offs, lens, 2, out=out, status=status)
    ctx.synchronize()
    per = {}
    ctx.profile_enable(True)
    for _ in range(steps):
        ctx.profile_reset()
        ctx.analyze(pcm, offs, lens, 2, out=out, status=status)
        ctx.synchronize()
        for k, (ms, launches) in ctx.profile().items():
            per.setdefault(k, []).append(ms)
    ctx.profile_enable(False)
    ctx.close()
    return {k: {"min_ms": round(min(v), 4), "median_ms": round(statistics.median(v), 4)} for k, v in per.items() if max(v) > 0}


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    songs = int(args[0]) if args else 1024
    steps = int(args[1]) if len(args) > 1 else 6
    t = measure(songs, steps)
    doc = {"songs": songs, "samples_per_song": 3969000, "steps": steps, "mode": "BLISSGPU_OPT_SERIAL = 1 (kernels one after another)",
           "kernels": t}
    for k in sorted(t, key=lambda k: -t[k]["min_ms"]):
        print(f"{k:24s} min {t[k]['min_ms']:8.3f} ms   median {t[k]['median_ms']:8.3f} ms")
    if "--write" in sys.argv:
        path = sys.argv[sys.argv.index("--write") + 1]
        json.dump(doc, open(path, "w"), indent=1)
        print("wrote", path)


if __name__ == "__main__":
    main()
