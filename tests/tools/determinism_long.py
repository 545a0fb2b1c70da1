#!/usr/bin/env python3
"""Where does a run-to-run difference come from?  ONE chunk of long songs analysed REPS times; per run the feature rows,
the tuning estimates and the pitch histograms (tuning pass 2's output) of every song are compared with the first run.

    python tests/tools/determinism_long.py [--songs 160] [--reps 40]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--songs", type=int, default=160)
    ap.add_argument("--reps", type=int, default=40)
    ap.add_argument("--seed", type=int, default=5)
    args = ap.parse_args()
    import torch

    import bliss_rs_amd as bliss

    ctx = bliss.Context(0)
    rng = np.random.default_rng(args.seed)
    n = args.songs
    lens = rng.integers(4_000_000, 5_300_000, n).astype(np.uint64)
    padded = (lens + np.uint64(63)) // np.uint64(64) * np.uint64(64)
    offs = np.zeros(n, np.uint64)
    offs[1:] = np.cumsum(padded)[:-1]
    pcm = torch.empty(int(padded.sum()) + 64, dtype=torch.float32, device="cuda")
    ctx.synth_white_noise(pcm, offs, lens, first_song_index=7000)
    ref = None
    events = []
    for rep in range(args.reps):
        out, status = ctx.analyze(pcm, offs, lens, 2)
        ctx.synchronize()
        assert ctx.last_chunks() == 1
        rows = out.cpu().numpy().copy()
        tuning, n_bpms = ctx.last_tuning(n)
        hist = np.stack([ctx.debug_fetch("pitch_hist", i) for i in range(n)])
        cur = (rows, tuning.copy(), hist)
        if ref is None:
            ref = cur
            continue
        for i in np.nonzero((rows != ref[0]).any(axis=1) | (tuning != ref[1]) | (hist != ref[2]).any(axis=1))[0]:
            dh = hist[i].astype(np.int64) - ref[2][i].astype(np.int64)
            events.append({"rep": rep, "song": int(i), "len": int(lens[i]), "row_max_abs_diff": float(np.abs(rows[i] - ref[0][i]).max()),
                           "tuning": [float(ref[1][i]), float(tuning[i])], "hist_total": [int(ref[2][i].sum()), int(hist[i].sum())],
                           "hist_bins_changed": int((dh != 0).sum()), "hist_delta_abs_sum": int(np.abs(dh).sum())})
    print(json.dumps({"songs": n, "reps": args.reps, "events": events[:10], "n_events": len(events)}))


if __name__ == "__main__":
    main()
