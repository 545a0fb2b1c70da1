#!/usr/bin/env python3
"""Run-to-run determinism of a ragged multi-chunk batch: the same random-length songs analysed REPS times on the GPU, every
run compared bit for bit with the first.  Prints one JSON line.

    python tests/tools/determinism_check.py [--songs 768] [--seed 3] [--reps 8] [--ws-limit-gb 4]
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
    ap.add_argument("--songs", type=int, default=768)
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--reps", type=int, default=8)
    ap.add_argument("--ws-limit-gb", type=float, default=4.0)
    ap.add_argument("--max-seconds", type=float, default=240.0, help="longest song")
    args = ap.parse_args()
    import torch

    import bliss_rs_amd as bliss

    ctx = bliss.Context(0)
    ctx.set_workspace_limit(int(args.ws_limit_gb * (1 << 30)))
    rng = np.random.default_rng(args.seed)
    n = args.songs
    lens = rng.integers(8192, int(args.max_seconds * 22050), n).astype(np.uint64)
    padded = (lens + np.uint64(63)) // np.uint64(64) * np.uint64(64)
    offs = np.zeros(n, np.uint64)
    offs[1:] = np.cumsum(padded)[:-1]
    pcm = torch.empty(int(padded.sum()) + 64, dtype=torch.float32, device="cuda")
    ctx.synth_white_noise(pcm, offs, lens, first_song_index=100000 * args.seed)
    first, diffs = None, []
    for rep in range(args.reps):
        out, status = ctx.analyze(pcm, offs, lens, 2)
        ctx.synchronize()
        got = out.cpu().numpy().copy()
        tuning, _ = ctx.last_tuning(n)
        tuning = tuning.copy()
        if first is None:
            first, first_tuning = got, tuning
            continue
        bad = np.nonzero((got != first).any(axis=1))[0]
        for i in bad:
            cols = np.nonzero(got[i] != first[i])[0]
            diffs.append({"rep": rep, "song": int(i), "song_len": int(lens[i]), "features": [int(c) for c in cols],
                          "max_abs_diff": float(np.abs(got[i] - first[i]).max()),
                          "tuning": [float(first_tuning[i]), float(tuning[i])]})
    print(json.dumps({"songs": n, "reps": args.reps, "chunks": int(ctx.last_chunks()), "rows_that_differ": len(diffs), "diffs": diffs[:12]}))


if __name__ == "__main__":
    main()
