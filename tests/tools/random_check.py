#!/usr/bin/env python3
"""Tool-side parity check of RAGGED batches: songs of random lengths (seeded; 8192 samples .. 4 minutes, a few of
awkward lengths around the framing boundaries) analysed in one GPU call -- several chunks when a small workspace limit is
given -- against the CPU oracle.  Prints one JSON line per seed.

    python tests/tools/random_check.py [--songs 768] [--seeds 1 2 3] [--ws-limit-gb 4]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--songs", type=int, default=768)
    ap.add_argument("--seeds", type=int, nargs="+", default=[1, 2, 3])
    ap.add_argument("--threads", type=int, default=64)
    ap.add_argument("--flux-order", type=int, default=0, help="BLISSGPU_OPT_FLUX_ORDER")
    ap.add_argument("--ws-limit-gb", type=float, default=4.0, help="small slots force a multi-chunk pipeline")
    args = ap.parse_args()
    import torch

    import bliss_rs_amd as bliss
    import oracle as O

    ctx = bliss.Context(0)
    ctx.set_option("flux_order", args.flux_order)
    ctx.set_workspace_limit(int(args.ws_limit_gb * (1 << 30)))
    for seed in args.seeds:
        rng = np.random.default_rng(seed)
        n = args.songs
        lens = rng.integers(8192, 4 * 60 * 22050, n).astype(np.uint64)
        awkward = [8192, 8193, 8192 + 2205, 2205 * 40, 2205 * 40 + 1, 2205 * 40 - 1, 512 + 256 * 127, 512 + 256 * 128, 32768 + 5,
                   128 * 1000 + 511, 128 * 1000 + 512, 1024 * 77, 1024 * 77 + 1]
        lens[:len(awkward)] = awkward
        rng.shuffle(lens)
        padded = (lens + np.uint64(63)) // np.uint64(64) * np.uint64(64)
        offs = np.zeros(n, np.uint64)
        offs[1:] = np.cumsum(padded)[:-1]
        pcm = torch.empty(int(padded.sum()) + 64, dtype=torch.float32, device="cuda")
        ctx.synth_white_noise(pcm, offs, lens, first_song_index=100000 * seed)
        out, status = ctx.analyze(pcm, offs, lens, 2)
        ctx.synchronize()
        got = out.cpu().numpy()
        chunks = ctx.last_chunks()
        t0 = time.perf_counter()
        host = pcm.cpu().numpy()
        ref, st = O.song_analyze_batch(host, offs, lens, 2, min(args.threads, n))
        assert (st == 0).all() and (status.cpu().numpy() == 0).all()
        err = np.abs(got - ref)
        # the tests' tolerance: 1e-5, plus the explicit allowance of two flipped rolloff bins (tests/test_gpu_parity.py)
        n_t = (lens.astype(np.int64) - 512) // 128 + 1
        flip = 2.0 * (22050.0 / 512.0) / 11025.0 / n_t
        tol = np.full(err.shape, 1e-5)
        tol[:, 4] += 2 * flip
        tol[:, 5] += 2 * flip * np.sqrt(np.maximum(n_t, 1)) * 0.5
        over = (err[:, 1:] > tol[:, 1:])
        tm = np.nonzero(err[:, 0] > 1e-4)[0]
        detail = {"over_tolerance": [(int(lens[i]), int(j) + 1, float(err[i, j + 1])) for i, j in zip(*np.nonzero(over))][:8],
                  "rolloff_flip_songs": [(int(lens[i]), float(err[i, 4]), float(err[i, 5])) for i in np.nonzero(err[:, 4] > 1e-5)[0]][:8],
                  "tempo_mismatch_songs": [(int(lens[i]), float(got[i, 0]), float(ref[i, 0])) for i in tm][:8]}
        worst = int(err[:, 1:].max(axis=1).argmax())
        print(json.dumps({"seed": seed, "songs": n, "chunks": int(chunks), "samples": int(lens.sum()),
                          "oracle_seconds": round(time.perf_counter() - t0, 1),
                          "max_abs_err_non_tempo": float(err[:, 1:].max()), "worst_song_len": int(lens[worst]),
                          "songs_over_1e-5_non_tempo": int((err[:, 1:].max(axis=1) > 1e-5).sum()),
                          "tempo_mismatches_over_1e-4": int((err[:, 0] > 1e-4).sum()),
                          "tempo_over_1e-5": int((err[:, 0] > 1e-5).sum()), "tempo_over_3e-5": int((err[:, 0] > 3e-5).sum()),
                          "tempo_fraction_within_1e-5": round(float((err[:, 0] <= 1e-5).mean()), 6),
                          "max_abs_err_tempo": float(err[:, 0].max()), **detail}))


if __name__ == "__main__":
    main()
