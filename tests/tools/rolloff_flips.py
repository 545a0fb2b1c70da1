#!/usr/bin/env python3
"""Frames whose spectral rolloff bin differs between the GPU and the oracle, on the random musical songs of
musical_check.py (rolloff is the one discontinuous timbral quantity: a bin count).  BLISSGPU_LIB selects a build.
    python tests/tools/rolloff_flips.py [--songs 300] [--seed 2]"""
import argparse
import json
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "tools")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--songs", type=int, default=300)
    ap.add_argument("--seed", type=int, default=2)
    ap.add_argument("--noise", type=int, default=8, help="white-noise songs (60 s) appended to the musical ones")
    args = ap.parse_args()
    import torch

    import bliss_rs_amd as bliss
    import musical_check as M
    import oracle as O

    rng = np.random.default_rng(args.seed)
    songs = [M.make_song(rng)[0] for _ in range(args.songs)] + [O.white_noise(500 + i, 60 * 22050) for i in range(args.noise)]
    lens = np.array([len(s) for s in songs], np.uint64)
    padded = (lens + np.uint64(63)) // np.uint64(64) * np.uint64(64)
    offs = np.zeros(len(songs), np.uint64)
    offs[1:] = np.cumsum(padded)[:-1]
    buf = np.zeros(int(padded.sum()) + 64, np.float32)
    for s, o in zip(songs, offs):
        buf[int(o):int(o) + len(s)] = s
    ctx = bliss.Context(0)
    ctx.analyze(torch.from_numpy(buf).cuda(), offs, lens, 2)
    ctx.synchronize()
    gpu = [ctx.debug_fetch("rolloff", i) for i in range(len(songs))]
    with ThreadPoolExecutor(64) as ex:
        ref = list(ex.map(lambda x: O.SpectralDesc().run(x).series()[1], songs))
    flips = np.array([int((np.abs(g - r) > 1e-3).sum()) for g, r in zip(gpu, ref)])
    frames = np.array([len(r) for r in ref])
    m = args.songs
    print(json.dumps({"lib": os.path.basename(os.environ.get("BLISSGPU_LIB", "libblissgpu.so")), "seed": args.seed,
                      "musical_songs": m, "musical_frames": int(frames[:m].sum()), "musical_flipped_frames": int(flips[:m].sum()),
                      "musical_songs_with_flips": int((flips[:m] > 0).sum()), "musical_worst_song_flips": int(flips[:m].max()),
                      "musical_worst_song": int(flips[:m].argmax()),
                      "noise_frames": int(frames[m:].sum()), "noise_flipped_frames": int(flips[m:].sum())}))


if __name__ == "__main__":
    main()
