#!/usr/bin/env python3
"""Why does a frame's rolloff bin differ from the oracle's?  For every differing frame of one song of musical_check's
generator: both bins, and how far the oracle's own sequential running energy is from the 95 % threshold at those bins,
relative to the frame's energy -- a margin at the level of FFT rounding (1e-7) means the inputs, not the order, decide.
    python tests/tools/rolloff_why.py <seed> <song index> [mods-every-other: 0|1]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "tools")]


def main():
    import torch

    import bliss_rs_amd as bliss
    import musical_check as M
    import oracle as O

    seed, idx, alt = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 0
    rng = np.random.default_rng(seed)
    songs = [M.make_song(rng, mods=(alt and i % 2 == 1))[0] for i in range(idx + 1)]
    x = songs[idx]
    c = bliss.Context(0)
    lens = np.array([len(x)], np.uint64)
    offs = np.array([0], np.uint64)
    buf = np.zeros(len(x) + 64, np.float32)
    buf[:len(x)] = x
    c.analyze(torch.from_numpy(buf).cuda(), offs, lens, 2)
    c.synchronize()
    g = c.debug_fetch("rolloff", 0)
    r = O.SpectralDesc().run(x).series()[1]
    d = np.flatnonzero(np.abs(g - r) > 1e-3)
    print(f"song {idx} of seed {seed}: {len(x)} samples, {len(r)} frames, {len(d)} differ")
    for k in d[:12]:
        a = O.pvoc512_norms(x[128 * k:128 * k + 512])[0]
        sq = (a * a).astype(np.float32)
        S = np.zeros(256, np.float32)
        run = np.float32(0)
        for j in range(256):
            run = np.float32(run + sq[j])
            S[j] = run
        thr = np.float32(S[255] * np.float32(0.95))
        gb, ob = int(round(g[k] * 512 / 22050)), int(round(r[k] * 512 / 22050))
        lo, hi = min(gb, ob), max(gb, ob)
        marg = [(int(j), float((S[j] - thr) / S[255])) for j in range(max(lo - 2, 0), min(hi + 1, 256))]
        print(f" frame {k}: gpu bin {gb}, oracle bin {ob}; (S_j - thr) / total around them: {marg[:6]}")


if __name__ == "__main__":
    main()
