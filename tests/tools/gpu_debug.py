"""Developer aid (GPU box): per-feature GPU-vs-oracle errors on a handful of songs + per-kernel times."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import torch  # noqa: E402

import bliss_rs_amd as bliss  # noqa: E402
import oracle as O  # noqa: E402

np.set_printoptions(linewidth=200, precision=7, suppress=True)
names = [m.name for m in bliss.AnalysisIndex]


def main():
    n3 = 3969000
    golden = (np.load(os.path.join(ROOT, "tests/golden/s16_mono_22_5kHz.pcm_s16.npy")).astype(np.float32) / np.float32(32768)).astype(np.float32)
    click = np.tile(np.concatenate([np.zeros(22000, np.float32), np.ones(100, np.float32)]), 30)
    songs = {
        "golden": golden,
        "noise3min_0": O.white_noise(0, n3),
        "noise3min_1": O.white_noise(1, n3),
        "noise_30s": O.white_noise(2, 30 * 22050 + 17),
        "silence": np.zeros(100000, np.float32),
        "click60": click,
        "min8192": O.white_noise(3, 8192),
        "sine440": (0.5 * np.sin(2 * np.pi * 440.0 * np.arange(10 * 22050) / 22050.0)).astype(np.float32),
    }
    ctx = bliss.Context(0)
    keys = list(songs)
    lens = [len(songs[k]) for k in keys]
    offs = np.concatenate([[0], np.cumsum([(l + 63) // 64 * 64 for l in lens])[:-1]]).astype(np.uint64)
    buf = np.zeros(int(offs[-1]) + lens[-1] + 64, np.float32)
    for k, o in zip(keys, offs):
        buf[int(o):int(o) + len(songs[k])] = songs[k]
    pcm = torch.from_numpy(buf).cuda()
    ctx.profile_enable(True)
    for version in (2, 1):
        t0 = time.time()
        out, status = ctx.analyze(pcm, offs, lens, version)
        ctx.synchronize()
        print(f"version {version}: gpu wall {time.time() - t0:.3f}s status {status.cpu().tolist()}")
        tun, nb = ctx.last_tuning(len(keys))
        g = out.cpu().numpy()
        for i, k in enumerate(keys):
            t0 = time.time()
            ref = O.song_analyze(songs[k], version)
            _, otun = O.chroma_desc(songs[k])
            err = np.abs(g[i] - ref)
            print(f"--- {k} (n={lens[i]}) oracle {time.time() - t0:.2f}s  max err {err.max():.3g} at {names[int(err.argmax())]}"
                  f"  tuning gpu {tun[i]:+.2f} oracle {otun:+.2f}  n_bpms {nb[i]}")
            print("   gpu   ", g[i])
            print("   oracle", ref)
            print("   err   ", err)
    print("profile (ms, launches):")
    for k, v in ctx.profile().items():
        print(f"   {k:22s} {v[0]:10.3f} {v[1]}")
    # distances
    rng = np.random.default_rng(0)
    A = rng.uniform(-1, 1, (300, 23)).astype(np.float32)
    B = rng.uniform(-1, 1, (517, 23)).astype(np.float32)
    W = O.feature_weights(2)
    M = rng.uniform(0, 1, (23, 23)).astype(np.float32)
    for metric, m in (("euclidean", None), ("cosine", None), ("mahalanobis", W), ("mahalanobis", M)):
        got = bliss.playlist.pairwise_distances(A, B, metric, m)
        ref = O.pairwise(A, B, metric, m)
        print(metric, "diag" if m is W else "", "bit-exact:", np.array_equal(got, ref), "max diff", np.abs(got - ref).max())
    A20, B20 = A[:, :20].copy(), B[:, :20].copy()
    print("d=20 euclid bit-exact:", np.array_equal(bliss.playlist.pairwise_distances(A20, B20), O.pairwise(A20, B20)))
    A7, B7 = A[:, :7].copy(), B[:, :7].copy()
    print("d=7 generic bit-exact:", np.array_equal(bliss.playlist.pairwise_distances(A7, B7), O.pairwise(A7, B7)))
    # synth generator parity
    t = torch.empty(10007, dtype=torch.float32, device="cuda")
    ctx.synth_white_noise(t, [0], [10007], 5)
    print("synth bit-exact:", np.array_equal(t.cpu().numpy(), O.white_noise(5, 10007)))


if __name__ == "__main__":
    main()
