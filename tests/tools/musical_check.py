#!/usr/bin/env python3
"""Tool-side parity check on RANDOM MUSICAL signals (seeded): detuned harmonic notes played as sequences at a random
tempo, percussive bursts (straight or swung), a noise floor, a random gain from 1e-3 to 1 -- the places where the tuning
estimate, the filter-bank choice and the beat tracker's lock are decisive rather than near-ties as on white noise.  All
songs go through one GPU call and, one by one, through the CPU oracle (features AND the tuning estimate).  One JSON line.

    python tests/tools/musical_check.py [--songs 300] [--seed 1] [--threads 64]
"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
SR = 22050


def make_song(rng, mods=False, n_samples=None):
    seconds = float(rng.uniform(12.0, 60.0))
    n = int(seconds * SR) + int(rng.integers(0, 2205))
    if n_samples is not None:   # a fixed length (the bench's three-minute songs): the same draws, then the length is overruled
        n = int(n_samples)
        seconds = n / SR
    t = np.arange(n) / SR
    x = np.zeros(n, np.float64)
    cents = float(rng.uniform(-50.0, 50.0))
    a4 = 440.0 * 2.0 ** (cents / 1200.0)
    bpm = float(rng.uniform(60.0, 200.0))
    beat = 60.0 / bpm
    kind = int(rng.integers(0, 4))          # 0 tonal, 1 percussive, 2 both, 3 both over a loud noise floor
    if kind != 1:
        # a melody / chord sequence: every beat (or half beat) a new set of 1-4 notes from a scale on the detuned a4
        scale = np.array([0, 2, 4, 5, 7, 9, 11])
        step = beat * float(rng.choice([0.5, 1.0, 2.0]))
        n_part = int(rng.integers(1, 6))
        pos = 0.0
        while pos < seconds:
            a, b = int(pos * SR), min(n, int((pos + step) * SR))
            if b <= a:
                break
            for _ in range(int(rng.integers(1, 5))):
                semi = int(rng.choice(scale)) + 12 * int(rng.integers(-2, 2))
                f0 = a4 * 2.0 ** (semi / 12.0)
                tt = t[a:b] - t[a]
                env = np.exp(-tt * float(rng.uniform(0.5, 6.0))) * np.minimum(1.0, tt * 200.0)
                for h in range(1, n_part + 1):
                    if f0 * h < 0.45 * SR:
                        x[a:b] += env * np.sin(2 * np.pi * f0 * h * tt) / h * 0.15
            pos += step
    if kind != 0:
        width = float(rng.uniform(0.01, 0.08))
        swing = float(rng.choice([0.0, 0.0, 0.12, 0.2]))
        noise = rng.standard_normal(n) * float(rng.uniform(0.2, 0.7))
        k, t0 = 0, 0.0
        while t0 < seconds - width:
            a = int(t0 * SR)
            b = min(n, a + int(width * SR))
            x[a:b] += noise[a:b] * np.hanning(b - a)
            k += 1
            t0 = k * beat + (swing * beat if k % 2 else 0.0)
    x += rng.standard_normal(n) * (0.05 if kind == 3 else float(rng.choice([0.0, 1e-4, 3e-3])))
    x *= 10.0 ** float(rng.uniform(-3.0, 0.0)) / max(1e-9, np.abs(x).max())
    meta = {"kind": kind, "cents": round(cents, 2), "bpm": round(bpm, 2), "seconds": round(seconds, 2)}
    if mods:
        # what recordings do to the paths white noise never enters: stretches of digital silence (the beat tracker's
        # silence gate, zero-energy frames), a DC offset (zero crossings, energies), a fade, a song barely long enough
        m = []
        if rng.random() < 0.3:
            for _ in range(int(rng.integers(1, 4))):
                a = int(rng.uniform(0.0, max(0.1, seconds - 3.0)) * SR)
                x[a:a + int(rng.uniform(0.3, 3.0) * SR)] = 0.0
            m.append("gaps")
        if rng.random() < 0.2:
            x += float(rng.uniform(-0.05, 0.05)) * np.abs(x).max()
            m.append("dc")
        if rng.random() < 0.2:
            k = int(rng.uniform(1.0, 8.0) * SR)
            x[:k] *= np.linspace(0.0, 1.0, k) ** 2
            x[-k:] *= np.linspace(1.0, 0.0, k) ** 2
            m.append("fade")
        if rng.random() < 0.1:
            x = x[:int(rng.integers(8192, 40000))]
            m.append("short")
        meta["mods"] = "+".join(m)
    return x.astype(np.float32), meta


def _one_of_batch(job):
    seed, i, n_samples = job
    x, meta = make_song(np.random.default_rng([int(seed), int(i)]), False, n_samples)
    return x, meta


def musical_batch(n_songs, n_samples, seed=1, procs=None):
    """n_songs songs of exactly n_samples samples, song i from the generator seeded with (seed, i) -- the same song whatever the
    batch it is part of -- made by a pool of processes (a three-minute song is a few seconds of numpy).  -> (songs, metas)"""
    import multiprocessing as mp

    procs = procs or max(1, min(96, (os.cpu_count() or 2) // 2, n_songs))
    jobs = [(seed, i, n_samples) for i in range(n_songs)]
    if procs == 1:
        res = [_one_of_batch(j) for j in jobs]
    else:
        with mp.get_context("fork").Pool(procs) as pool:
            res = pool.map(_one_of_batch, jobs, chunksize=max(1, n_songs // (4 * procs)))
    return [r[0] for r in res], [r[1] for r in res]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--songs", type=int, default=300)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--threads", type=int, default=64)
    ap.add_argument("--flux-order", type=int, default=0, help="BLISSGPU_OPT_FLUX_ORDER")
    ap.add_argument("--mods", action="store_true", help="silent gaps, DC offsets, fades, barely-long-enough songs on top")
    args = ap.parse_args()
    import torch

    import bliss_rs_amd as bliss
    import oracle as O

    rng = np.random.default_rng(args.seed)
    songs, meta = zip(*(make_song(rng, args.mods) for _ in range(args.songs)))
    lens = np.array([len(s) for s in songs], np.uint64)
    padded = (lens + np.uint64(63)) // np.uint64(64) * np.uint64(64)
    offs = np.zeros(len(songs), np.uint64)
    offs[1:] = np.cumsum(padded)[:-1]
    buf = np.zeros(int(padded.sum()) + 64, np.float32)
    for s, o in zip(songs, offs):
        buf[int(o):int(o) + len(s)] = s
    ctx = bliss.Context(0)
    ctx.set_option("flux_order", args.flux_order)
    out, status = ctx.analyze(torch.from_numpy(buf).cuda(), offs, lens, 2)
    ctx.synchronize()
    got = out.cpu().numpy()
    tuning, n_bpms = ctx.last_tuning(len(songs))
    assert (status.cpu().numpy() == 0).all()
    t0 = time.perf_counter()

    def one(x):
        return O.song_analyze(x, 2), O.chroma_desc(x)[1]

    with ThreadPoolExecutor(args.threads) as ex:
        res = list(ex.map(one, songs))
    ref = np.stack([r[0] for r in res])
    otuning = np.array([r[1] for r in res])
    # the oracle's distance from ITSELF when only its FFT precision changes (f32 -> f64 arithmetic, rounded to f32): what
    # no f32 implementation can be held under.  (A global switch of the oracle library: set for the whole pass.)
    O.set_fft_double(True)
    with ThreadPoolExecutor(args.threads) as ex:
        ref64 = np.stack(list(ex.map(lambda x: O.song_analyze(x, 2), songs)))
    O.set_fft_double(False)
    floor = np.abs(ref.astype(np.float64) - ref64)
    # a tuning that differs is a near-tie of the pitch histogram's argmax if the oracle's own answer moves with its FFT precision
    O.set_fft_double(True)
    otuning64 = {int(i): float(O.chroma_desc(songs[i])[1]) for i in np.nonzero(np.abs(tuning - otuning) > 1e-12)[0]}
    O.set_fft_double(False)
    err = np.abs(got.astype(np.float64) - ref)
    n_t = (lens.astype(np.int64) - 512) // 128 + 1
    flip = 2.0 * (22050.0 / 512.0) / 11025.0 / n_t
    tol = np.full(err.shape, 1e-5)
    tol[:, 4] += 2 * flip
    tol[:, 5] += 2 * flip * np.sqrt(np.maximum(n_t, 1)) * 0.5
    over = np.nonzero((err[:, 1:] > tol[:, 1:]).any(axis=1))[0]
    beyond = np.nonzero(((err[:, 1:] > tol[:, 1:]) & (err[:, 1:] > floor[:, 1:])).any(axis=1))[0]
    feat_over = sorted({int(j) + 1 for i in over for j in np.nonzero(err[i, 1:] > tol[i, 1:])[0]})
    tun_bad = np.nonzero(np.abs(tuning - otuning) > 1e-12)[0]
    print(json.dumps({
        "seed": args.seed, "mods": bool(args.mods), "songs": len(songs), "seconds_of_audio": round(float(lens.sum()) / SR, 1),
        "oracle_seconds": round(time.perf_counter() - t0, 1),
        "kinds": {str(k): int(sum(m["kind"] == k for m in meta)) for k in range(4)},
        "distinct_tunings": int(len(set(np.round(otuning, 2)))),
        "tuning_mismatches": [{"song": int(i), **meta[i], "gpu": float(tuning[i]), "oracle": float(otuning[i]),
                               "oracle_with_f64_ffts": otuning64[int(i)]} for i in tun_bad][:10],
        "max_abs_err_non_tempo": float(err[:, 1:].max()),
        "songs_over_1e-5_non_tempo": int(len(over)), "features_over": feat_over,
        "songs_over_1e-5_AND_over_the_oracle_f32_vs_f64_floor": [{"song": int(i), **meta[i], "err": [float(v) for v in err[i]],
                                                                 "floor": [float(v) for v in floor[i]]} for i in beyond][:10],
        "over_tolerance_examples": [{"song": int(i), **meta[i], "worst_feature": int(err[i, 1:].argmax()) + 1,
                                     "err": float(err[i, 1:].max()), "oracle_f32_vs_f64": float(floor[i, int(err[i, 1:].argmax()) + 1])}
                                    for i in over][:6],
        "oracle_f32_vs_f64_max_non_tempo": float(floor[:, 1:].max()),
        "oracle_f32_vs_f64_tempo_over_1e-5": int((floor[:, 0] > 1e-5).sum()),
        "tempo_over_1e-5": int((err[:, 0] > 1e-5).sum()), "tempo_over_3e-5": int((err[:, 0] > 3e-5).sum()),
        "tempo_over_1e-4": int((err[:, 0] > 1e-4).sum()), "max_abs_err_tempo": float(err[:, 0].max()),
        "tempo_fraction_within_1e-5": round(float((err[:, 0] <= 1e-5).mean()), 6),
        "tempo_mismatch_songs": [{"song": int(i), **meta[i], "gpu": float(got[i, 0]), "oracle": float(ref[i, 0])}
                                 for i in np.nonzero(err[:, 0] > 1e-4)[0]][:10],
        "distinct_tempo_values": int(len(set(np.round(ref[:, 0], 4))))}))


if __name__ == "__main__":
    main()
