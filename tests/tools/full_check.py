#!/usr/bin/env python3
"""Tool-side (not bench) parity check of EVERY song of the default bench batch: the 1024 synthetic 3-minute songs of
configs[1] analysed on the GPU against the CPU oracle (all host cores, ~40 s).  Prints one JSON line.

    python tests/tools/full_check.py [--songs 1024] [--threads 64] [--noise-floor]

Tempo is reported as a per-song error HISTOGRAM (counts over 1e-5 -- the reference's tolerance, src/song/mod.rs:582-590 --
3e-5 and 1e-4, every song over 1e-5 listed).  --noise-floor repeats the oracle with its FFTs in f64 (bo_set_fft_double)
and reports the same histogram for oracle(f32 FFT) against oracle(f64 FFT): what FFT rounding alone does to the tempo
value on these songs, i.e. the floor no f32 implementation (rustfft included) can be held under.
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
    ap.add_argument("--songs", type=int, default=1024)
    ap.add_argument("--samples", type=int, default=3969000)
    ap.add_argument("--threads", type=int, default=64)
    ap.add_argument("--noise-floor", action="store_true")
    ap.add_argument("--flux-order", type=int, default=0, help="BLISSGPU_OPT_FLUX_ORDER: 1 = SpecFlux summed in the reference's bin order")
    ap.add_argument("--musical", action="store_true", help="the batch of `bench.py --config musical` (seeded musical signals) instead of white noise")
    ap.add_argument("--seed", type=int, default=20260927, help="--musical: bench.py's default seed")
    args = ap.parse_args()
    import torch

    import bliss_rs_amd as bliss
    import oracle as O

    n, N = args.songs, args.samples
    ctx = bliss.Context(0)
    ctx.set_option("flux_order", args.flux_order)
    offs = np.arange(n, dtype=np.uint64) * np.uint64(N)
    lens = np.full(n, N, np.uint64)
    pcm = torch.empty(n * N, dtype=torch.float32, device="cuda")
    meta = None
    if args.musical:
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from musical_check import musical_batch

        songs, meta = musical_batch(n, N, seed=args.seed)
        for i, x in enumerate(songs):
            pcm[i * N:(i + 1) * N] = torch.from_numpy(x).cuda()
        del songs
    else:
        ctx.synth_white_noise(pcm, offs, lens, first_song_index=0)
    out, status = ctx.analyze(pcm, offs, lens, 2)
    ctx.synchronize()
    got = out.cpu().numpy()
    tuning, _ = ctx.last_tuning(n)
    t0 = time.perf_counter()

    def oracle_rows():
        ref = np.empty((n, 23), np.float32)
        step = 128  # bounded host memory: 128 songs = 2 GB of PCM at a time
        for i0 in range(0, n, step):
            k = min(step, n - i0)
            host = pcm[i0 * N:(i0 + k) * N].cpu().numpy()
            r, st = O.song_analyze_batch(host, np.arange(k, dtype=np.uint64) * np.uint64(N), np.full(k, N, np.uint64), 2,
                                         min(args.threads, k))
            assert (st == 0).all()
            ref[i0:i0 + k] = r
        return ref

    def tempo_histogram(e):
        over = np.flatnonzero(e > 1e-5)
        return {"songs": int(len(e)), "over_1e-5": int((e > 1e-5).sum()), "over_3e-5": int((e > 3e-5).sum()),
                "over_1e-4": int((e > 1e-4).sum()), "max": float(e.max()),
                "fraction_within_1e-5": round(float((e <= 1e-5).mean()), 6),
                "songs_over_1e-5": [{"song": int(i), "abs_err": float(e[i])} for i in over[:64]]}

    ref = oracle_rows()
    err = np.abs(got - ref)
    noise = None
    if args.noise_floor:
        O.set_fft_double(True)
        ref64 = oracle_rows()
        O.set_fft_double(False)
        e64 = np.abs(ref.astype(np.float64) - ref64.astype(np.float64))
        g64 = np.abs(got.astype(np.float64) - ref64.astype(np.float64))

        def side(a, b):   # per-song distances to the f64-FFT oracle: the GPU's against the f32-FFT oracle's own
            q = lambda v: {"median": float(np.median(v)), "p90": float(np.percentile(v, 90)), "p99": float(np.percentile(v, 99)),
                           "max": float(v.max())}
            return {"gpu_to_f64_oracle": q(a), "f32_oracle_to_f64_oracle": q(b),
                    "songs_where_gpu_is_closer": int((a < b).sum()), "songs_where_f32_oracle_is_closer": int((b < a).sum()),
                    "gpu_median_at_or_below_f32_oracle_median": bool(np.median(a) <= np.median(b))}

        noise = {"what": "oracle with f32 FFTs against the same oracle with f64 FFTs (bo_set_fft_double), same songs",
                 "tempo": tempo_histogram(e64[:, 0]), "max_abs_err_non_tempo": float(e64[:, 1:].max()),
                 "gpu_vs_f64_oracle_tempo": tempo_histogram(g64[:, 0]),
                 "which_side_of_the_floor": {"tempo": side(g64[:, 0], e64[:, 0]), "flatness_mean": side(g64[:, 6], e64[:, 6]),
                                             "flatness_std": side(g64[:, 7], e64[:, 7]),
                                             "all_22_non_tempo_features_max": side(g64[:, 1:].max(axis=1), e64[:, 1:].max(axis=1))}}
    musical = None
    if args.musical:
        # the tuning estimate is the decision music exercises (white noise lands on one tuning): device tuning against the oracle's
        from concurrent.futures import ThreadPoolExecutor

        def otun(i):
            return float(O.chroma_desc(pcm[i * N:(i + 1) * N].cpu().numpy())[1])

        with ThreadPoolExecutor(args.threads) as ex:
            otuning = np.array(list(ex.map(otun, range(n))))
        bad = np.flatnonzero(np.abs(tuning - otuning) > 1e-12)
        musical = {"seed": args.seed, "kinds": {str(k): int(sum(m["kind"] == k for m in meta)) for k in range(4)},
                   "distinct_tunings": int(len(set(np.round(otuning, 2)))), "tuning_mismatches": int(len(bad)),
                   "tuning_mismatch_songs": [{"song": int(i), **meta[i], "gpu": float(tuning[i]), "oracle": float(otuning[i])} for i in bad[:10]],
                   "distinct_tempo_values": int(len(set(np.round(ref[:, 0], 4)))),
                   "songs_without_a_beat": int((ref[:, 0] == -1.0).sum())}
    res = {"songs": n, "samples_per_song": N, "content": "musical (tests/tools/musical_check.py)" if args.musical else "white noise",
           "musical": musical, "oracle_seconds": round(time.perf_counter() - t0, 1),
           "tempo_gpu_vs_oracle": tempo_histogram(err[:, 0]), "tempo_noise_floor": noise,
           "max_abs_err_non_tempo": float(err[:, 1:].max()),
           "max_abs_err_per_feature": [float(x) for x in err.max(axis=0)],
           "songs_over_1e-5_non_tempo": int((err[:, 1:].max(axis=1) > 1e-5).sum()),
           "tempo_mismatches_over_1e-4": int((err[:, 0] > 1e-4).sum()),
           "max_abs_err_tempo": float(err[:, 0].max()),
           "status_all_ok": bool((status.cpu().numpy() == 0).all())}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
