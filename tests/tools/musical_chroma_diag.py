#!/usr/bin/env python3
"""Where do the chroma features of the MUSICAL bench batch (bench.py --config musical: 1024 seeded three-minute songs) leave
the oracle?  Lists every song with a chroma feature (10 .. 22) further than 1e-5 from the oracle, its tuning on both sides,
and repeats the first few of them ALONE (batch of one), with a larger candidate pool, and with the chroma taps, to tell a
batch effect from a per-song one.  One JSON document on stdout.

    python tests/tools/musical_chroma_diag.py [--songs 1024] [--alone 6]
"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.dirname(os.path.abspath(__file__))]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--songs", type=int, default=1024)
    ap.add_argument("--samples", type=int, default=3969000)
    ap.add_argument("--seed", type=int, default=20260927)
    ap.add_argument("--alone", type=int, default=6)
    ap.add_argument("--threads", type=int, default=64)
    ap.add_argument("--only", type=int, default=-1, help="just this song of the batch, alone: its device pitch histogram beside the oracle's")
    args = ap.parse_args()
    import torch

    import bliss_rs_amd as bliss
    import oracle as O
    from musical_check import musical_batch

    n, N = args.songs, args.samples
    if args.only >= 0:
        from musical_check import make_song

        x, m = make_song(np.random.default_rng([args.seed, args.only]), False, N)
        c = bliss.Context(0)
        o, _ = c.analyze(torch.from_numpy(x).cuda(), np.zeros(1, np.uint64), np.array([N], np.uint64), 2)
        c.synchronize()
        hist = c.debug_fetch("pitch_hist", 0)
        t, _ = c.last_tuning(1)
        # the oracle's histogram (src/chroma.rs:334-391: peaks at or above the median magnitude, 100 bins of the fractional pitch)
        pit, mag = O.pip_track(22050, O.stft(x, 8192, 2205), 8192)
        srt = np.sort(mag)
        fi = 0.5 * (len(mag) - 1)
        thr = srt[int(np.floor(fi))] + (srt[int(np.ceil(fi))] - srt[int(np.floor(fi))]) / 2
        xx = np.fmod(12 * np.log2(pit[mag >= thr] / (440.0 / 16.0)), 1.0)
        xx[xx >= 0.5] -= 1.0
        cnt = np.bincount(np.clip(((xx + 0.5) / 0.01).astype(np.int64), 0, 99), minlength=100)
        diff = np.flatnonzero(hist[:100].astype(np.int64) != cnt)
        # the reference's estimate on the DEVICE's spectrogram: is the device's answer the reference algorithm's answer on the
        # magnitudes the device computed (then the two sides differ by FFT rounding only), or a fault of the tuning kernels?
        gspec = c.debug_fetch("spectrogram", 0).astype(np.float64)      # [frames, 4097]
        ospec = O.stft(x, 8192, 2205)                                    # [4097, frames]
        gpit, gmag = O.pip_track(22050, gspec.T, 8192)
        on_gpu_spec = O.estimate_tuning(22050, gspec.T, 8192, 0.01, 12)
        extra = {"oracle_algorithm_on_the_device_spectrogram": float(on_gpu_spec), "peaks_on_the_device_spectrogram": int(len(gpit)),
                 "spectrogram_max_abs_diff": float(np.abs(gspec - ospec.T).max()), "spectrogram_max": float(ospec.max()),
                 "spectrogram_rel_rms_diff": float(np.sqrt(((gspec - ospec.T) ** 2).mean()) / np.sqrt((ospec ** 2).mean()))}
        print(json.dumps({"song": args.only, **m, "tuning_gpu": float(t[0]), "tuning_oracle": float(O.chroma_desc(x)[1]), "peaks": int(len(pit)),
                          "selected": int(cnt.sum()), "gpu_selected": int(hist[:100].sum()), "hist_len": int(len(hist)),
                          "oracle_top": [[int(b), int(cnt[b])] for b in np.argsort(-cnt)[:4]],
                          "gpu_top": [[int(b), int(hist[b])] for b in np.argsort(-hist[:100].astype(np.int64))[:4]],
                          "bins_that_differ": [[int(b), int(hist[b]), int(cnt[b])] for b in diff], **extra}))
        return
    songs, meta = musical_batch(n, N, seed=args.seed)
    ctx = bliss.Context(0)
    offs = np.arange(n, dtype=np.uint64) * np.uint64(N)
    lens = np.full(n, N, np.uint64)
    pcm = torch.empty(n * N, dtype=torch.float32, device="cuda")
    for i, x in enumerate(songs):
        pcm[i * N:(i + 1) * N] = torch.from_numpy(x).cuda()
    out, status = ctx.analyze(pcm, offs, lens, 2)
    ctx.synchronize()
    got = out.cpu().numpy()
    tuning, _ = ctx.last_tuning(n)

    def ochroma(x):
        c, t = O.chroma_desc(x)
        return np.asarray(O.chroma_get_values(c, 2), np.float32), float(t), c.shape[1]

    t0 = time.perf_counter()
    with ThreadPoolExecutor(args.threads) as ex:
        res = list(ex.map(ochroma, songs))
    ref = np.stack([r[0] for r in res])
    otun = np.array([r[1] for r in res])
    err = np.abs(got[:, 10:].astype(np.float64) - ref)
    bad = np.flatnonzero(err.max(axis=1) > 1e-5)
    doc = {"songs": n, "oracle_seconds": round(time.perf_counter() - t0, 1), "chroma_songs_over_1e-5": int(len(bad)),
           "max_err": float(err.max()), "tuning_mismatches": int((np.abs(tuning - otun) > 1e-12).sum()),
           "by_kind": {str(k): [int(sum(meta[i]["kind"] == k for i in bad)), int(sum(m["kind"] == k for m in meta))] for k in range(4)},
           "bad": [{"song": int(i), **meta[i], "err_max": float(err[i].max()), "worst_feature": 10 + int(err[i].argmax()),
                    "tuning_gpu": float(tuning[i]), "tuning_oracle": float(otun[i])} for i in bad]}
    # ---- the first few alone ----
    alone = []
    for i in bad[: args.alone]:
        x = songs[int(i)]
        one = torch.from_numpy(x).cuda()
        rec = {"song": int(i)}
        for label, opts in (("alone", {}), ("alone_cand_budget_192", {"cand_budget": 192}), ("alone_serial", {"serial": 1})):
            c2 = bliss.Context(0)
            for k, v in opts.items():
                c2.set_option(k, v)
            c2.set_option("debug_chroma", 1)
            o2, _ = c2.analyze(one, np.zeros(1, np.uint64), np.array([N], np.uint64), 2)
            c2.synchronize()
            g2 = o2.cpu().numpy()[0]
            t2, _ = c2.last_tuning(1)
            rec[label] = {"err_max": float(np.abs(g2[10:].astype(np.float64) - ref[i]).max()), "tuning": float(t2[0]),
                          "equal_to_batch_row": bool(np.array_equal(g2.view(np.uint32), got[i].view(np.uint32)))}
            if label == "alone":
                ch = c2.debug_fetch("chroma", 0)          # [frames, 12]
                oc, _ = O.chroma_desc(x)                  # [12, frames]
                dch = np.abs(ch[: oc.shape[1]] - oc.T)
                fr = np.flatnonzero(dch.max(axis=1) > 1e-9)
                rec["chroma_frames"] = int(oc.shape[1])
                rec["chroma_frames_off_by_1e-9"] = int(len(fr))
                rec["first_off_frames"] = [int(f) for f in fr[:12]]
                rec["max_chroma_frame_err"] = float(dch.max())
                if len(fr):
                    f = int(fr[0])
                    rec["frame_example"] = {"frame": f, "gpu": [float(v) for v in ch[f]], "oracle": [float(v) for v in oc[:, f]]}
                    # the spectrogram of that frame on both sides
                    sp = c2.debug_fetch("spectrogram", 0)
                    osp = O.stft(x, 8192, 2205)   # [4097, frames] f64
                    rec["frame_example"]["gpu_spec_max"] = float(sp[f].max())
                    rec["frame_example"]["gpu_spec_nonzero"] = int((sp[f] != 0).sum())
                    if osp is not None:
                        o_f = osp[:, f]
                        rec["frame_example"]["oracle_spec_max"] = float(o_f.max())
                        rec["frame_example"]["spec_max_abs_diff"] = float(np.abs(sp[f] - o_f).max())
            c2.close()
        alone.append(rec)
    doc["alone"] = alone
    print(json.dumps(doc))


if __name__ == "__main__":
    main()
