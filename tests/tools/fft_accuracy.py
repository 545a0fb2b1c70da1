#!/usr/bin/env python3
"""Which side of the f32 noise floor is the DEVICE's transform on?  (GPU box; round-5 review item 3.)

The feature-level comparison of tests/tools/full_check.py mixes the transforms with everything behind them (v_log / v_exp
against libm, summation orders).  This tool compares the per-frame series that come straight out of the two FFT kernels
-- the STFT magnitudes (FFT-8192, tap SPECTROGRAM) and the SpecFlux onset values + spectral centroids (FFT-512, taps FLUX /
CENTROID) -- with the oracle run on an f64 FFT, for the device and for the oracle's own textbook radix-2 f32 FFT:

    rms relative deviation from the f64-FFT oracle:   device   |   f32 oracle

Output: one JSON object (-> profiles/r05_fft_accuracy.json).
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))


def rms_rel(a, ref):
    a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    return float(np.sqrt(((a - ref) ** 2).mean()) / np.sqrt((ref ** 2).mean()))


def main():
    import torch

    import bliss_rs_amd as bliss
    import musical_check
    import oracle as O

    rng = np.random.default_rng(11)
    songs = [O.white_noise(900 + i, 30 * 22050) for i in range(6)] + [musical_check.make_song(rng, mods=False)[0] for _ in range(6)]
    names = ["noise"] * 6 + ["musical"] * 6
    ctx = bliss.Context(0)
    lens = [len(s) for s in songs]
    offs = np.concatenate([[0], np.cumsum([(l + 63) // 64 * 64 for l in lens])[:-1]]).astype(np.uint64)
    buf = np.zeros(int(offs[-1]) + lens[-1] + 64, np.float32)
    for s, o in zip(songs, offs):
        buf[int(o):int(o) + len(s)] = s
    ctx.analyze(torch.from_numpy(buf).cuda(), offs, lens, 2)
    ctx.synchronize()
    rows = []
    for i, (x, name) in enumerate(zip(songs, names)):
        dev = {"spec": ctx.debug_fetch("spectrogram", i), "flux": ctx.debug_fetch("flux", i), "centroid": ctx.debug_fetch("centroid", i)}

        def oracle_series():
            spec = O.stft(x, 8192, 2205).T  # [frames, bins]
            bd = O.BPMDesc().run(x)
            flux = bd.series()[0]
            cen = O.SpectralDesc().run(x).series()[0]
            return {"spec": spec[: dev["spec"].shape[0]], "flux": flux[: len(dev["flux"])], "centroid": cen[: len(dev["centroid"])]}

        f32 = oracle_series()
        O.set_fft_double(True)
        try:
            f64 = oracle_series()
        finally:
            O.set_fft_double(False)
        row = {"song": i, "kind": name}
        for key in ("spec", "flux", "centroid"):
            row[key] = {"device": rms_rel(dev[key][: len(f64[key])], f64[key]), "oracle_f32": rms_rel(f32[key], f64[key])}
        rows.append(row)
    out = {"what": "rms relative deviation of per-frame series from the oracle on an f64 FFT: device kernels vs the oracle's radix-2 f32 FFT",
           "series": {"spec": "STFT magnitudes (FFT-8192)", "flux": "SpecFlux onset values (FFT-512)", "centroid": "spectral centroid (FFT-512)"},
           "songs": rows}
    for key in ("spec", "flux", "centroid"):
        d = np.array([r[key]["device"] for r in rows])
        o = np.array([r[key]["oracle_f32"] for r in rows])
        out[key + "_summary"] = {"device_median": float(np.median(d)), "oracle_f32_median": float(np.median(o)),
                                 "device_closer_on": int((d < o).sum()), "of": len(rows)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
