#!/usr/bin/env python3
"""One very long song (default 10 hours = 793.8 M samples, 3.2 GB of PCM -- past every 32-bit byte offset) through the
GPU path and the CPU oracle: python tests/tools/long_song_check.py [hours] > gpurun_out/long_song.json
The oracle takes ~20 s per hour of audio on one core."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]


def main():
    import torch

    import bliss_rs_amd as bliss
    import oracle as O

    hours = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
    n = int(hours * 3600 * 22050)
    c = bliss.Context(0)
    lens = np.array([n], np.uint64)
    offs = np.array([0], np.uint64)
    pcm = torch.empty(n + 64, dtype=torch.float32, device="cuda")
    c.synth_white_noise(pcm, offs, lens, first_song_index=4242)
    t0 = time.perf_counter()
    out, status = c.analyze(pcm, offs, lens, 2)
    c.synchronize()
    t_gpu = time.perf_counter() - t0
    got = out.cpu().numpy()[0]
    again, _ = c.analyze(pcm, offs, lens, 2)
    c.synchronize()
    x = O.white_noise(4242, n)
    same_pcm = bool(np.array_equal(pcm[n - (1 << 20):n].cpu().numpy(), x[-(1 << 20):]))
    t0 = time.perf_counter()
    ref = O.song_analyze(x)
    t_cpu = time.perf_counter() - t0
    err = np.abs(got.astype(np.float64) - ref)
    print(json.dumps({
        "samples": n, "hours": hours, "pcm_bytes": 4 * n, "status": int(status.cpu().numpy()[0]),
        "gpu_seconds_first_call": round(t_gpu, 3), "oracle_seconds": round(t_cpu, 1),
        "pcm_tail_identical_to_oracle_generator": same_pcm,
        "run_to_run_identical": bool(np.array_equal(again.cpu().numpy()[0], got)),
        "max_abs_err_non_tempo": float(err[1:].max()), "tempo_abs_err": float(err[0]),
        "gpu": [float(v) for v in got], "oracle": [float(v) for v in ref]}))


if __name__ == "__main__":
    main()
