"""Developer aid (GPU box): stage-by-stage tempo comparison + first timing of a 64-song batch."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import torch  # noqa: E402

import bliss_rs_amd as bliss  # noqa: E402
import oracle as O  # noqa: E402

np.set_printoptions(linewidth=220, precision=6, suppress=True)


def main():
    golden = (np.load(os.path.join(ROOT, "tests/golden/s16_mono_22_5kHz.pcm_s16.npy")).astype(np.float32) / np.float32(32768)).astype(np.float32)
    songs = {"golden": golden, "noise5s": O.white_noise(1, 5 * 22050), "noise3min_1": O.white_noise(1, 3969000)}
    ctx = bliss.Context(0)
    keys = list(songs)
    lens = [len(songs[k]) for k in keys]
    offs = np.concatenate([[0], np.cumsum([(l + 63) // 64 * 64 for l in lens])[:-1]]).astype(np.uint64)
    buf = np.zeros(int(offs[-1]) + lens[-1] + 64, np.float32)
    for k, o in zip(keys, offs):
        buf[int(o):int(o) + len(songs[k])] = songs[k]
    pcm = torch.from_numpy(buf).cuda()
    out, status = ctx.analyze(pcm, offs, lens, 2)
    ctx.synchronize()
    for i, k in enumerate(keys):
        d = O.BPMDesc().run(songs[k])
        onset, thr = d.series()
        bpms = d.bpms()
        gflux, gthr = ctx.debug_fetch("flux", i), ctx.debug_fetch("thresholded", i)
        rb, rc = ctx.debug_fetch("run_bpm", i), ctx.debug_fetch("run_count", i)
        print(f"=== {k}: frames {len(onset)} gpu {len(gflux)}; flux max abs diff {np.abs(onset - gflux).max():.3g} (scale {np.abs(onset).max():.3g}) at {int(np.abs(onset - gflux).argmax())}")
        print(f"    thr max abs diff {np.abs(thr - gthr).max():.3g} at {int(np.abs(thr - gthr).argmax())}")
        # oracle bpms grouped into consecutive equal runs
        ob, oc = [], []
        for b in bpms:
            if ob and ob[-1] == b:
                oc[-1] += 1
            else:
                ob.append(b); oc.append(1)
        print("    oracle (bpm,count):", [(round(float(b), 3), c) for b, c in zip(ob, oc)][:40])
        print("    gpu    (bpm,count):", [(round(float(b), 3), int(c)) for b, c in zip(rb, rc)][:40])
    # ---- timing: 64 x 3-min songs generated on device ----
    n, N = 64, 3969000
    offs = (np.arange(n, dtype=np.uint64) * N)
    lens = np.full(n, N, np.uint64)
    pcm = torch.empty(n * N, dtype=torch.float32, device="cuda")
    ctx.synth_white_noise(pcm, offs, lens, 0)
    ctx.profile_enable(True)
    for it in range(3):
        ctx.profile_reset()
        torch.cuda.synchronize()
        t0 = time.time()
        out, status = ctx.analyze(pcm, offs, lens, 2)
        ctx.synchronize()
        dt = time.time() - t0
        print(f"iter {it}: {n} songs in {dt * 1e3:.1f} ms -> {n / dt:.1f} songs/s")
    for k, v in ctx.profile().items():
        print(f"   {k:22s} {v[0]:10.3f} ms  x{v[1]}")
    g = out.cpu().numpy()
    print("row0", g[0])
    t0 = time.time()
    ref = O.song_analyze(O.white_noise(0, N))
    print("oracle 1 song:", time.time() - t0, "s; err", np.abs(ref - g[0]).max())


if __name__ == "__main__":
    main()
