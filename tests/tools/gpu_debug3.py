import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import torch
import bliss_rs_amd as bliss
import oracle as O
np.set_printoptions(linewidth=220, precision=6, suppress=True)
golden = (np.load(os.path.join(ROOT, "tests/golden/s16_mono_22_5kHz.pcm_s16.npy")).astype(np.float32) / np.float32(32768)).astype(np.float32)
for batch in (["golden", "noise5s", "short"], ["noise30s", "golden"], ["noise3min_1"]):
    songs = {"golden": golden, "noise5s": O.white_noise(1, 5 * 22050), "short": O.white_noise(1, 4000),
             "noise30s": O.white_noise(2, 30 * 22050 + 17), "noise3min_1": O.white_noise(1, 3969000)}
    ctx = bliss.Context(0)
    keys = batch
    lens = [len(songs[k]) for k in keys]
    offs = np.concatenate([[0], np.cumsum([(l + 63) // 64 * 64 for l in lens])[:-1]]).astype(np.uint64)
    buf = np.zeros(int(offs[-1]) + lens[-1] + 64, np.float32)
    for k, o in zip(keys, offs):
        buf[int(o):int(o) + len(songs[k])] = songs[k]
    pcm = torch.from_numpy(buf).cuda()
    out, status = ctx.analyze(pcm, offs, lens, 2)
    ctx.synchronize()
    g = out.cpu().numpy()
    tun, nb = ctx.last_tuning(len(keys))
    print("##### batch", batch, "status", status.cpu().tolist())
    for i, k in enumerate(keys):
        if lens[i] < 8192:
            continue
        d = O.BPMDesc().run(songs[k]); bpms = d.bpms()
        ob, oc = [], []
        for b in bpms:
            if ob and ob[-1] == b: oc[-1] += 1
            else: ob.append(b); oc.append(1)
        rb, rc = ctx.debug_fetch("run_bpm", i), ctx.debug_fetch("run_count", i)
        print(f"== {k}: gpu tempo {g[i][0]:.7f} oracle {d.get_value():.7f}  gpu n_bpms {nb[i]} sum(run_count) {rc.sum()} oracle {len(bpms)} runs {len(rb)}")
        print("   gpu counts   ", rc.tolist())
        print("   oracle counts", oc)
        gm = np.sort(np.repeat(rb, rc)); n = len(gm)
        if n:
            lo, hi = gm[(n - 1) // 2], gm[n - 1 - (n - 1) // 2]
            print("   median from fetched runs:", np.float32(lo + (hi - lo) / 2), "->", np.float32(2 * (lo + (hi - lo) / 2) / 206 - 1))
    del ctx
