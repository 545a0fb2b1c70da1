"""Probe: does running two half-batches on two independent contexts (own streams + workspaces) beat one full batch?
(HBM-bound tails of one half under the VALU-bound kernels of the other)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bliss_rs_amd as bliss

n, N, d = int(os.environ.get("SONGS", 1024)), 3969000, 23
lanes = int(os.environ.get("LANES", 2))
ctxs = [bliss.Context(0) for _ in range(lanes)]
offs = np.arange(n, dtype=np.uint64) * np.uint64(N)
lens = np.full(n, N, np.uint64)
pcm = torch.empty(n * N, dtype=torch.float32, device="cuda")
ctxs[0].synth_white_noise(pcm, offs, lens, 0)
torch.cuda.synchronize()
out = torch.empty((n, d), dtype=torch.float32, device="cuda")
status = torch.empty((n,), dtype=torch.int32, device="cuda")
ref, _ = ctxs[0].analyze(pcm, offs, lens, 2)
torch.cuda.synchronize()
ref = ref.clone()
for c in ctxs:  # lanes must not be chained through torch's stream: inputs are ready, outputs are read after a device sync
    c._pre = lambda: None
    c._post = lambda: None

def step(parts):
    per = n // parts
    for p in range(parts):
        lo, hi = p * per, (p + 1) * per if p + 1 < parts else n
        ctxs[p % lanes].analyze(pcm, offs[lo:hi], lens[lo:hi], 2, out=out[lo:hi], status=status[lo:hi])

for parts in (1, 2, 4, 8):
    step(parts); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): step(parts)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"parts {parts} lanes {lanes}: {dt*1e3:.2f} ms/step  {n/dt:.0f} songs/s  identical {bool(torch.equal(out, ref))}")
