import os, sys, time, numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [R, os.path.join(R, "oracle")]
import torch, bliss_rs_amd as bliss, oracle as O
n = 45 * 60 * 22050 + 777
x = O.white_noise(4242, n)
ctx = bliss.Context(0)
pcm = torch.from_numpy(x).cuda()
t0 = time.perf_counter()
out, st = ctx.analyze(pcm, [0], [n], 2); ctx.synchronize()
print("gpu", time.perf_counter() - t0, "s", st.cpu().numpy())
t0 = time.perf_counter(); ref = O.song_analyze(x); print("oracle", time.perf_counter() - t0, "s")
err = np.abs(out.cpu().numpy()[0] - ref)
print("max non-tempo err", err[1:].max(), "tempo err", err[0])
