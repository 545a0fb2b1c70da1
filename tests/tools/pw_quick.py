import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import torch
import bliss_rs_amd as bliss
import oracle as O
ctx = bliss.Context(0)
rng = np.random.default_rng(1)
for d in (23, 20):
    A = rng.uniform(-1, 1, (300, d)).astype(np.float32); B = rng.uniform(-1, 1, (517, d)).astype(np.float32)
    R = rng.uniform(-1, 1, (d, d)).astype(np.float32); psd = (R @ R.T).astype(np.float32); diag = np.diag(rng.uniform(0, 1, d).astype(np.float32)).astype(np.float32)
    for metric, m in (("euclidean", None), ("cosine", None), ("mahalanobis", diag), ("mahalanobis", psd)):
        got = bliss.playlist.pairwise_distances(A, B, metric, m); ref = O.pairwise(A, B, metric, m)
        print(d, metric, "diag" if m is diag else "", "bit-exact:", np.array_equal(got, ref, equal_nan=True))
n = 100000
g = torch.Generator(device="cuda").manual_seed(1234)
A = torch.rand((n, 23), generator=g, device="cuda") * 2 - 1
D = torch.empty((n, n), dtype=torch.float32, device="cuda")
W = torch.from_numpy(O.feature_weights(2)).cuda()
B = A.clone()
for metric, M in (("euclidean", None), ("cosine", None), ("mahalanobis", W)):
    for name, rhs in (("self", A), ("general", B)):
        ctx.pairwise(A, rhs, metric, M=M, out=D); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): ctx.pairwise(A, rhs, metric, M=M, out=D)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        print(f"{metric:12s} {name:8s} {dt*1e3:8.3f} ms  {n*n/dt/1e12:.3f} Tpairs/s  {4.0*n*n/dt/1e9:.0f} GB/s")
