import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
import torch
import bliss_rs_amd as bliss
ctx = bliss.Context(0)
n = 100000
g = torch.Generator(device="cuda").manual_seed(1234)
A = torch.rand((n, 23), generator=g, device="cuda") * 2 - 1
D = torch.empty((n, n), dtype=torch.float32, device="cuda")
ctx.pairwise(A, A, "euclidean", out=D); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): ctx.pairwise(A, A, "euclidean", out=D)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
print(os.environ.get("BLISSGPU_ABLPW"), f"{dt*1e3:8.3f} ms")
