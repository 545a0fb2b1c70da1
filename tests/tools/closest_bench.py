#!/usr/bin/env python3
"""closest_to_songs on the device: mean wall time of the whole call (distance kernel + stable sort + NaN check) over 100 k
candidates, and the sort's share.    python tests/tools/closest_bench.py [n=100000] [reps=200]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

import bliss_rs_amd as bliss

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rng = np.random.default_rng(1)
X = torch.from_numpy(rng.uniform(-1, 1, (n, 23)).astype(np.float32)).cuda()
seeds = X[:3].clone()
ctx = bliss.Context(0)
for metric in ("euclidean", "mahalanobis"):
    M = torch.eye(23, device="cuda") if metric == "mahalanobis" else None
    order = ctx.closest_to_songs(seeds, X, metric, M)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        order = ctx.closest_to_songs(seeds, X, metric, M)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    t0 = time.perf_counter()
    for _ in range(reps):
        d = ctx.set_distance(seeds, X, metric, M)
    torch.cuda.synchronize()
    ms_d = (time.perf_counter() - t0) / reps * 1e3
    print(f"closest_to_songs n={n} {metric}: {ms:.3f} ms per call; the distance kernel alone {ms_d:.3f} ms")
