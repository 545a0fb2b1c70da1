#!/usr/bin/env python3
"""Call-by-call times of the s16 / f32 host feed from pageable memory (no medians): is the spread per call or per ring start?
    python tests/tools/stage_calls.py [songs] [calls] [starts] [lanes,slabs,kib,numa]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]


def main():
    import torch

    import bliss_rs_amd as bliss
    from bliss_rs_amd import _ffi

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    calls = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    starts = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    shape = tuple(int(v) for v in sys.argv[4].split(",")) if len(sys.argv) > 4 else (6, 3, 4096, 0)
    N = 3969000
    L = _ffi.lib()
    dctx = bliss.Context.default(0)
    rng = np.random.default_rng(0)
    s16 = rng.integers(-20000, 20000, n * N, dtype=np.int16)
    f32 = (rng.random(n * N, np.float32) - np.float32(0.5))
    p_s16 = torch.from_numpy(s16).pin_memory()
    offs = np.arange(n, dtype=np.uint64) * np.uint64(N)
    lens = np.full(n, N, np.uint64)
    res = np.empty((n, 23), np.float32)
    st = np.empty(n, np.int32)
    u64p = C.POINTER(C.c_uint64)

    def batch(fn, ptr):
        t0 = time.perf_counter()
        _ffi.check(fn(ptr, offs.ctypes.data_as(u64p), lens.ctypes.data_as(u64p), n, 2, res.ctypes.data, st.ctypes.data_as(C.POINTER(C.c_int32))))
        return (time.perf_counter() - t0) * 1e3

    print(os.environ.get("BLISSGPU_LIB") or "libblissgpu.so", "shape", shape, f"{n} songs; ms per call (s16 link time at 50 GB/s: {n * N * 2 / 50e6:.1f} ms)")
    batch(L.blissgpu_analyze_batch_s16, p_s16.data_ptr())
    print("pinned s16  ", " ".join(f"{batch(L.blissgpu_analyze_batch_s16, p_s16.data_ptr()):6.1f}" for _ in range(calls)))
    for s in range(starts):
        dctx.set_option("stage_lanes", 0)
        batch(L.blissgpu_analyze_batch_s16, s16.ctypes.data)   # (the ring is stopped only by a call that does not use it)
        for k, v in zip(("stage_lanes", "stage_slabs", "stage_slab_kib", "stage_numa"), shape):
            dctx.set_option(k, v)
        print(f"start {s} s16 ", " ".join(f"{batch(L.blissgpu_analyze_batch_s16, s16.ctypes.data):6.1f}" for _ in range(calls)))
        print(f"start {s} f32 ", " ".join(f"{batch(L.blissgpu_analyze_batch, f32.ctypes.data):6.1f}" for _ in range(calls)), flush=True)


if __name__ == "__main__":
    main()
