#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry points from PAGEABLE memory through the pinned staging ring, against the same
call from page-locked memory, over ring shapes (lanes x slabs x slab size): the measurement behind STAGE_*_DEFAULT in ctx.hpp.

    python tests/tools/stage_sweep.py [songs] [shapes]      shapes = "lanes,slabs,kib[,numa];..." (default: a sweep)

Three feed forms (3-minute songs): mono f32 at 22 050 Hz (blissgpu_analyze_batch), mono s16 (blissgpu_analyze_batch_s16),
44.1 kHz stereo s16 as a decoder delivers it (blissgpu_analyze_batch_decoded).  Every number is the median of 3 calls."""
import ctypes as C
import os
import statistics
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
    shapes = [(0, 3, 4096), (2, 3, 4096), (4, 3, 4096), (6, 3, 4096), (8, 3, 4096), (12, 3, 4096), (16, 3, 4096),
              (6, 2, 4096), (6, 4, 4096), (6, 3, 1024), (6, 3, 2048), (6, 3, 8192), (6, 3, 16384), (8, 3, 2048), (8, 2, 8192)]
    if len(sys.argv) > 2:
        shapes = [tuple(int(v) for v in s.split(",")) for s in sys.argv[2].split(";")]
    N = 3969000
    L = _ffi.lib()
    dctx = bliss.Context.default(0)
    rng = np.random.default_rng(0)
    f32 = (rng.random(n * N, np.float32) - np.float32(0.5))
    s16 = rng.integers(-20000, 20000, n * N, dtype=np.int16)
    n44 = min(n, 64)
    s44 = rng.integers(-20000, 20000, (n44 * 2 * N, 2), dtype=np.int16)
    p_f32, p_s16, p_s44 = (torch.from_numpy(a).pin_memory() for a in (f32, s16, s44))
    offs = np.arange(n, dtype=np.uint64) * np.uint64(N)
    lens = np.full(n, N, np.uint64)
    res = np.empty((n, 23), np.float32)
    st = np.empty(n, np.int32)
    u64p = C.POINTER(C.c_uint64)

    def batch(fn, ptr):
        t0 = time.perf_counter()
        _ffi.check(fn(ptr, offs.ctypes.data_as(u64p), lens.ctypes.data_as(u64p), n, 2, res.ctypes.data, st.ctypes.data_as(C.POINTER(C.c_int32))))
        return time.perf_counter() - t0

    def decoded(base):
        arr = (_ffi.DecodedSong * n44)()
        for i in range(n44):
            arr[i] = _ffi.DecodedSong(base + i * 2 * N * 4, 2 * N, 44100, 2, _ffi.SAMPLE_S16)
        t0 = time.perf_counter()
        _ffi.check(L.blissgpu_analyze_batch_decoded(arr, n44, 2, res.ctypes.data, st.ctypes.data_as(C.POINTER(C.c_int32))))
        return time.perf_counter() - t0

    def med(f):
        f()
        return statistics.median(f() for _ in range(3))

    forms = (("f32", lambda p: batch(L.blissgpu_analyze_batch, p), f32.ctypes.data, p_f32.data_ptr(), n, N * 4),
             ("s16", lambda p: batch(L.blissgpu_analyze_batch_s16, p), s16.ctypes.data, p_s16.data_ptr(), n, N * 2),
             ("44k1_stereo_s16", decoded, s44.ctypes.data, p_s44.data_ptr(), n44, 2 * N * 4))
    print(f"host cpus {os.cpu_count()}, {n} songs ({n44} for the decoded form), 3-minute songs; songs/s (GB/s)")
    pinned = {}
    hashes = {}
    for name, fn, pageable_ptr, pinned_ptr, k, bytes_per in forms:
        t = med(lambda: fn(pinned_ptr))
        pinned[name] = k / t
        hashes[name] = hash(res[:k].tobytes())
        print(f"pinned   {name:16s} {k / t:8.1f} ({k * bytes_per / t / 1e9:5.1f})")
    for shape in shapes:
        lanes, slabs, kib = shape[:3]
        numa = shape[3] if len(shape) > 3 else 1
        dctx.set_option("stage_lanes", lanes)
        dctx.set_option("stage_slabs", slabs)
        dctx.set_option("stage_slab_kib", kib)
        dctx.set_option("stage_numa", numa)
        line = f"pageable lanes {lanes:2d} slabs {slabs} slab {kib:6d} KiB numa {numa}:"
        for name, fn, pageable_ptr, pinned_ptr, k, bytes_per in forms:
            t = med(lambda: fn(pageable_ptr))
            same = hash(res[:k].tobytes()) == hashes[name]
            line += f"  {name} {k / t:7.1f} ({k * bytes_per / t / 1e9:5.1f}) = {k / t / pinned[name]:.2f}x{'' if same else ' ROWS DIFFER'}"
        print(line, flush=True)


if __name__ == "__main__":
    main()
