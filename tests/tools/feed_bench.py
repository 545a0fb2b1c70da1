#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry points (blissgpu_analyze_batch / _s16) on N three-minute songs in pinned
memory: BLISSGPU_LIB=<variant.so> python tests/tools/feed_bench.py [songs]"""
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

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    N = 3969000
    c = bliss.Context(0)
    lens = np.full(n, N, np.uint64)
    offs = np.arange(n, dtype=np.uint64) * np.uint64(N)
    pcm = torch.empty(n * N + 64, dtype=torch.float32, device="cuda")
    c.synth_white_noise(pcm, offs, lens, first_song_index=0)
    h_f32 = torch.empty(n * N, dtype=torch.float32, pin_memory=True)
    h_f32.copy_(pcm[: n * N])
    h_s16 = torch.empty(n * N, dtype=torch.int16, pin_memory=True)
    h_s16.copy_((pcm[: n * N] * 32768.0).round().clamp(-32768, 32767).to(torch.int16))
    del pcm
    L = _ffi.lib()
    res = np.empty((n, 23), np.float32)
    st = np.empty(n, np.int32)

    def run(fn, ptr):
        t0 = time.perf_counter()
        _ffi.check(fn(ptr, offs.ctypes.data_as(C.POINTER(C.c_uint64)), lens.ctypes.data_as(C.POINTER(C.c_uint64)), n, 2,
                      res.ctypes.data, st.ctypes.data_as(C.POINTER(C.c_int32))))
        return time.perf_counter() - t0

    out = {}
    for name, fn, ptr, bps in (("f32", L.blissgpu_analyze_batch, h_f32.data_ptr(), 4), ("s16", L.blissgpu_analyze_batch_s16, h_s16.data_ptr(), 2)):
        run(fn, ptr)
        t = min(run(fn, ptr) for _ in range(4))
        out[name] = f"{n / t:8.1f} songs/s {n * N * bps / t / 1e9:6.2f} GB/s {t * 1e3:7.2f} ms"
    print(os.environ.get("BLISSGPU_LIB", "libblissgpu.so"), out, "row hash", hash(res.tobytes()) & 0xffffffff)


if __name__ == "__main__":
    main()
