#!/usr/bin/env python3
"""Device time of the resample kernel alone (kernels_pcm.hip: resample_kernel), input resident in HBM (GPU box).
One 3-minute song per case; microseconds per song, input bytes per second, outputs per second -> profiles/r05_resample_kernel.txt"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bliss_rs_amd as bliss  # noqa: E402

ctx = bliss.Context(0)
g = torch.Generator(device="cuda").manual_seed(1)
print("rate  channels fmt  taps phases  us_per_3min_song  input_GBps  Moutputs_per_s")
for rate, ch, dt in ((44100, 2, torch.int16), (44100, 1, torch.int16), (48000, 2, torch.int16), (48000, 2, torch.float32),
                     (96000, 2, torch.int32), (88200, 2, torch.int16), (44056, 2, torch.int16), (16000, 1, torch.int16)):
    frames = 180 * rate
    if dt == torch.float32:
        x = torch.rand((frames, ch), device="cuda", generator=g) - 0.5
    else:
        x = torch.randint(-20000, 20000, (frames, ch), device="cuda", generator=g).to(dt)
    if ch == 1:
        x = x[:, 0].contiguous()
    out = ctx.pcm_decode(x, rate)
    ctx.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        ctx.pcm_decode(x, rate, out=out)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    import ctypes as C
    from bliss_rs_amd import _ffi
    taps, pc = C.c_uint32(), C.c_uint32()
    _ffi.lib().blissgpu_resample_filter(rate, None, 0, C.byref(taps), C.byref(pc))
    print(f"{rate:6d} {ch} {str(dt).split('.')[-1]:8s} {taps.value:4d} {pc.value:5d}  {us:10.1f}  {x.numel() * x.element_size() / us / 1e3:8.1f}  {out.numel() / us:8.1f}")
