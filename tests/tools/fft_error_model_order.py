#!/usr/bin/env python3
"""How much of the distance between the device's SpecFlux / centroid series and the oracle's is the TRANSFORM, how much the
ORDER in which 257 floats are added?  The reference adds them one by one in f32 (src/aubio.rs:455-467, 16-29); the kernel
adds 16 per lane and the 16 lane sums by a DPP tree.  Same magnitudes, the two orders; and the two transforms, the same
order.  See fft_error_model_512.py.
"""
import numpy as np
from fft_error_model_512 import *
rng=np.random.default_rng(3)
F=4000
x=(rng.random((F,512),dtype=f32)-f32(0.5))
i=np.arange(512,dtype=f32)
hannz=(f32(0.5)*(f32(1)-np.cos((f32(2)*f32(np.pi))*i/f32(512)).astype(f32))).astype(f32)
xw_full=mul(x,hannz); xw_half=mul(x,mul(hannz,f32(0.5)))
ref=np.fft.rfft(xw_full.astype(f64),axis=1)
m64=np.abs(ref).astype(f32)                       # "oracle on an f64 FFT": magnitudes rounded to f32 once
orr,oi=radix2_fft512(xw_full); m32=np.sqrt(add(mul(orr,orr),mul(oi,oi)).astype(f64)).astype(f32)
gr,gi=gpu_fft512(xw_half); mg=np.sqrt(add(mul(gr,gr),mul(gi,gi)).astype(f64)).astype(f32)
def flux_seq(m):
    d=np.maximum(sub(m[1:],m[:-1]),f32(0)); acc=np.zeros(len(d),f32)
    for j in range(257): acc=add(acc,d[:,j])
    return acc
def flux_gpu(m):
    d=np.maximum(sub(m[1:],m[:-1]),f32(0))
    lane=np.zeros((len(d),16),f32)
    for e in range(16):
        for l in range(16): pass
    for l in range(16):
        a=np.zeros(len(d),f32)
        for e in range(16): a=add(a,d[:,16*l+e])
        lane[:,l]=a
    lane[:,0]=add(lane[:,0],d[:,256])
    v=lane
    for step in (1,2):   # quad xor1, xor2
        idx=np.arange(16)^step; v=add(v,v[:,idx])
    idx=np.array([7-(k%8)+8*(k//8) for k in range(16)]); v=add(v,v[:,idx])   # row_half_mirror
    idx=15-np.arange(16); v=add(v,v[:,idx])
    return v[:,0]
def cen_seq(m):
    s=np.zeros(len(m),f32); w=np.zeros(len(m),f32)
    for j in range(256):
        s=add(s,m[:,j]); w=add(w,mul(f32(j),m[:,j]))
    return (w/s).astype(f32)
ref_flux=flux_seq(m64); ref_c=cen_seq(m64)
def rr(a,b): return float(np.sqrt(((a.astype(f64)-b)**2).mean())/np.sqrt((b.astype(f64)**2).mean()))
print('flux  : radix2 mags, sequential sum (= f32 oracle)',rr(flux_seq(m32),ref_flux))
print('flux  : radix2 mags, GPU-order sum              ',rr(flux_gpu(m32),ref_flux))
print('flux  : GPU-style mags, GPU-order sum            ',rr(flux_gpu(mg),ref_flux))
print('flux  : GPU-style mags, sequential sum           ',rr(flux_seq(mg),ref_flux))
print('flux  : f64 mags, GPU-order sum (order alone)    ',rr(flux_gpu(m64),ref_flux))
print('centr : radix2 seq',rr(cen_seq(m32),ref_c),' gpu-style mags seq',rr(cen_seq(mg),ref_c))
