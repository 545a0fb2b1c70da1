#!/usr/bin/env python3
"""Round-off model of the device's FFT-512 (kernels_fft512.hip) in numpy f32, operation for operation: half window, radix-16
(two radix-4 layers, W16 twiddles by one rounded product + one fused multiply-add), inter-pass twiddle W_256^(l k1) from the
table, second radix-16, the real-input split with W_512^k -- against a textbook radix-2 c2c f32 FFT (what the CPU oracle
runs in place of rustfft) and numpy's f64 FFT as the truth.  No GPU needed.  (Round-5 review item 3.)

    python tests/tools/fft_error_model_512.py       rms relative error of X[k], of |X[k]|, of a SpecFlux-like sum
    python tests/tools/fft_error_model_order.py     the same series with the SUMMATION ORDER varied instead of the transform
    python tests/tools/fft_error_model_8192.py      the FFT-8192 kernel's form with factored / directly rounded twiddles
Results of this round: profiles/r05_fft_error_model.txt
"""
import numpy as np
f32=np.float32; f64=np.float64
def fma(a,b,c): return (a.astype(f64)*b.astype(f64)+c.astype(f64)).astype(f32)   # single rounding (double product exact; sum may double-round rarely)
def mul(a,b): return (a*b).astype(f32)
def add(a,b): return (a+b).astype(f32)
def sub(a,b): return (a-b).astype(f32)
class C:
    def __init__(s,x,y): s.x=x.astype(f32); s.y=y.astype(f32)
def cadd(a,b): return C(add(a.x,b.x),add(a.y,b.y))
def csub(a,b): return C(sub(a.x,b.x),sub(a.y,b.y))
def cmul_pk(a,wx,wy):
    # t=(-a.y*w.y, a.y*w.x); r=(fma(a.x,w.x,t.x), fma(a.x,w.y,t.y))
    tx=mul(-a.y,wx*0+wy); ty=mul(a.y,wx*0+wx)
    return C(fma(a.x,wx*0+wx if np.ndim(wx) else np.full_like(a.x,wx),tx), fma(a.x,np.full_like(a.x,wy) if not np.ndim(wy) else wy,ty))
def bc(w,like): return np.full_like(like,w) if not np.ndim(w) else w.astype(f32)
def cmul(a,wx,wy):
    wx=bc(wx,a.x); wy=bc(wy,a.x)
    tx=mul(-a.y,wy); ty=mul(a.y,wx)
    return C(fma(a.x,wx,tx),fma(a.x,wy,ty))
def mul_mi(a): return C(a.y.copy(),-a.x)
def radix4(v0,v1,v2,v3):
    t0=cadd(v0,v2); t1=csub(v0,v2); t2=cadd(v1,v3); d=csub(v1,v3)
    o0=cadd(t0,t2); o2=csub(t0,t2)
    o1=C(add(t1.x,d.y),sub(t1.y,d.x)); o3=C(sub(t1.x,d.y),add(t1.y,d.x))
    return o0,o1,o2,o3
C1=f32(0.92387953251128674); S1=f32(0.38268343236508977); R2=f32(0.70710678118654752)
def R16(k): return 4*(k&3)+(k>>2)
def radix16(v):
    v=list(v)
    for b in range(4): v[b],v[4+b],v[8+b],v[12+b]=radix4(v[b],v[4+b],v[8+b],v[12+b])
    v[5]=cmul(v[5],C1,-S1); v[6]=cmul(v[6],R2,-R2); v[7]=cmul(v[7],S1,-C1); v[9]=cmul(v[9],R2,-R2)
    v[10]=mul_mi(v[10]); v[11]=cmul(v[11],-R2,-R2); v[13]=cmul(v[13],S1,-C1); v[14]=cmul(v[14],-R2,-R2); v[15]=cmul(v[15],-C1,S1)
    for c in range(4): v[4*c],v[4*c+1],v[4*c+2],v[4*c+3]=radix4(v[4*c],v[4*c+1],v[4*c+2],v[4*c+3])
    return v
def tw(n,N): a=-2*np.pi*n/N; return f32(np.cos(a)),f32(np.sin(a))
def gpu_fft512(xw):
    """xw: [F,512] windowed (half-window already applied) f32 -> complex X[0..256] as (re,im) f32 arrays [F,257] and 'sq' form"""
    F=xw.shape[0]
    z=C(xw[:,0::2],xw[:,1::2])   # z[n], n<256 ; n=16*n1+l
    # lane l holds n1=0..15: z[16 n1 + l]
    A=[[None]*16 for _ in range(16)]  # A[l][k1]
    for l in range(16):
        v=[C(z.x[:,16*n1+l],z.y[:,16*n1+l]) for n1 in range(16)]
        v=radix16(v)
        for k1 in range(16):
            a=v[R16(k1)]
            if k1>0:
                wx,wy=tw(l*k1,256); a=cmul(a,wx,wy)
            A[l][k1]=a
    Z=[None]*256
    for k1 in range(16):   # lane l'=k1 reads tile[l'*17+n2] = A[n2][k1] for n2..  (tile[k1*17+l]=A[l][k1]; read v[n2]=tile[l*17+n2] -> A[n2][l])
        v=[A[n2][k1] for n2 in range(16)]
        v=radix16(v)
        for k2 in range(16): Z[k1+16*k2]=v[R16(k2)]
    # split: X[k]=A+P with zk=Z[k], zm=Z[256-k]
    Xr=np.zeros((F,257),f32); Xi=np.zeros((F,257),f32)
    for k in range(1,256):
        zk=Z[k]; zm=Z[256-k]; wx,wy=tw(k,512)
        a=C(add(zk.x,zm.x),sub(zk.y,zm.y)); b=C(sub(zk.x,zm.x),add(zk.y,zm.y))
        wxb=bc(wx,a.x); wyb=bc(wy,a.x)
        tx=mul(b.x,wyb); ty=mul(-b.x,wxb)
        p=C(fma(b.y,wxb,tx),fma(b.y,wyb,ty))
        Xr[:,k]=add(a.x,p.x); Xi[:,k]=add(a.y,p.y)
    z0=Z[0]
    Xr[:,0]=mul(f32(2),add(z0.x,z0.y)); Xr[:,256]=mul(f32(2),sub(z0.x,z0.y))
    return Xr,Xi
def radix2_fft512(xw):
    """textbook iterative radix-2 DIT c2c f32 with table twiddles (the oracle's transform); xw = windowed FULL window"""
    F=xw.shape[0]; n=512
    rev=np.array([int(format(i,'09b')[::-1],2) for i in range(n)])
    re=xw[:,rev].astype(f32); im=np.zeros_like(re)
    k=np.arange(n//2); twr=np.cos(-2*np.pi*k/n).astype(f32); twi=np.sin(-2*np.pi*k/n).astype(f32)
    size=2
    while size<=n:
        half=size//2; step=n//size
        for start in range(0,n,size):
            i=np.arange(half); a=start+i; b=a+half
            wr=twr[i*step]; wi=twi[i*step]
            tr=sub(mul(re[:,b],wr),mul(im[:,b],wi)); ti=add(mul(re[:,b],wi),mul(im[:,b],wr))
            re[:,b]=sub(re[:,a],tr); im[:,b]=sub(im[:,a],ti)
            re[:,a]=add(re[:,a],tr); im[:,a]=add(im[:,a],ti)
        size*=2
    return re[:,:257],im[:,:257]
if __name__=='__main__':
    rng=np.random.default_rng(1)
    F=2000
    x=(rng.random((F,512),dtype=f32)-f32(0.5))
    i=np.arange(512,dtype=f32)
    hannz=(f32(0.5)*(f32(1)-np.cos((f32(2)*f32(np.pi))*i/f32(512)).astype(f32))).astype(f32)
    xw_full=mul(x,hannz); xw_half=mul(x,mul(hannz,f32(0.5)))
    ref=np.fft.rfft(xw_full.astype(f64),axis=1)
    gr,gi=gpu_fft512(xw_half)
    orr,oi=radix2_fft512(xw_full)
    scale=np.sqrt((np.abs(ref)**2).mean())
    eg=np.sqrt(((gr-ref.real)**2+(gi-ref.imag)**2).mean())/scale
    eo=np.sqrt(((orr-ref.real)**2+(oi-ref.imag)**2).mean())/scale
    print('complex rms rel err: gpu-style',eg,' radix2',eo)
    # magnitudes
    mref=np.abs(ref)
    mg=np.sqrt(add(mul(gr,gr),mul(gi,gi)).astype(f64)).astype(f32)  # correctly rounded sqrt
    mo=np.sqrt(add(mul(orr,orr),mul(oi,oi)).astype(f64)).astype(f32)
    print('mag rms rel err (correct sqrt): gpu-style',np.sqrt(((mg-mref)**2).mean())/scale,' radix2',np.sqrt(((mo-mref)**2).mean())/scale)
    # flux-like: sum over bins of max(m - prev,0) across consecutive frames
    def flux(m): return np.maximum(m[1:]-m[:-1],0).sum(axis=1)
    fr=flux(mref); print('flux rel err rms: gpu',np.sqrt((((flux(mg.astype(f64))-fr)/fr)**2).mean()),' radix2',np.sqrt((((flux(mo.astype(f64))-fr)/fr)**2).mean()))
