#!/usr/bin/env python3
"""Round-off model of stft8192_kernel's transform (kernels_chroma.hip) in numpy f32: 4096 complex points as 16 x 16 x 16,
pass-1 twiddle W_256^(m1 k1) from the table, pass-2 twiddle as the kernel forms it -- the PRODUCT W_256^(m2 j1) * W_4096^(m2 k1)
of two rounded values -- or rounded directly, the split twiddle W_8192^(t + 256 j) = W_8192^t * W_32^j likewise; against the
radix-2 c2c f32 FFT of the oracle.  See fft_error_model_512.py.  (48 frames of white noise: a minute of numpy.)
"""
import numpy as np, sys
from fft_error_model_512 import *
def twc(n,N):
    a=-2*np.pi*(n%N)/N; return f32(np.cos(a)),f32(np.sin(a))
def cm(ax,ay,bx,by):   # f32 complex product of two scalars the way cmul_pk does
    a=C(np.array([ax],f32),np.array([ay],f32)); r=cmul(a,bx,by); return r.x[0],r.y[0]
def gpu_fft8192(xw, factored_p2=True, factored_split=True):
    F=xw.shape[0]
    zx=xw[:,0::2]; zy=xw[:,1::2]     # 4096 complex
    A=[[None]*256 for _ in range(16)]   # A[k1][t]
    for t in range(256):
        m1=t>>4
        v=[C(zx[:,256*n1+t],zy[:,256*n1+t]) for n1 in range(16)]
        v=radix16(v)
        for k1 in range(16):
            a=v[R16(k1)]
            if k1>0:
                wx,wy=twc(m1*k1,256); a=cmul(a,wx,wy)
            A[k1][t]=a
    Bm=[[None]*256 for _ in range(16)]  # B[j1][16*m2+k1]
    for t in range(256):
        k1=t&15; m2=t>>4
        v=[A[k1][16*m1+m2] for m1 in range(16)]
        v=radix16(v)
        cpx,cpy=twc(2*m2*k1,8192)
        for j1 in range(16):
            a=v[R16(j1)]
            if factored_p2:
                if j1==0: wx,wy=cpx,cpy
                else:
                    tx,ty=twc(m2*j1,256); wx,wy=cm(tx,ty,cpx,cpy)
            else:
                wx,wy=twc(m2*(k1+16*j1),4096)
            a=cmul(a,wx,wy)
            Bm[j1][16*m2+k1]=a
    Z=[None]*4096
    for t in range(256):
        k1=t&15; j1=t>>4
        v=[Bm[j1][16*m2+k1] for m2 in range(16)]
        v=radix16(v)
        for j2 in range(16): Z[t+256*j2]=v[R16(j2)]
    Xr=np.zeros((F,4097),f32); Xi=np.zeros((F,4097),f32)
    C32=[twc(j,32) for j in range(8)]
    for k in range(1,4096):
        zk=Z[k]; zm=Z[4096-k]
        kk=k if k<=2048 else 4096-k
        # thread t pairs bins k=t+256j (j<8) with mirrors; w for the LOW bin; the mirror uses the pair form
        if factored_split:
            t=kk&255; j=kk>>8
            if kk==2048: wx,wy=f32(0),f32(-1)
            else:
                sx,sy=twc(t,8192)
                wx,wy=(sx,sy) if j==0 else cm(sx,sy,C32[j][0],C32[j][1])
        else:
            wx,wy=twc(kk,8192)
        zk=Z[kk]; zm=Z[(4096-kk)%4096]
        a=C(add(zk.x,zm.x),sub(zk.y,zm.y)); b=C(sub(zk.x,zm.x),add(zk.y,zm.y))
        wxb=bc(wx,a.x); wyb=bc(wy,a.x)
        tx=mul(b.x,wyb); ty=mul(-b.x,wxb)
        p=C(fma(b.y,wxb,tx),fma(b.y,wyb,ty))
        if k<=2048:
            Xr[:,k]=add(a.x,p.x); Xi[:,k]=add(a.y,p.y)
        else:      # X[4096-kk] = conj(A - P)
            Xr[:,k]=sub(a.x,p.x); Xi[:,k]=-sub(a.y,p.y)
    z0=Z[0]
    Xr[:,0]=mul(f32(2),add(z0.x,z0.y)); Xr[:,4096]=mul(f32(2),sub(z0.x,z0.y))
    return Xr,Xi
def radix2_fft(xw,n):
    F=xw.shape[0]; bits=int(np.log2(n))
    rev=np.array([int(format(i,'0%db'%bits)[::-1],2) for i in range(n)])
    re=xw[:,rev].astype(f32); im=np.zeros_like(re)
    k=np.arange(n//2); twr=np.cos(-2*np.pi*k/n).astype(f32); twi=np.sin(-2*np.pi*k/n).astype(f32)
    size=2
    while size<=n:
        half=size//2; step=n//size
        re3=re.reshape(F,n//size,size); im3=im.reshape(F,n//size,size)
        wr=twr[np.arange(half)*step]; wi=twi[np.arange(half)*step]
        br=re3[:,:,half:]; bi=im3[:,:,half:]; ar=re3[:,:,:half]; ai=im3[:,:,:half]
        tr=sub(mul(br,wr),mul(bi,wi)); ti=add(mul(br,wi),mul(bi,wr))
        nbr=sub(ar,tr); nbi=sub(ai,ti); nar=add(ar,tr); nai=add(ai,ti)
        re3[:,:,half:]=nbr; im3[:,:,half:]=nbi; re3[:,:,:half]=nar; im3[:,:,:half]=nai
        size*=2
    return re[:,:n//2+1],im[:,:n//2+1]
if __name__=='__main__':
    rng=np.random.default_rng(2); F=48
    x=(rng.random((F,8192),dtype=f32)-f32(0.5))
    n=np.arange(8192,dtype=f32)
    hann=(f32(0.5)-f32(0.5)*np.cos(((f32(2)*n)*f32(np.pi))/f32(8192)).astype(f32)).astype(f32)
    xw=mul(x,hann); xh=mul(x,mul(hann,f32(0.5)))
    ref=np.fft.rfft(xw.astype(f64),axis=1); scale=np.sqrt((np.abs(ref)**2).mean())
    def rep(name,r,i):
        m=np.sqrt(add(mul(r,r),mul(i,i)).astype(f64)).astype(f32)
        print('%-44s complex %.4e   magnitude %.4e'%(name,np.sqrt(((r-ref.real)**2+(i-ref.imag)**2).mean())/scale,np.sqrt(((m-np.abs(ref))**2).mean())/scale),flush=True)
    rr,ri=radix2_fft(xw,8192); rep('radix-2 c2c 8192 (oracle)',rr,ri)
    for fp2,fs in ((True,True),(False,True),(True,False),(False,False)):
        gr,gi=gpu_fft8192(xh,fp2,fs); rep('device form: pass-2 tw %s, split tw %s'%('factored' if fp2 else 'direct','factored' if fs else 'direct'),gr,gi)
