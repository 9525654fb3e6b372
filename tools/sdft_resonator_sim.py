"""Rounding of three sliding formulations of Harvest's band-pass (reference src/harvest.cpp:1261-1305: a Nuttall x cosine FIR) against
the FIR sum in 80-bit arithmetic, over one 2048-sample chunk of six bands (numpy, CPU; round-5 verdict item 1c):
  complex   seven complex rotators (hv_bandpass_sdft_kernel): 7 x 6 + 4 = 46 FP64 operations per output sample
  real2     real second-order resonators, direct form: 7 x 4 + 8 = 36
  reinsch   the same in Reinsch's form (differences): 7 x 5 + 9 = 44
    python tools/sdft_resonator_sim.py > profiles/r06_b_bandpass_resonators.txt"""
import numpy as np, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from world_class_amd.synth import make_utterance
fs=8000.0
x48=make_utterance(48000,3.0,3000)
y=x48[::6].copy()   # crude decimation, just a test signal at 8 kHz
nb=174
bands=71*0.9*2.0**((np.arange(nb)+1)/40.0)
hl=np.floor(fs/bands*2.0+0.5).astype(int)
mult=np.array([0,1,-1,2,-2,3,-3])
coef=np.array([0.355768,-0.243698,-0.243698,0.072116,0.072116,-0.006302,-0.006302])
i0=4096; CH=2048
ypad=np.concatenate([np.zeros(4096),y,np.zeros(8192)]); off=4096
def exact(b):
    h=hl[b]; k=np.arange(2*h+1,dtype=np.longdouble)
    t=k/(2*h)
    w=0.355768-0.487396*np.cos(2*np.pi*t)+0.144232*np.cos(4*np.pi*t)-0.012604*np.cos(6*np.pi*t)
    w=w*np.cos(2*np.pi*np.longdouble(bands[b])*(k-h)/fs)
    out=np.zeros(CH,dtype=np.longdouble)
    for i in range(CH):
        seg=ypad[off+i0+i+1-h: off+i0+i+2+h].astype(np.longdouble)
        out[i]=np.dot(seg,w)
    return out
errs={'complex':[], 'real2':[], 'reinsch':[]}
for b in [0,20,60,100,140,173]:
    h=hl[b]; w=2*np.pi*bands[b]/fs; om=np.pi/h
    nu=w+mult*om
    R=np.exp(1j*nu.astype(np.longdouble)).astype(np.complex128)
    P=np.complex128(np.exp(-1j*np.longdouble(w)*h))
    ex=exact(b)
    # complex sliding (current kernel): first window then slide
    D=np.zeros(7,dtype=np.complex128)
    for q in range(i0+1-h, i0+1+h+1):
        D=R*D+ypad[off+q]*P
    def outv(D): return float(np.sum(coef*D.real))
    oc=np.zeros(CH); oc[0]=outv(D)
    Dc=D.copy()
    for i in range(CH-1):
        yn=ypad[off+i0+i+2+h]; yo=ypad[off+i0+i+1-h]
        Dc=R*(Dc-yo*np.conj(P))+yn*P
        oc[i+1]=outv(Dc)
    errs['complex'].append(np.max(np.abs(oc-ex.astype(float)))/np.max(np.abs(ex)))
    # real second order: x(i+1) = 2 Rx x(i) - x(i-1) + A - Rx B + Ry C
    D1=R*(D-ypad[off+i0+1-h]*np.conj(P))+ypad[off+i0+2+h]*P
    xm=D.real.copy(); x0=D1.real.copy()
    Rx=R.real; Ry=R.imag
    orr=np.zeros(CH); orr[0]=np.sum(coef*xm); orr[1]=np.sum(coef*x0)
    # reinsch state
    xr=x0.copy(); dl=x0-xm; k4=4*np.sin(nu/2)**2
    ore=orr.copy()
    def uv(i):
        yn=ypad[off+i0+i+2+h]; yo=ypad[off+i0+i+1-h]
        u=yn*P; v=yo*np.conj(P); return u,v
    for i in range(1,CH-1):
        u,v=uv(i); up,vp=uv(i-1)
        A=u.real+vp.real; B=v.real+up.real; C=v.imag-up.imag
        g=A-Rx*B+Ry*C
        xn=2*Rx*x0-xm+g
        xm=x0; x0=xn
        orr[i+1]=np.sum(coef*x0)
        # reinsch: x(i+1)-x(i) = x(i)-x(i-1) - k4 x(i) + g
        dl=dl-k4*xr+g
        xr=xr+dl
        ore[i+1]=np.sum(coef*xr)
    errs['real2'].append(np.max(np.abs(orr-ex.astype(float)))/np.max(np.abs(ex)))
    errs['reinsch'].append(np.max(np.abs(ore-ex.astype(float)))/np.max(np.abs(ex)))
    print(b, bands[b], h, {k:'%.2e'%v[-1] for k,v in errs.items()}, flush=True)
