"""Rounds-of-workgroups model of the step's 3x3 forward / data-gradient launches (HISTORY.md section 8): work items per launch from
the plans of csrc/conv2d.hip (tile shapes restated here), 512 slots (two workgroups per CU), and the share of each launch's measured
time that a partly filled last round can account for AT MOST (a workgroup alone on its CU runs ~1.7x faster, and in the step another
stream fills idle CUs).  Input: profiles/r4_roofline_by_shape_church256.txt.   python tools/round_model.py"""
import math
import os
import re
rows=[]
for line in open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'r4_roofline_by_shape_church256.txt')):
    m=re.match(r'conv 3x3 s(\d) (fwd\S*|dgrad|wgrad)\s*(\(modulated\))?\s+n(\d+)\s+(\d+)->(\d+)\s+(\d+)x(\d+)\s+mfma\s+([\d.]+)\s+([\d.]+) GF\s+([\d.]+)\s+([\d.]+) TF/s\s+([\d.]+)',line)
    if not m: continue
    s,op,mod,n,ci,co,h,w,calls,gf,ms,tf,frac=m.groups()
    rows.append((int(s),op[:5],bool(mod),int(n),int(ci),int(co),int(h),int(w),float(calls),float(ms),float(frac)))
def cd(a,b): return -(-a//b)
tot=0; lost=0; 
out=[]
for s,op,mod,n,ci,co,h,w,calls,ms,frac in rows:
    if op.startswith('wgrad'): continue
    if s==1:
        oh=h if h&(h-1)==0 else h-2
        M = co if op.startswith('fwd') else ci
        pix = oh*oh if op.startswith('fwd') else h*w
        if M<=32: bm,bn=32,256
        elif M<=64: bm,bn=64,256
        elif oh>=128 or cd(M,64)*64*100 < cd(M,128)*128*92: bm,bn=64,256
        else: bm,bn=128,128
        items=cd(M,bm)*cd(n*pix,bn) if pix<bn else cd(M,bm)*n*cd(pix,bn)
        slots=512 if bm>32 else 768
    else:
        oh=(h-3)//2+1 if h%2 else h//2
        if op.startswith('fwd'):
            M=co; pix=oh*oh
            bm,bn=(128,128) if M>64 else (64,256)
            items=cd(M,bm)*(cd(n*pix,bn) if pix<bn else n*cd(pix,bn))
            slots=512
        else:
            M=ci; q=oh+1
            bm=64 if M>32 else 32
            items=cd(M,bm)*n*(cd((q-1)*(q-1),128)+ (2 if q>20 else 0)) if q>20 else cd(M,bm)*n*cd(q*q,128)
            slots=512 if bm==64 else 768
    rounds=items/slots
    eff=rounds/math.ceil(rounds)
    out.append((ms*(1-eff),ms,eff,rounds,frac,s,op,mod,n,ci,co,h))
    tot+=ms; lost+=ms*(1-eff)
out.sort(reverse=True)
print("total ms",round(tot,1),"lost to round quantization (upper bound)",round(lost,1))
for o in out[:40]: print("lost %.2f of %.2f ms  eff %.2f rounds %.2f frac %.3f  s%d %s mod=%d n%d %d->%d @%d"%o)
