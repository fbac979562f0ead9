#!/usr/bin/env python3
"""Wave-level statistics of the SSAO tap culling on the 4K benchmark frame (float64 restatement of Ssao.hlsl:141-191 in numpy):
how many taps the nearest-depth cells cull, how many tap pairs a wavefront still executes, with 8-texel cells, 4-texel cells and a
perfect test.  First: python tools/ssao_tap_stats.py --gen  (ray-casts the scene on the CPU, ~40 s, into /tmp/an/scene4k.npz)."""
import os, sys
import numpy as np
if "--gen" in sys.argv:
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from crychic_renderer_amd import scene
    c = scene.Constants(3840, 2160, 64)
    p = scene._camera_planes(c, torch.device("cpu"))
    os.makedirs("/tmp/an", exist_ok=True)
    np.savez("/tmp/an/scene4k.npz", depth=p["depth"].numpy(), normal=p["normal"].numpy().view(np.uint16), randvec=c.randvec,
             ssao_cb=np.frombuffer(bytes(c.ssao_cb), dtype=np.uint8))
d=np.load('/tmp/an/scene4k.npz')
depth=d['depth'].astype(np.int64)&0xFFFFFF; H,W=depth.shape
cb=d['ssao_cb'].view(np.float32)
Proj=cb[0:16]; InvProj=cb[16:32]; PT=cb[32:48]; off=cb[48:48+56].reshape(14,4)[:,:3].astype(np.float64)
R=float(cb[48+56+12+4]); eps=float(cb[48+56+12+4+3]); print('R',R,'eps',eps, 'fadeS,E',cb[48+56+12+5],cb[48+56+12+6])
A=float(Proj[10]); B=float(Proj[11])
z=depth/16777215.0
w2,h2=W//2,H//2
nrm=d['normal'].view(np.float16)[1::2,1::2,:3].astype(np.float64)
# centre depth
zc=(z[0::2,0::2]+z[1::2,0::2]+z[0::2,1::2]+z[1::2,1::2])/4
pz=B/(zc-A)
xs=(np.arange(w2)+0.5)/w2; ys=(np.arange(h2)+0.5)/h2
hx=2*xs-1; hy=1-2*ys
# PosV = InvProj applied: perspective: x*InvProj[0], y*InvProj[5]; z = InvProj[11]?; use general
def mulcol(x,y,zz,w,col): return x*col[0]+y*col[1]+zz*col[2]+w*col[3]
HX,HY=np.meshgrid(hx,hy)
ph=[mulcol(HX,HY,0,1,InvProj[4*j:4*j+4].astype(np.float64)) for j in range(4)]
PosV=[ph[0]/ph[3],ph[1]/ph[3],ph[2]/ph[3]]
t=pz/PosV[2]
p=[t*PosV[0],t*PosV[1],t*PosV[2]]
nl=np.sqrt((nrm**2).sum(-1)); n=nrm/np.maximum(nl,1e-30)[...,None]
# randvec bilinear wrap at 4uv
rv8=d['randvec'][...,:3].astype(np.float64)/255.0
U,V=np.meshgrid(4*xs,4*ys)
tx=(U-np.floor(U))*256-0.5; ty=(V-np.floor(V))*256-0.5
i0=np.floor(tx).astype(int); j0=np.floor(ty).astype(int); fx=(tx-i0)[...,None]; fy=(ty-j0)[...,None]
def T(i,j): return rv8[j&255,i&255]
rv=(T(i0,j0)*(1-fx)+T(i0+1,j0)*fx)*(1-fy)+(T(i0,j0+1)*(1-fx)+T(i0+1,j0+1)*fx)*fy
rv=2*rv-1
sky=(depth[0::2,0::2]==0xFFFFFF)&(depth[1::2,0::2]==0xFFFFFF)&(depth[0::2,1::2]==0xFFFFFF)&(depth[1::2,1::2]==0xFFFFFF)
# padded depth + cell mins (9x9 stride 8 over padded coords)
zp=np.ones((H+4+16,W+4+16)); zp[2:2+H,2:2+W]=z
def cellmin(cs, ov):
    ch=(H+4+cs-1)//cs; cw=(W+4+cs-1)//cs
    out=np.full((ch,cw),np.inf)
    for dy in range(cs+ov):
        for dx in range(cs+ov):
            out=np.minimum(out, zp[dy:dy+ch*cs:cs, dx:dx+cw*cs:cs][:ch,:cw])
    return out
cm8=cellmin(8,1); cm4=cellmin(4,1)
def toview(zz): return B/(zz-A)
surv=np.zeros((14,h2,w2),bool); true_occ=np.zeros((14,h2,w2),bool); surv4=np.zeros((14,h2,w2),bool)
zfull=np.ones((H+6,W+6)); zfull[3:3+H,3:3+W]=z   # border 1 padded by 3
for i in range(14):
    o=off[i]
    dt=(rv*o).sum(-1)
    offs=o[None,None,:]-2*dt[...,None]*rv
    s=np.sign((offs*n).sum(-1))
    q=[p[k]+s*R*offs[...,k] for k in range(3)]
    u=(q[0]*PT[0]+q[2]*PT[2])/q[2]; v=(q[1]*PT[5]+q[2]*PT[6])/q[2]
    txx=u*W-0.5; tyy=v*H-0.5
    ii=np.clip(np.floor(txx),-2,W).astype(int); jj=np.clip(np.floor(tyy),-2,H).astype(int)
    ffx=np.clip(txx-np.floor(txx),0,1); ffy=np.clip(tyy-np.floor(tyy),0,1)
    ffx=np.nan_to_num(ffx); ffy=np.nan_to_num(ffy)
    c8=cm8[(jj+2)>>3,(ii+2)>>3]; c4=cm4[(jj+2)>>2,(ii+2)>>2]
    thr=pz-eps
    cull8=(toview(c8-2**-21)*0.999998>=thr)&(q[2]>=1e-3)
    cull4=(toview(c4-2**-21)*0.999998>=thr)&(q[2]>=1e-3)
    surv[i]=~cull8&~sky; surv4[i]=~cull4&~sky
    zz=(zfull[jj+3,ii+3]*(1-ffx)+zfull[jj+3,ii+4]*ffx)*(1-ffy)+(zfull[jj+4,ii+3]*(1-ffx)+zfull[jj+4,ii+4]*ffx)*ffy
    rz=toview(zz); rr=rz/q[2]; r2=rr*q[2]
    true_occ[i]=((pz-r2)>eps)&~sky
nsky=(~sky).sum()
print('non-sky px',nsky,'of',sky.size)
print('survive8 frac of taps', surv.sum()/(14*nsky), 'survive4', surv4.sum()/(14*nsky),'true occluding', true_occ.sum()/(14*nsky))
# wave-level: waves = 64 px rows segments
def wave(a): # a: (h2,w2) bool -> any over 64-lane segments
    return a.reshape(h2,w2//64,64).any(-1)
wsky=sky.reshape(h2,w2//64,64).all(-1)
nw=(~wsky).sum(); print('non-sky waves',nw,'of',wsky.size)
for name,S in (('cell8',surv),('cell4',surv4),('true',true_occ)):
    pairs=sum(wave(S[i]|S[i+1]).sum() for i in range(0,14,2))
    anyw=wave(S.any(0)).sum()
    print(name,'pairs executed per non-sky wave',pairs/nw,' waves with any survivor',anyw/nw, ' lane-items per exec pair', S.sum()/max(pairs,1))
