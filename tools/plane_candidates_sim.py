"""How many of random_scene's small spheres does a one-plane test keep per closest-hit query?  (the gate for the plane
screen.)  A numpy path tracer that only approximates the materials -- it needs the distribution of rays, not the image -- over
the library's own random_scene: vertical plane through the ray, the tilted plane through the ray, the better of the two, vertical
plane + forward half-space, and all three (a proxy for the full test).  CPU only: python tools/plane_candidates_sim.py"""
import sys, importlib, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tor = importlib.import_module('trace-of-radiance_amd')
recs = tor.random_scene().to_records()
n = len(recs)
rng = np.random.default_rng(1)
kind = recs[:,0].astype(int); c0 = recs[:,1:4]; c1 = recs[:,4:7]; rad = recs[:,9]; mat = recs[:,10].astype(int)
small = np.abs(rad-0.2) < 1e-9
print("objects", n, "small", small.sum(), "movers", (kind!=0).sum(), "kinds", np.unique(kind), "mats", np.bincount(mat))
# camera
lf = np.array([13.,2,3]); la = np.zeros(3); vup = np.array([0,1.,0]); vfov=20; aspect=16/9; ap=0.1; fd=10
h = np.tan(np.radians(vfov)/2); vh = 2*h; vw = aspect*vh
w = (lf-la)/np.linalg.norm(lf-la); u = np.cross(vup,w); u/=np.linalg.norm(u); v = np.cross(w,u)
hor = fd*vw*u; ver = fd*vh*v; llc = lf - hor/2 - ver/2 - fd*w
N = 40000
s = rng.random(N); t = rng.random(N)
O = np.repeat(lf[None],N,0); D = llc + s[:,None]*hor + t[:,None]*ver - lf
T = rng.random(N)
def unit(x): return x/np.linalg.norm(x,axis=1,keepdims=True)
stats = []
def closest(O,D,T):
    C = c0[None] + T[:,None,None]*(c1-c0)[None]   # time0=0,time1=1
    oc = O[:,None,:]-C
    a = (D*D).sum(1)[:,None]; hb = (oc*D[:,None,:]).sum(2); cc=(oc*oc).sum(2)-rad[None]**2
    disc = hb*hb-a*cc
    sq = np.sqrt(np.maximum(disc,0))
    t1 = (-hb-sq)/a; t2=(-hb+sq)/a
    tt = np.where((disc>0)&(t1>1e-3),t1,np.where((disc>0)&(t2>1e-3),t2,np.inf))
    idx = tt.argmin(1); tm = tt[np.arange(len(O)),idx]
    return idx, tm, C, disc>0
def planes(O,D,T):
    # stage-1 candidate counts among small spheres
    m = small
    cx = c0[m,0][None]; cz = c0[m,2][None]
    cy = (c0[m,1][None] + T[:,None]*(c1[m,1]-c0[m,1])[None])
    R = 0.2
    d = unit(D)
    dxz = np.sqrt(d[:,0]**2+d[:,2]**2)+1e-300
    nx = -d[:,2]/dxz; nz = d[:,0]/dxz
    sV = nx[:,None]*(cx-O[:,0:1]) + nz[:,None]*(cz-O[:,2:3])
    V = np.abs(sV)<=R
    # tilted plane: normal n2 = d x hperp, hperp=(nx,0,nz)
    hp = np.stack([nx,np.zeros_like(nx),nz],1)
    n2 = np.cross(d,hp); n2 = unit(n2)
    sT = n2[:,0:1]*(cx-O[:,0:1]) + n2[:,1:2]*(cy-O[:,1:2]) + n2[:,2:3]*(cz-O[:,2:3])
    Tt = np.abs(sT)<=R
    # forward half-space
    u_ = d[:,0:1]*(cx-O[:,0:1]) + d[:,1:2]*(cy-O[:,1:2]) + d[:,2:3]*(cz-O[:,2:3])
    F = u_ >= -R
    full = V & Tt & F   # ~ true (perp dist<=R√2 box) proxy
    return V.sum(1), Tt.sum(1), np.minimum(V.sum(1),Tt.sum(1)), (V&F).sum(1), full.sum(1)
depth=0
allV=[];allT=[];allM=[];allVF=[];allFull=[]
while len(O)>0 and depth<50:
    idx,tm,C,dpos = closest(O,D,T)
    v_,t_,m_,vf_,f_ = planes(O,D,T)
    allV.append(v_);allT.append(t_);allM.append(m_);allVF.append(vf_);allFull.append(f_)
    hit = np.isfinite(tm)
    O2 = O[hit]+tm[hit,None]*D[hit]; i2=idx[hit]; Cc=C[hit,i2]; nrm=(O2-Cc)/rad[i2][:,None]; D2=D[hit]; T2=T[hit]
    front = (D2*nrm).sum(1)<0; nrm = np.where(front[:,None],nrm,-nrm)
    mk = mat[i2]
    r = unit(rng.normal(size=(len(O2),3)))
    lam = nrm + r
    ud = unit(D2); refl = ud-2*(ud*nrm).sum(1)[:,None]*nrm
    fz = recs[i2,14][:,None]
    met = refl + fz*r*rng.random((len(O2),1))**(1/3)
    # dielectric: coin flip reflect / straight through (approximation)
    die = np.where(rng.random((len(O2),1))<0.3, refl, ud)
    newD = np.where((mk==0)[:,None],lam,np.where((mk==1)[:,None],met,die))
    keep = ~((mk==1)&((newD*nrm).sum(1)<=0))
    # russian: lambertian always continue
    O=O2[keep];D=newD[keep];T=T2[keep]
    depth+=1
V=np.concatenate(allV);Tt=np.concatenate(allT);M=np.concatenate(allM);VF=np.concatenate(allVF);Fu=np.concatenate(allFull)
print("queries",len(V),"per primary",len(V)/N)
for name,a in (("vertical",V),("tilted",Tt),("min",M),("vertical+fwd",VF),("all3",Fu)):
    # wave stats: random groups of 64
    k = len(a)//64*64; g = a[:k].reshape(-1,64)
    print(f"{name:14s} mean {a.mean():6.2f}  p50 {np.median(a):5.1f} p99 {np.percentile(a,99):5.1f} max {a.max():4d}  mean wave-max {g.max(1).mean():6.2f}")
