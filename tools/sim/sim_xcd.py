"""CPU simulation behind DESIGN.md section 4.2: per-XCD load of the face-major texel-gradient backward at N = 16 (cost model:
setup + visits x 914 cycles for front faces) and a processor-sharing SIMD model of the launch for several start orders."""
import numpy as np, torch, sys
sys.path.insert(0,'/root/repo')
from oracle import torch_ref as TR
from umr_amd.synthetic import make_s1_inputs
B=16; H=256; IS=512
tv, faces, out, batch = make_s1_inputs(B, H, 3, seed=100, device='cpu')
verts = out['pred_vs'].detach(); cams = out['cam'].detach()
proj = TR.orthographic_proj_withz(verts, cams, 5.) * torch.tensor([1.,-1.,1.])
fv = TR.face_vertices(TR.look_at_ortho(proj), faces[None].expand(B,-1,-1)).numpy()
thr2 = np.log(1/1e-10 - 1)*1e-5; thr=np.sqrt(thr2)
F=fv.shape[1]
cost=np.zeros((B,F))
for n in range(B):
    for f in range(F):
        p=fv[n,f]; x=p[:,0]; y=p[:,1]
        front = (y[2]-y[0])*(x[1]-x[0]) < (y[1]-y[0])*(x[2]-x[0])
        xlo,xhi,ylo,yhi=x.min()-thr,x.max()+thr,y.min()-thr,y.max()+thr
        i0=max(int(np.floor((xlo*IS+IS-1)/2)),0); i1=min(int(np.ceil((xhi*IS+IS-1)/2)),IS-1)
        j0=max(int(np.floor((ylo*IS+IS-1)/2)),0); j1=min(int(np.ceil((yhi*IS+IS-1)/2)),IS-1)
        if i0>i1 or j0>j1: cost[n,f]=800; continue
        nt=((i1//4)-(i0//4)+1)*((j1//4)-(j0//4)+1)
        # triangle-ish: about 60% of bbox subtiles survive the cull
        cost[n,f] = 1500 + (np.ceil(0.6*nt/4)*914 if front else 0)
print("front frac", (cost>1500).mean(), "mean cost", cost.mean())
per=F//8
x = cost.reshape(B,8,per).sum(2)   # [mesh, eighth]
print("per-mesh eighth cost / mean:", np.round(x[0]/x[0].mean(),2))
for N in (16,):
    tot = x[:N].sum(0)
    print("N=%d XCD totals / mean:"%N, np.round(tot/tot.mean(),3), "max/mean", tot.max()/tot.mean())

import heapq
def sim_xcd(costs, nsimd=128, slots=7):
    """processor-sharing SIMDs, in-order dispatch to the first free slot; returns makespan (cycles)"""
    costs=list(costs); q=0
    simd=[[] for _ in range(nsimd)]   # remaining work per wave
    # initial fill round robin
    for s in range(slots):
        for i in range(nsimd):
            if q<len(costs): simd[i].append(costs[q]); q+=1
    t=0.0
    while True:
        # next completion: for each simd, min remaining * nwaves
        best=None
        for i,w in enumerate(simd):
            if w:
                dt=min(w)*len(w)
                if best is None or dt<best[0]: best=(dt,i)
        if best is None: break
        dt,_=best; t+=dt
        for i,w in enumerate(simd):
            if w:
                dec=dt/len(w)
                nw=[r-dec for r in w]
                done=[r for r in nw if r<=1e-6]
                nw=[r for r in nw if r>1e-6]
                for _ in done:
                    if q<len(costs): nw.append(costs[q]); q+=1
                simd[i]=nw
    return t
N=16
ideal = cost[:N].sum()/ (8*128)
def run(order_fn,label):
    ms=[]
    for xcd in range(8):
        ms.append(sim_xcd(order_fn(xcd)))
    print("%-40s makespan %.0f  (ideal %.0f, ratio %.3f)  per-XCD ratio %s"%(label,max(ms),ideal,max(ms)/ideal,np.round(np.array(ms)/ideal,2)))
cur=lambda x: [cost[n, x*per+k] for n in range(N) for k in range(per)]
run(cur,"current (index order, eighths)")
hf=lambda x: [c for n in range(N) for c in sorted(cost[n, x*per:(x+1)*per], reverse=True)]
run(hf,"heavy-first within (mesh, eighth)")
ghf=lambda x: sorted([cost[n, x*per+k] for n in range(N) for k in range(per)], reverse=True)
run(ghf,"global heavy-first within XCD")
# balanced: all faces of all meshes sorted heavy-first dealt round-robin to XCDs (no locality)
allc=sorted(cost[:N].ravel(), reverse=True)
run(lambda x: allc[x::8],"global LPT deal (no locality)")
