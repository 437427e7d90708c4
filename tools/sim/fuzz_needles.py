"""CPU fuzz of the pair-geometry model (eval_pair_model.py) on needles, sub-pixel faces and other ill-conditioned shapes against
the oracle: > 1e-4 alpha differences and non-finite values for (old) the pick-first formulation without the non-finite skip,
(new) with it (= the shipped kernel), (+exact3) with the reference's evaluate-all-three rule on faces flagged |den| < 1e-5."""
import sys, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import eval_pair_model as M
from oracle import softras
f32=np.float32
IS=48; n=1500
rng=np.random.default_rng(7)
px=((2*np.arange(IS)+1-IS)/IS).astype(f32)
r=lambda *s: rng.uniform(-1,1,s).astype(f32)
a=r(n,2); pc=px[rng.integers(0,IS,(n,2))]
th=rng.uniform(0,2*np.pi,n).astype(f32)
u=np.stack([np.cos(th),np.sin(th)],1).astype(f32); v=np.stack([-np.sin(th),np.cos(th)],1).astype(f32)
cases={
 "needle_rot": np.stack([pc - 2e-5*u - 1e-5*v, pc + 2e-5*u - 1e-5*v, pc + (0.03+0.05*np.abs(r(n,1)))*v + 0.01*r(n,1)*u],1),
 "needle_rot_wide": np.stack([pc - 2e-4*u - 1e-4*v, pc + 2e-4*u - 1e-4*v, pc + (0.03+0.05*np.abs(r(n,1)))*v],1),
 "two_collapsed": np.stack([pc, pc + 2e-5*u, pc + 3e-5*v],1),
 "big_coords": np.stack([a*20, a*20 + r(n,2), a*20 + r(n,2)],1),
 "edge_on_pixel_row": np.stack([np.stack([r(n), pc[:,1]],1), np.stack([r(n), pc[:,1]],1), r(n,2)],1),
 "right_angle_tiny": np.stack([pc, pc + np.array([1e-3,0],f32), pc + np.array([0,1e-3],f32)],1),
}
cases={k:np.concatenate([v_.astype(f32), np.full((n,3,1),7.7,f32)],2) for k,v_ in cases.items()}
sigma=f32(1e-5); del_=f32(np.log(1./1e-10-1.)); threshold=f32(del_*sigma); thr=f32(np.sqrt(threshold)); nis=f32(-1.0/sigma)
xi,yi=np.meshgrid(np.arange(IS),np.arange(IS)); xp=px[xi.ravel()]; yp=px[(IS-1-yi).ravel()]
cfg=dict(near=1.,far=100.,eps=1e-3,sigma_val=float(sigma),dist_eps_log=float(del_),gamma_val=1e-4,func_id_rgb=1,double_side=True)
for name,fv in cases.items():
    ref=softras.raster_forward(fv.reshape(n,1,9), np.ones((n,1,1,3),f32), IS, background=(0,0,0), backend="port", n_threads=8, **cfg)
    ra=ref["soft_colors"][:,3].reshape(n,-1)
    rec=M.face_setup(fv)
    res=[]
    for fb in (False,True,"x3"):
        live,frag,nf=M.eval_pair(rec,xp,yp,thr,threshold,nis,fallback=bool(fb),exact3_below=(1e-5 if fb=="x3" else 0.0))
        al=np.where(live,frag,f32(0)); bad=~np.isfinite(al)
        err=np.abs(np.where(bad,1.0,al.astype(np.float64))-ra)
        res.append((int((err>1e-4).sum()), int(bad.sum())))
    print("%-20s live %6d | old: >1e-4 %4d nonfinite %4d | new: >1e-4 %4d nonfinite %4d | +exact3: >1e-4 %4d nonfinite %4d"%(name, (ra>0).sum(), res[0][0],res[0][1],res[1][0],res[1][1],res[2][0],res[2][1]))
