"""CPU simulation behind DESIGN.md section 4.1: packing several faces whose quadrant footprints in an 8x8 tile are disjoint
into one visit (in-order greedy) -- 20 081 -> 19 430 visits per mesh: not worth building."""
import numpy as np, torch, sys
sys.path.insert(0,'/root/repo')
from oracle import torch_ref as TR
from umr_amd.synthetic import make_s1_inputs
B=2; H=256; IS=512
tv, faces, out, batch = make_s1_inputs(B, H, 3, seed=100, device='cpu')
verts = out['pred_vs'].detach(); cams = out['cam'].detach()
proj = TR.orthographic_proj_withz(verts, cams, 5.) * torch.tensor([1.,-1.,1.])
fv = TR.face_vertices(TR.look_at_ortho(proj), faces[None].expand(B,-1,-1)).numpy()  # [B,F,3,3]
thr2 = np.log(1/1e-10 - 1)*1e-5; thr=np.sqrt(thr2)
xs = (2*np.arange(IS)+1-IS)/IS
def seg_d2(px,py,ax,ay,bx,by):
    ex,ey=bx-ax,by-ay; l2=ex*ex+ey*ey+1e-30
    t=np.clip(((px-ax)*ex+(py-ay)*ey)/l2,0,1)
    dx=ax+t*ex-px; dy=ay+t*ey-py
    return dx*dx+dy*dy
tot_pairs=0; visits8=0; visits_pack=0; lanes8=0; hist=np.zeros(5,int)
visits_pack16=0
for n in range(B):
    # per tile list of (face, quadmask, npix)
    tiles = {}
    for f in range(fv.shape[1]):
        p=fv[n,f]; x=p[:,0]; y=p[:,1]
        xlo,xhi,ylo,yhi=x.min()-thr,x.max()+thr,y.min()-thr,y.max()+thr
        i0=max(int(np.floor((xlo*IS+IS-1)/2)),0); i1=min(int(np.ceil((xhi*IS+IS-1)/2)),IS-1)
        j0=max(int(np.floor((ylo*IS+IS-1)/2)),0); j1=min(int(np.ceil((yhi*IS+IS-1)/2)),IS-1)
        if i0>i1 or j0>j1: continue
        px,py=np.meshgrid(xs[i0:i1+1], xs[j0:j1+1])
        d2=np.minimum(np.minimum(seg_d2(px,py,x[0],y[0],x[1],y[1]),seg_d2(px,py,x[1],y[1],x[2],y[2])),seg_d2(px,py,x[2],y[2],x[0],y[0]))
        # inside test
        def cr(ax,ay,bx,by): return (bx-ax)*(py-ay)-(by-ay)*(px-ax)
        c0=cr(x[0],y[0],x[1],y[1]); c1=cr(x[1],y[1],x[2],y[2]); c2=cr(x[2],y[2],x[0],y[0])
        inside=((c0>0)&(c1>0)&(c2>0))|((c0<0)&(c1<0)&(c2<0))
        need = inside | (d2<thr2)
        jj,ii=np.nonzero(need)
        if len(ii)==0: continue
        ii=ii+i0; jj=jj+j0
        rows = IS-1-jj
        tot_pairs+=len(ii)
        tx=ii//8; ty=rows//8; q=((rows%8)//4)*2+((ii%8)//4)
        # finer 2x2-subtile (16 per tile) mask
        q16=((rows%8)//2)*4+((ii%8)//2)
        key=ty*64+tx
        for k in np.unique(key):
            m=key==k
            qm=0
            for qq in np.unique(q[m]): qm|=1<<int(qq)
            qm16=0
            for qq in np.unique(q16[m]): qm16|=1<<int(qq)
            tiles.setdefault(int(k),[]).append((f,qm,int(m.sum()),qm16))
    for k,lst in tiles.items():
        visits8+=len(lst)
        occ=0; v=0; occ16=0; v16=0
        for f,qm,npx,qm16 in lst:   # faces ascending
            hist[bin(qm).count('1')]+=1
            if occ & qm: v+=1; occ=qm
            else:
                if occ==0: v+=1
                occ|=qm
            if occ16 & qm16: v16+=1; occ16=qm16
            else:
                if occ16==0: v16+=1
                occ16|=qm16
        visits_pack+=v; visits_pack16+=v16
print("pairs/mesh", tot_pairs/B, "visits8/mesh", visits8/B, "lane eff", tot_pairs/(visits8*64))
print("packed visits (4x4 quadrant masks)/mesh", visits_pack/B, "eff", tot_pairs/(visits_pack*64))
print("packed visits (2x2 masks)/mesh", visits_pack16/B, "eff", tot_pairs/(visits_pack16*64))
print("quadrants needed hist", hist)
