"""CPU simulation behind DESIGN.md section 4.1 / 7: visits per mesh of the pixel-major forward on the bench scene for
(a) one 8x8 tile per wave (today), (b) four independent 4x4 quadrants per wave, (c) a 16x16 block whose 16 quadrants are
dealt to the four 16-lane groups of a wave (LPT).  Prints contributing pairs, visits and lane efficiency."""
import numpy as np, torch, sys
sys.path.insert(0,'/root/repo')
from oracle import torch_ref as TR
from umr_amd.synthetic import make_s1_inputs
B=2; H=256; IS=512
tv, faces, out, batch = make_s1_inputs(B, H, 3, seed=100, device='cpu')
verts = out['pred_vs'].detach(); cams = out['cam'].detach()
proj = TR.orthographic_proj_withz(verts, cams, 5.) * torch.tensor([1.,-1.,1.])
fv = TR.face_vertices(TR.look_at_ortho(proj), faces[None].expand(B,-1,-1)).numpy()
thr2 = np.log(1/1e-10 - 1)*1e-5; thr=np.sqrt(thr2)
xs = (2*np.arange(IS)+1-IS)/IS
def seg_d2(px,py,ax,ay,bx,by):
    ex,ey=bx-ax,by-ay; l2=ex*ex+ey*ey+1e-30
    t=np.clip(((px-ax)*ex+(py-ay)*ey)/l2,0,1)
    dx=ax+t*ex-px; dy=ay+t*ey-py
    return dx*dx+dy*dy
tot_pairs=0; visits8=0; vq=0; sumq=0
vq16=0; visits16=0  # wave owns 16x16 with 16 quadrants, 4 groups each own a 8x8's...? skip
# variant B: wave owns 16x4 strip? variant C: 4 groups own 4 quadrants of 8x8 (fixed)
# variant D: wave owns 16x16 block; group g owns quadrant rows: 4 quadrants each (sequential lists) -> visits = max over groups of sum of its 4 quadrant lists
vD=0; vE=0
for n in range(B):
    tiles = {}
    blocks = {}
    for f in range(fv.shape[1]):
        p=fv[n,f]; x=p[:,0]; y=p[:,1]
        xlo,xhi,ylo,yhi=x.min()-thr,x.max()+thr,y.min()-thr,y.max()+thr
        i0=max(int(np.floor((xlo*IS+IS-1)/2)),0); i1=min(int(np.ceil((xhi*IS+IS-1)/2)),IS-1)
        j0=max(int(np.floor((ylo*IS+IS-1)/2)),0); j1=min(int(np.ceil((yhi*IS+IS-1)/2)),IS-1)
        if i0>i1 or j0>j1: continue
        px,py=np.meshgrid(xs[i0:i1+1], xs[j0:j1+1])
        d2=np.minimum(np.minimum(seg_d2(px,py,x[0],y[0],x[1],y[1]),seg_d2(px,py,x[1],y[1],x[2],y[2])),seg_d2(px,py,x[2],y[2],x[0],y[0]))
        def cr(ax,ay,bx,by): return (bx-ax)*(py-ay)-(by-ay)*(px-ax)
        c0=cr(x[0],y[0],x[1],y[1]); c1=cr(x[1],y[1],x[2],y[2]); c2=cr(x[2],y[2],x[0],y[0])
        inside=((c0>0)&(c1>0)&(c2>0))|((c0<0)&(c1<0)&(c2<0))
        need = inside | (d2<thr2)
        jj,ii=np.nonzero(need)
        if len(ii)==0: continue
        ii=ii+i0; jj=jj+j0
        rows = IS-1-jj
        tot_pairs+=len(ii)
        qk = (rows//4)*128 + ii//4          # 4x4 quadrant key
        for k in np.unique(qk):
            tiles[int(k)] = tiles.get(int(k),0)+1
        tk = (rows//8)*64 + ii//8
        for k in np.unique(tk):
            blocks[int(k)] = blocks.get(int(k),0)+1
    visits8 += sum(blocks.values())
    sumq += sum(tiles.values())
    for k in blocks:
        ty,tx=divmod(k,64)
        ql=[tiles.get((ty*2+a)*128+tx*2+b,0) for a in (0,1) for b in (0,1)]
        vq += max(ql)
    # variant D: 16x16 block, 16 quadrants, 4 groups; greedy LPT assign quadrants to groups
    for by in range(32):
        for bx in range(32):
            ql=sorted([tiles.get((by*4+a)*128+bx*4+b,0) for a in range(4) for b in range(4)],reverse=True)
            g=[0,0,0,0]
            for q in ql:
                g[g.index(min(g))]+=q
            vD+=max(g)
            vE+=ql[0]+ql[4]+ql[8]+ql[12]   # variant E: 4 waves per block, each takes 4 quadrants of similar list length
print("pairs/mesh", tot_pairs/B)
print("8x8 visits/mesh", visits8/B, "eff", tot_pairs/(visits8*64))
print("quadrant visits total/mesh", sumq/B, "-> /4 =", sumq/B/4, "eff", tot_pairs/(sumq*16))
print("indep quadrants in 8x8 (max of 4)/mesh", vq/B, "eff", tot_pairs/(vq*64))
print("16x16 block LPT over 4 groups /mesh", vD/B, "eff", tot_pairs/(vD*64))
print("16x16 block, 4 waves, quadrants sorted by list length /mesh", vE/B, "eff", tot_pairs/(vE*64))
