"""CPU simulation behind HISTORY.md section 4.1: how tight is the conservative tile-vs-dilated-triangle test?  bbox-only
candidates, the current test, the current test + vertex-axis separating tests, and the exact need, per mesh."""
import numpy as np, torch, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import torch_ref as TR
from umr_amd.synthetic import make_s1_inputs
B=2; H=256; IS=512
tv, faces, out, batch = make_s1_inputs(B, H, 3, seed=100, device='cpu')
verts = out['pred_vs'].detach(); cams = out['cam'].detach()
proj = TR.orthographic_proj_withz(verts, cams, 5.) * torch.tensor([1.,-1.,1.])
fv = TR.face_vertices(TR.look_at_ortho(proj), faces[None].expand(B,-1,-1)).numpy().astype(np.float64)
thr2 = np.log(1/1e-10 - 1)*1e-5; thr=np.sqrt(thr2)
xs = (2*np.arange(IS)+1-IS)/IS
def seg_d2(px,py,ax,ay,bx,by):
    ex,ey=bx-ax,by-ay; l2=ex*ex+ey*ey+1e-30
    t=np.clip(((px-ax)*ex+(py-ay)*ey)/l2,0,1)
    dx=ax+t*ex-px; dy=ay+t*ey-py
    return dx*dx+dy*dy
T=8
need=0; cur=0; new=0; bboxonly=0
for n in range(B):
    for f in range(fv.shape[1]):
        p=fv[n,f]; x=p[:,0]; y=p[:,1]
        xlo,xhi,ylo,yhi=x.min()-thr,x.max()+thr,y.min()-thr,y.max()+thr
        det = x[2]*(y[0]-y[1]) + x[0]*(y[1]-y[2]) + x[1]*(y[2]-y[0])
        if abs(det)<1e-10: continue
        inv=np.array([[y[1]-y[2], x[2]-x[1], x[1]*y[2]-x[2]*y[1]],[y[2]-y[0], x[0]-x[2], x[2]*y[0]-x[0]*y[2]],[y[0]-y[1], x[1]-x[0], x[0]*y[1]-x[1]*y[0]]])/det
        K=[]
        for c in range(3):
            a,b=(c+1)%3,(c+2)%3
            K.append(det*det/max((x[a]-x[b])**2+(y[a]-y[b])**2,1e-30))
        # candidate tiles from bbox (pixel rows: row = IS-1-j)
        i0=max(int(np.floor((xlo*IS+IS-1)/2)),0); i1=min(int(np.ceil((xhi*IS+IS-1)/2)),IS-1)
        j0=max(int(np.floor((ylo*IS+IS-1)/2)),0); j1=min(int(np.ceil((yhi*IS+IS-1)/2)),IS-1)
        if i0>i1 or j0>j1: continue
        r0,r1=IS-1-j1, IS-1-j0
        for ty in range(r0//T, r1//T+1):
            for tx in range(i0//T, i1//T+1):
                # tile pixel-centre rect
                cxl,cxh=xs[tx*T], xs[min(tx*T+T-1,IS-1)]
                cyh,cyl=xs[IS-1-ty*T], xs[IS-1-min(ty*T+T-1,IS-1)]
                # bbox test like kernel
                if cxl>xhi or cxh<xlo or cyl>yhi or cyh<ylo: continue
                bboxonly+=1
                cx,cy,hx,hy=0.5*(cxl+cxh),0.5*(cyl+cyh),0.5*(cxh-cxl),0.5*(cyh-cyl)
                outp=False
                for c in range(3):
                    w=inv[c,0]*cx+inv[c,1]*cy+inv[c,2] + hx*abs(inv[c,0])+hy*abs(inv[c,1])
                    if w < -(thr/np.sqrt(K[c])) - 1e-3: outp=True
                if outp: continue
                cur+=1
                # new: vertex axes
                out2=False
                for k in range(3):
                    qx=min(max(x[k],cxl),cxh); qy=min(max(y[k],cyl),cyh)
                    ax,ay=x[k]-qx,y[k]-qy; l=np.hypot(ax,ay)
                    if l<=thr: continue
                    ax/=l; ay/=l
                    gap=min(x[0]*ax+y[0]*ay, x[1]*ax+y[1]*ay, x[2]*ax+y[2]*ay)-(qx*ax+qy*ay)
                    if gap>thr*1.001+1e-6: out2=True
                if not out2: new+=1
                # exact need
                px,py=np.meshgrid(xs[tx*T:min(tx*T+T,IS)], xs[IS-1-min(ty*T+T-1,IS-1):IS-ty*T])
                d2=np.minimum(np.minimum(seg_d2(px,py,x[0],y[0],x[1],y[1]),seg_d2(px,py,x[1],y[1],x[2],y[2])),seg_d2(px,py,x[2],y[2],x[0],y[0]))
                def cr(ax_,ay_,bx_,by_): return (bx_-ax_)*(py-ay_)-(by_-ay_)*(px-ax_)
                c0=cr(x[0],y[0],x[1],y[1]); c1=cr(x[1],y[1],x[2],y[2]); c2=cr(x[2],y[2],x[0],y[0])
                ins=((c0>0)&(c1>0)&(c2>0))|((c0<0)&(c1<0)&(c2<0))
                nd=(ins|(d2<thr2)).any()
                if nd:
                    need+=1
                    assert not out2, "new test culled a needed tile!"
print("T=%d per mesh: bbox-only %d, current test %d, +vertex axes %d, exact need %d"%(T,bboxonly/B,cur/B,new/B,need/B))
