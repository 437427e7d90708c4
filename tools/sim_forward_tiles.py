"""Work count (CPU, numpy): (tile, face) visits of the pixel-major forward on the SURVEY 8d scene for other hand-outs than one 8x8 tile per
wave -- two 8x4 half-waves or four 4x4 quarter-waves, each walking its OWN face list (a visit then carries 2 / 4 faces; the wave makes
max-over-parts visits).  Result (2 meshes, 1280 faces, IS 512): 8x8 26 458 / 20 705 visits per mesh; half-waves 25 297 / 19 590 (-4 %);
quarter-waves 24 127 / 18 454 (-9 %): an 18-pixel face spans the parts of an 8x8 tile alike, the lists barely shrink -- not worth
giving up wave-uniform face records.  usage: python tools/sim_forward_tiles.py"""
import sys, numpy as np, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import scene
from oracle import torch_ref as TR
IS=512; thr=np.sqrt(np.log(1e10-1)*1e-5)   # band (outside); inside: all pixels inside count
verts, faces, cams, gen = scene(2, 3, seed=0)
proj = TR.orthographic_proj_withz(verts, cams, 5.) * torch.tensor([1., -1., 1.])
fv = TR.face_vertices(TR.look_at_ortho(proj), faces).numpy()   # [N,F,3,3]
def pix(i): return (2*i+1-IS)/IS
res={}
for n in range(2):
    inband = {}   # (ty,tx) for 4x4 subtiles -> set of faces, built from per-pixel test
    # per-pixel membership at 4x4 granularity is enough: store set per 4x4 tile
    T4=IS//4
    tiles4=[set() for _ in range(T4*T4)]
    lanes4 = np.zeros(T4*T4, int)
    for f in range(fv.shape[1]):
        p = fv[n,f,:,:2]
        lo = p.min(0)-thr; hi = p.max(0)+thr
        x0=max(int(np.floor((lo[0]*IS+IS-1)/2)),0); x1=min(int(np.ceil((hi[0]*IS+IS-1)/2)),IS-1)
        y0=max(int(np.floor((lo[1]*IS+IS-1)/2)),0); y1=min(int(np.ceil((hi[1]*IS+IS-1)/2)),IS-1)
        if x0>x1 or y0>y1: continue
        xs=pix(np.arange(x0,x1+1)); ys=pix(np.arange(y0,y1+1))
        X,Y=np.meshgrid(xs,ys)
        P=np.stack([X,Y],-1)
        # distance to triangle
        def seg(a,b):
            ab=b-a; t=np.clip(((P-a)@ab)/max(ab@ab,1e-30),0,1); q=a+t[...,None]*ab; return ((P-q)**2).sum(-1)
        d2=np.minimum(np.minimum(seg(p[0],p[1]),seg(p[1],p[2])),seg(p[2],p[0]))
        def cross(a,b,c): return (b[0]-a[0])*(c[...,1]-a[1])-(b[1]-a[1])*(c[...,0]-a[0])
        c0=cross(p[0],p[1],P); c1=cross(p[1],p[2],P); c2=cross(p[2],p[0],P)
        inside=((c0>0)&(c1>0)&(c2>0))|((c0<0)&(c1<0)&(c2<0))
        live = inside | (d2 < thr*thr)
        ys_i, xs_i = np.nonzero(live)
        rows = IS-1-(ys_i+y0); cols = xs_i+x0
        t4 = (rows//4)*T4 + cols//4
        for t in np.unique(t4): tiles4[t].add(f)
    # aggregate
    def visits(th, tw):   # tile of th x tw pixels (multiples of 4)
        a, b = th//4, tw//4
        tot=0; per=[]
        for ty in range(0,T4,a):
            for tx in range(0,T4,b):
                s=set()
                for yy in range(a):
                    for xx in range(b): s|=tiles4[(ty+yy)*T4+tx+xx]
                tot+=len(s)
        return tot
    v88=visits(8,8); v84=visits(4,8); v48=visits(8,4); v44=visits(4,4)
    # half-wave pairing: per 8x8 tile, max over its two 8-wide x 4-high halves
    def paired(th_half, horizontal=True):
        tot=0
        for ty in range(0,T4,2):
            for tx in range(0,T4,2):
                if horizontal:  # halves = top rows (4 high, 8 wide) and bottom
                    h0=set(); h1=set()
                    for xx in range(2): h0|=tiles4[ty*T4+tx+xx]; h1|=tiles4[(ty+1)*T4+tx+xx]
                else:
                    h0=set(); h1=set()
                    for yy in range(2): h0|=tiles4[(ty+yy)*T4+tx]; h1|=tiles4[(ty+yy)*T4+tx+1]
                tot+=max(len(h0),len(h1))
        return tot
    def quad4():   # four 4x4 quarter-waves, each its own list: visits = max of 4
        tot=0
        for ty in range(0,T4,2):
            for tx in range(0,T4,2):
                tot+=max(len(tiles4[(ty+a)*T4+tx+b]) for a in range(2) for b in range(2))
        return tot
    print("mesh",n,"visits 8x8",v88,"| sum over 8x4 halves",v84,"paired(max) horiz",paired(4,True),"vert",paired(4,False),"| 4x4 sum",v44,"quarter-wave max",quad4())
