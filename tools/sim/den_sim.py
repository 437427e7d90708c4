"""CPU emulation behind DESIGN.md section 5 (the collapsed-edge NaN): how often the reference's fp32 `den` of
soft_rasterize_cuda_kernel.cu:82-86 is exactly 0 / negative for an edge of a given screen-space length."""
import numpy as np
f32=np.float32
rng=np.random.default_rng(0)
# needle faces: A, B = A + tiny, C far.  How often is den (edge A-B) exactly 0 / negative / garbage?
def den_edge(pa, pb, pc):
    px=[pa[0],pb[0],pc[0]]; py=[pa[1],pb[1],pc[1]]
    sym=[[f32(f32(f32(px[j]*px[k])+f32(py[j]*py[k]))+f32(1)) for k in range(3)] for j in range(3)]
    e,e1=0,1
    a=[f32(sym[e][j]-sym[e1][j]) for j in range(3)]
    return f32(a[e]-a[e1]), a
for eps in (1e-3,3e-4,1e-4,3e-5,1e-5):
    zero=neg=0; n=20000; rel=[]
    for _ in range(n):
        A=rng.uniform(-0.8,0.8,2).astype(f32); d=rng.normal(size=2); d=d/np.linalg.norm(d)*eps
        B=(A+d.astype(f32)).astype(f32); C=rng.uniform(-0.8,0.8,2).astype(f32)
        den,_=den_edge(A,B,C)
        true=float((np.float64(A[0])-np.float64(B[0]))**2+(np.float64(A[1])-np.float64(B[1]))**2)
        zero+= den==0; neg+= den<0
        rel.append(abs(float(den)-true)/true)
    print("edge %.0e: den==0 %.3f  den<0 %.3f  median rel err %.2g"%(eps,zero/n,neg/n,np.median(rel)))
