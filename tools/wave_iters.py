import sys, numpy as np, torch
sys.path.insert(0,'.')
import bench
dev=torch.device('cuda:0')
wl=bench.Workload(dev,0,1)
out=wl.step(); torch.cuda.synchronize()
r=wl.renderer
vs,ws=r.last
from dreammesh4d_amd import _lib
L=_lib.lib()
B=vs.B; stride=L.dm4d_views_geom_bytes(1,r.N,512,512)
geom=ws["geom"].cpu().numpy()
T=1024
def al(x): return (x+255)//256*256
off=256; tc=off; off=al(off+T*4); ts=off; off=al(off+(T+1)*4); cc=off; off=al(off+T*64); cd=off; off=al(off+T*64); ck=off
tot_f=tot_b=0; tot_pairs=0; tot_cons=0
for b in range(B):
    g=geom[b*stride:(b+1)*stride]
    ccount=g[cc:cc+T*64].view(np.uint32).reshape(T,4,4)
    cdone=g[cd:cd+T*64].view(np.uint32).reshape(T,4,4)
    D=g[0:4].view(np.uint32)[0]; R=g[8:12].view(np.uint32)[0]
    tot_b+=cdone.max(axis=2).sum(); tot_f+=ccount.max(axis=2).sum(); tot_pairs+=ccount.sum(); tot_cons+=cdone.sum()
    if b==0: print("view0 D",D,"R",R,"cell entries",ccount.sum(),"consumed",cdone.sum(), "bwd wave-iters", cdone.max(axis=2).sum(), "fwd upper", ccount.max(axis=2).sum())
print("per step: bwd wave-iters",tot_b,"fwd upper bound",tot_f,"cell entries",tot_pairs,"consumed",tot_cons)
allmax=[]
for b in range(B):
    g=geom[b*stride:(b+1)*stride]
    cdone=g[cd:cd+T*64].view(np.uint32).reshape(T,4,4)
    allmax.append(cdone.max(axis=2).reshape(-1))
m=np.concatenate(allmax)
print("waves",m.size,"empty",(m==0).sum(),"mean",m.mean(),"p50",np.percentile(m,50),"p90",np.percentile(m,90),"p99",np.percentile(m,99),"max",m.max())
srt=np.sort(m)[::-1]; print("top10",srt[:10], "sum top 1024 / total", srt[:1024].sum()/m.sum())
