"""Fuzz the lane-per-column DIM kernels (host stand-in, tests/hipcpu) against the C oracle: random sizes, rates, geometries.\n    python tests/tools/fuzz_dim_host.py <seed> <cases>"""
import os, sys, numpy as np, torch
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, ROOT+'/oracle', ROOT+'/tests'): sys.path.insert(0,p)
import host_kernels, c_oracle as C
class P:
    def setattr(self,o,n,v): setattr(o,n,v)
    def setenv(self,n,v): os.environ[n]=v
host_kernels.install(P(), tag='fuzz', env={})
from transferattack_amd import _hip
rng=np.random.RandomState(int(sys.argv[1]) if len(sys.argv)>1 else 0)
bad=0
for it in range(int(sys.argv[2]) if len(sys.argv)>2 else 150):
    size=int(rng.randint(8,301)); rate=float(rng.uniform(1.01,1.49)); resize=int(size*rate)
    if resize<=size: continue
    rnd=int(rng.randint(size,resize)); rem=resize-rnd
    top=int(rng.randint(0,rem+1)) if rem>0 else 0; left=int(rng.randint(0,rem+1)) if rem>0 else 0
    x=torch.rand(1,2,size,size).numpy(); gy=torch.randn(1,2,size,size).numpy()
    y=torch.empty(1,2,size,size); gx=torch.empty(1,2,size,size)
    _hip.dim_fwd(torch.from_numpy(x),y,resize,rnd,top,left); _hip.dim_bwd(torch.from_numpy(gy),gx,resize,rnd,top,left)
    geom=(True,rnd,top,left)
    ok1=np.array_equal(y.numpy(),C.dim_fwd(x,geom,resize)); ok2=np.array_equal(gx.numpy(),C.dim_bwd(gy,geom,resize))
    if not(ok1 and ok2):
        bad+=1; print('MISMATCH',size,resize,rnd,top,left,ok1,ok2)
print('done, mismatches:',bad)
