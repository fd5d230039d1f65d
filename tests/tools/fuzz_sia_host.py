"""Fuzz the SIA kernels (host stand-in, tests/hipcpu) against the oracle: random shapes, block counts, copy counts.\n    python tests/tools/fuzz_sia_host.py <seed> <cases>"""
import os, sys, numpy as np, torch
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, ROOT+'/oracle', ROOT+'/tests'): sys.path.insert(0,p)
import host_kernels, c_oracle as C, fgsm_oracle as O
class P:
    def setattr(self,o,n,v): setattr(o,n,v)
    def setenv(self,n,v): os.environ[n]=v
host_kernels.install(P(), tag=None, env={})
import test_zz_hip_widened as W
W.DEV='cpu'
rng=np.random.RandomState(int(sys.argv[1]) if len(sys.argv)>1 else 0)
for it in range(int(sys.argv[2]) if len(sys.argv)>2 else 40):
    n=int(rng.randint(1,3)); c=int(rng.randint(1,4)); h=int(rng.randint(9,120)); w=int(rng.randint(9,400))
    nb=int(rng.randint(1,min(8,h-1,w-1)+1)); copies=int(rng.randint(1,7))
    try:
        W.test_sia_kernels_random((n,c,h,w), nb, copies)
    except AssertionError as e:
        print('MISMATCH', (n,c,h,w), nb, copies); raise
print('done ok')
