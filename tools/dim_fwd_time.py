import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transferattack_amd import _hip
_hip.load()
for n in (32, 160):
    xs = [torch.rand(n, 3, 224, 224, device="cuda") for _ in range(3)]
    y = torch.empty_like(xs[0])
    for i in range(4):
        _hip.dim_fwd(xs[i % 3], y, 246, 237, 3, 5)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for i in range(20):
        _hip.dim_fwd(xs[i % 3], y, 246, 237, 3, 5)
    e.record(); torch.cuda.synchronize()
    print("variant", os.environ.get("TA_DIM_FWD_VARIANT", "0"), "n", n, "dim_fwd us", round(s.elapsed_time(e) * 1e3 / 20, 2))
