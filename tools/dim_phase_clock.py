#!/usr/bin/env python
"""Where do a DIM tile's cycles go?  Builds csrc/dim.hip with -DTA_DIM_PHASE_CLOCK into tools/bin/libta_dim_phase_clock.so (a
tuning build, NOT libta_hip.so: thread 0 of every workgroup of the two lane-per-column kernels adds the shader-clock cycles
between consecutive barriers to a device-side table), runs the forward and the backward at N = 32 and 160 and prints the mean
cycles per workgroup and phase.  Round 4 ended with two rejected hypotheses about these kernels (VALU issue; vector-memory
instruction issue -- DESIGN.md section 5); this is the measurement that decides the next one.

    python tools/dim_phase_clock.py            # on the GPU box; ~20 s
"""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "transferattack_amd", "csrc")
LIB = os.path.join(ROOT, "tools", "bin", "libta_dim_phase_clock.so")
FWD = ("taps of the second resample (ty2 / tx2)", "taps of the first resample (ty1)", "H1: global gather of x + width pass -> T",
       "V1: height pass -> mid", "H2: width pass -> u", "V2: height pass + store")
BWD = ("hit tables of stage B", "hit tables of stage A", "stage A: global gather of gy -> mid", "stage B: LDS gather -> gx store",
       "tile sum of |gx|")


def build():
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    srcs = [os.path.join(CSRC, f) for f in ("dim.hip", "runtime.hip")]
    if os.path.isfile(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(s) for s in srcs):
        return
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                    "-fno-fast-math", "-Wno-unused-function", "-DTA_DIM_PHASE_CLOCK", "-shared"] + srcs + ["-o", LIB], check=True)


def main():
    build()
    lib = ctypes.CDLL(LIB)
    vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
    lib.ta_dim_fwd.argtypes = [vp, vp, i64, i32, i32, i32, i32, i32, vp]
    lib.ta_dim_bwd.argtypes = [vp, vp, vp, i64, i32, i32, i32, i32, i32, vp]
    lib.ta_dim_phase_clock_read.argtypes = [vp, vp]
    cycles, groups = (ctypes.c_ulonglong * 16)(), (ctypes.c_ulonglong * 2)()
    size, resize, rnd, top, left = 224, 246, 237, 3, 5
    for n in (32, 160):
        x = torch.randn(n, 3, size, size, device="cuda")
        y = torch.empty_like(x)
        planes = n * 3
        lib.ta_dim_phase_clock_read(cycles, groups)                                  # reset
        for _ in range(5):
            assert lib.ta_dim_fwd(x.data_ptr(), y.data_ptr(), planes, size, resize, rnd, top, left, None) == 0
            assert lib.ta_dim_bwd(x.data_ptr(), y.data_ptr(), None, planes, size, resize, rnd, top, left, None) == 0
        assert lib.ta_dim_phase_clock_read(cycles, groups) == 0
        for k, (tag, names) in enumerate((("dim_fwd_lanes_kernel", FWD), ("dim_bwd_lanes_kernel", BWD))):
            wgs = max(int(groups[k]), 1)
            per = [cycles[8 * k + p] / wgs for p in range(len(names))]
            print("N = %d  %s: %d workgroups (x planes per workgroup in the backward), %.0f shader-clock cycles per workgroup"
                  % (n, tag, wgs, sum(per)))
            for name, c in zip(names, per):
                print("    %-48s %9.0f cycles  %5.1f %%" % (name, c, 100 * c / max(sum(per), 1)))
    print("done")


if __name__ == "__main__":
    sys.exit(main())
