"""Fuzz the update stack (host stand-in, tests/hipcpu) against the oracle on random ragged shapes: the assertions of
tests/test_hip_kernels.py::test_fused_update_random and ::test_normalize_and_producer_side_partials.
    python tests/tools/fuzz_update_host.py <seed> <cases>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, ROOT + '/oracle', ROOT + '/tests'):
    sys.path.insert(0, p)
import host_kernels           # noqa: E402


class P:
    def setattr(self, o, n, v):
        setattr(o, n, v)

    def setenv(self, n, v):
        os.environ[n] = v


host_kernels.install(P(), env={})
import test_hip_kernels as G  # noqa: E402
G.DEV = "cpu"
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    shape = (int(rng.randint(1, 6)), int(rng.randint(1, 4)), int(rng.randint(1, 70)), int(rng.randint(1, 90)))
    try:
        G.test_fused_update_random(shape)
        if shape[1] <= 3:
            G.test_normalize_and_producer_side_partials(shape)
    except AssertionError:
        print('MISMATCH', shape)
        raise
print('done ok')
