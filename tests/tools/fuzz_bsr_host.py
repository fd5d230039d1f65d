"""Fuzz the BSR kernels (host stand-in, tests/hipcpu) against the oracle: random shapes, block counts, copy counts -- the
reference-order backward must reproduce ATen bit for bit wherever the forward does -- and random plane counts against
the one-plane-per-thread form (geometry shared between the planes of a thread).
    python tests/tools/fuzz_bsr_host.py <seed> <cases>"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, ROOT + '/oracle', ROOT + '/tests'):
    sys.path.insert(0, p)
import host_kernels  # noqa: E402


class Patch:
    def setattr(self, obj, name, value):
        setattr(obj, name, value)

    def setenv(self, name, value):
        os.environ[name] = value

    def delenv(self, name):
        os.environ.pop(name, None)


host_kernels.install(Patch(), tag=None, env={})
import test_zz_hip_widened as W  # noqa: E402
W.DEV = 'cpu'
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 20
for it in range(cases):
    n, c = int(rng.randint(1, 3)), int(rng.randint(1, 4))
    h, w = int(rng.randint(9, 90)), int(rng.randint(9, 200))
    nb = int(rng.randint(1, min(8, h // 2, w // 2) + 1))
    copies = int(rng.randint(1, 5))
    try:
        W.test_bsr_kernels_random((n, c, h, w), nb, copies, Patch())
    except AssertionError:
        print('MISMATCH', (n, c, h, w), nb, copies)
        raise
for it in range(max(1, cases // 4)):
    # plane groups: enough tiny planes that the launch heuristics pick 2 / 3 / 6 / 12 planes per thread
    planes_per_image = int(rng.choice([1, 2, 3]))
    images = int(rng.choice([768, 1025, 1538, 2050, 4096])) + int(rng.randint(0, 3))
    try:
        W.test_bsr_plane_groups((images, planes_per_image, int(rng.randint(3, 7)), int(rng.randint(5, 12))))
    except AssertionError:
        print('MISMATCH (plane groups)', images, planes_per_image)
        raise
print('done ok')
