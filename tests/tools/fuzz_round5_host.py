"""Fuzz round 5's kernels on the host stand-in (tests/hipcpu): the folded-Normalize pair against the separate kernels on random
shapes (any channel count, plane sizes that are not multiples of 4, planes shorter than a tile, byte-valued and other images).
    python tests/tools/fuzz_round5_host.py <seed> <cases>"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, ROOT + '/oracle', ROOT + '/tests'):
    sys.path.insert(0, p)
import host_kernels


class P:
    def setattr(self, o, n, v): setattr(o, n, v)
    def setenv(self, n, v): os.environ[n] = v


host_kernels.install(P(), tag='fuzz5', env={})
from transferattack_amd import _hip
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
EPS, ALPHA = 16 / 255, 1.6 / 255
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 120):
    n, c = int(rng.randint(1, 5)), int(rng.randint(1, 5))
    h, w = int(rng.randint(1, 70)), int(rng.randint(1, 70))
    if rng.rand() < 0.15:
        h, w = 224, 224
    shape = (n, c, h, w)
    gen = torch.Generator().manual_seed(int(rng.randint(1 << 30)))
    byte_valued = rng.rand() < 0.7
    x = torch.randint(0, 256, shape, generator=gen).float() / 255 if byte_valued else torch.rand(shape, generator=gen)
    mean, std = torch.rand(c, generator=gen), torch.rand(c, generator=gen) * 0.5 + 0.1
    gy = torch.randn(shape, generator=gen) * 1e-3
    mom = torch.randn(shape, generator=gen)
    delta = (torch.rand(shape, generator=gen) - 0.5) * 2 * EPS
    delta = torch.min(torch.max(delta, 0 - x), 1 - x)
    src = _hip.u8_source_probe(x) if (c * h * w) % 4 == 0 else None
    # forward end
    xa, y_ref, y = x + delta, torch.empty(shape), torch.full(shape, float('nan'))
    _hip.normalize_fwd(xa, y_ref, mean, std)
    _hip.normalize_adv_fwd(x, delta, y, mean, std, data_u8=src)
    ok = torch.equal(y, y_ref)
    # backward end, three momentum shapes
    for m_in, keep in ((mom, True), (None, True), (None, False)):
        gx = torch.empty(shape)
        _hip.normalize_bwd(gy, gx, std)
        d_ref, m_ref = delta.clone(), (torch.empty(shape) if keep else None)
        _hip.mi_update(gx, None if m_in is None else m_in.clone(), m_ref, d_ref, x, 1.0 if keep else 0.0, ALPHA, EPS)
        for handed in (True, False):
            g_in = gy.clone()
            if handed:
                _hip.abs_sum_partials_std(g_in, std)
            d, m = delta.clone(), (torch.empty(shape) if keep else None)
            _hip.mi_update(g_in, None if m_in is None else m_in.clone(), m, d, x, 1.0 if keep else 0.0, ALPHA, EPS, data_u8=src, std=std)
            if (h * w) % 4 == 0 or (c * h * w) % 4 != 0:          # both sides take the same (vector or scalar) form: same bits
                ok = ok and torch.equal(d, d_ref) and (m is None or torch.equal(m, m_ref))
            else:        # plane size not a multiple of 4 in an image that is: the std form sums in the scalar form's fixed order
                ok = ok and (m is None or bool(((m - m_ref).abs() <= 8 * 2.0 ** -24 * (m_ref.abs() + (gy / std.view(1, -1, 1, 1)).abs()
                             / (gy / std.view(1, -1, 1, 1)).abs().mean(dim=(1, 2, 3), keepdim=True) + 1e-30)).all()))
                flips = (d != d_ref)
                ok = ok and (not bool(flips.any()) or bool((m_ref[flips].abs() < 1e-5).all()))
    if not ok:
        bad += 1
        print('MISMATCH', shape, byte_valued)
print('done, mismatches:', bad)
