"""Fuzz round 4's kernels on the host stand-in (tests/hipcpu) over random shapes: the byte source of the fused update against
its fp32-source twin, the resize + Normalize kernels against the torch ops the reference runs, the glue kernels' pass bits
against !(y <= 0) and threshold_backward.
    python tests/tools/fuzz_round4_host.py <seed> <cases>"""
import os
import sys

import numpy as np
import torch

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
from transferattack_amd import _hip    # noqa: E402
import test_hip_kernels as G           # noqa: E402
G.DEV = "cpu"
EPS, ALPHA = 16 / 255, 1.6 / 255
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rng = np.random.RandomState(seed)
gen = torch.Generator().manual_seed(seed)
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    # ---- byte source: any shape whose image size is a multiple of 4 takes the bytes, anything else the floats; same bits
    n, c = int(rng.randint(1, 5)), int(rng.randint(1, 4))
    h, w = int(rng.randint(1, 40)), int(rng.randint(1, 40)) * (4 if rng.rand() < 0.7 else 1)
    shape = (n, c, h, w)
    xb = torch.randint(0, 256, shape, generator=gen, dtype=torch.uint8)
    x = xb.float() / 255
    if rng.rand() < 0.3:
        x.view(-1)[int(rng.randint(0, x.numel()))] = float(rng.rand())          # not byte-valued: the flag must go up
    src = _hip.u8_source_probe(x.contiguous()) if (c * h * w) % 4 == 0 else None
    exact = bool(torch.equal(torch.round(x * 255).clamp(0, 255).to(torch.uint8).float() / 255, x))
    if src is not None:
        assert (int(src[1].item()) == 0) == exact, ("flag", shape)
    g = torch.randn(shape, generator=gen) * 1e-4
    m = torch.randn(shape, generator=gen)
    d = ((torch.randint(-10, 11, shape, generator=gen).float() * ALPHA).clamp(-EPS, EPS))
    d = torch.min(torch.max(d, 0 - x), 1 - x)
    outs = []
    for source in (None, src):
        dd, mm, xa = d.clone(), m.clone(), torch.empty(shape)
        _hip.mi_update(g.clone(), mm, mm, dd, x, 1.0, ALPHA, EPS, x_adv=xa, data_u8=source)
        outs.append((dd, mm, xa))
    for a, b in zip(*outs):
        assert torch.equal(a, b), ("byte source", shape)
    # ---- resize + Normalize
    a_in = int(rng.randint(4, 60))
    a_out = int(rng.randint(a_in + 1, int(1.5 * a_in) + 1)) if a_in >= 4 else a_in + 1
    a_out = min(a_out, (3 * a_in) // 2)
    if a_out > a_in:
        G.test_resize_normalize_kernels(int(rng.randint(1, 4)), a_in, a_out)
    # ---- pass bits
    numel8 = int(rng.randint(1, 300)) * 8
    ch = [c_ for c_ in (1, 2, 4, 8) if numel8 % c_ == 0][int(rng.randint(0, 3))]
    y = torch.randn(1, ch, numel8 // ch, 1, generator=gen)
    y[0, 0, 0, 0] = float("nan") if rng.rand() < 0.3 else 0.0
    bias = torch.randn(ch, generator=gen)
    bits = _hip.pass_bits_like(y)
    res = _hip.bias_act_(y.clone(), bias, mask=bits)
    want = np.packbits((~(res.reshape(-1) <= 0)).numpy().astype(np.uint8), bitorder="little")
    assert np.array_equal(bits.numpy(), want), ("pass bits", y.shape)
    ga = torch.randn_like(y)
    ref = torch.ops.aten.threshold_backward(ga, res, 0)
    got = _hip.relu_mask(ga, res, torch.empty_like(ga), mask=bits)
    assert np.array_equal(got.numpy(), ref.numpy(), equal_nan=True), ("relu_mask with bits", y.shape)
print('done ok')
