"""I-FGS2M (Gao et al., 2021) -- staircase sign: instead of sign(g) every element steps by sign(g) times the weight of
the percentile band its |g| falls in (64 bands of 1.5625 % per (image, channel) plane; band j weighs (2j+1) * k/100).
Mirror of transferattack/gradient/ifgssm.py:32-63.

The reference finds the bands with 64 ``torch.quantile`` calls (one sort of every plane each) and 64 masked
accumulation passes over the gradient.  Here: ONE quantile call for all band edges (same fp32 rank arithmetic,
interpolation 'lower'), one ``searchsorted`` for the band of every element, and the projected step itself is the
tensor-step form of ``ta_update_delta_linf`` -- step[i] = alpha * weight[band[i]], direction sign(g[i]) -- which is
bit-identical to ``alpha * (sign * weight)`` since the sign only flips the product."""
import numpy as np
import torch

from ..attack import Attack
from .. import _hip


class IFGSSM(Attack):
    """Official arguments: epsilon=16/255, alpha=epsilon/epoch=1.6/255, epoch=10, k=1.5625."""

    def __init__(self, model_name, epsilon=16/255, alpha=1.6/255, epoch=10, targeted=False, random_start=False,
                 norm='linfty', loss='crossentropy', device=None, k=1.5625, **kwargs):
        super().__init__('I-FGSSM', model_name, epsilon, targeted, random_start, norm, loss, device)
        self._schedule(alpha, epoch, 0)
        self.k = k

    def band_weights(self, noise):
        """weight of the band each element of ``noise`` [N, C, H, W] falls in (0 above the last band edge): the
        magnitude of the reference's ``ssign`` (ifgssm.py:38-53)"""
        n, c, h, w = noise.shape
        levels = np.arange(self.k, 100.1, self.k)
        base = self.k / 100
        q = torch.tensor([float(i / 100) for i in levels], dtype=noise.dtype, device=noise.device)
        planes = noise.abs().reshape(n * c, h * w)
        edges = torch.quantile(planes, q, dim=1, interpolation='lower').t().contiguous()        # [planes, bands]
        band = torch.searchsorted(edges, planes, right=False)           # first band whose edge is >= |g|
        weights = torch.tensor([base + 2 * base * j for j in range(len(levels))] + [0.0], dtype=torch.float64)
        return weights.to(noise.dtype).to(noise.device)[band].reshape(n, c, h, w)

    def ssign(self, noise):
        return torch.sign(noise) * self.band_weights(noise)

    def update_delta(self, delta, data, grad, alpha, **kwargs):
        src = delta.detach().contiguous()
        out = torch.empty_like(src)
        grad = grad.detach().contiguous()
        if self.norm == 'linfty':
            _hip.update_delta_linf(src, data.contiguous(), grad, alpha * self.band_weights(grad), self.epsilon, out)
        else:
            _hip.update_delta_l2(src, data.contiguous(), grad, alpha, self.epsilon, out)
        return out.requires_grad_(True)
