"""EMI-FGSM (Wang et al., BMVC 2021) -- enhanced momentum: the gradient is averaged over ``num_sample`` points
x + c_i * alpha * g_bar sampled along the previous normalised gradient g_bar (c_i linear in [-radius, radius]).
Mirror of transferattack/gradient/emifgsm.py:33-105.  The sample stack is built by ``ta_axpy`` launches writing
straight into the slices of one [S*N, C, H, W] buffer; its backward (sum of the S slices, in autograd's order) is
the unit-scale case of the SIM backward kernel."""
import numpy as np
import torch

from .mifgsm import MIFGSM
from .. import _hip


class _SampleStack(torch.autograd.Function):
    """y[i*N + b] = x[b] + coeff_i * g_bar[b];  dx = sum_i dy_i accumulated i = S-1 .. 0 (autograd's order)."""

    @staticmethod
    def forward(ctx, x, g_bar, coeffs):
        x = x.contiguous()
        n = x.shape[0]
        y = torch.empty((len(coeffs) * n,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        for i, c in enumerate(coeffs):
            _hip.axpy(x, g_bar, c, y[i * n:(i + 1) * n])
        ctx.samples = len(coeffs)
        ctx.in_shape = x.shape
        return y

    @staticmethod
    def backward(ctx, gy):
        gy = gy.contiguous()
        gx = torch.empty(ctx.in_shape, dtype=gy.dtype, device=gy.device)
        _hip.sum_copies_bwd(gy, gx, ctx.samples)
        return gx, None, None


class EMIFGSM(MIFGSM):
    """Official arguments: epsilon=16/255, alpha=1.6/255, epoch=10, decay=1., num_sample=11, radius=7, sample_method='linear'."""

    def __init__(self, model_name, epsilon=16/255, alpha=1.6/255, epoch=10, decay=1., num_sample=11, radius=7,
                 sample_method='linear', targeted=False, random_start=False, norm='linfty', loss='crossentropy',
                 device=None, attack='EMI-FGSM', **kwargs):
        super().__init__(model_name, epsilon, alpha, epoch, decay, targeted, random_start, norm, loss, device, attack)
        self.num_sample = num_sample
        self.radius = radius
        self.sample_method = sample_method.lower()

    def get_factors(self):
        if self.sample_method == 'linear':
            return np.linspace(-self.radius, self.radius, num=self.num_sample)
        if self.sample_method == 'uniform':
            return np.random.uniform(-self.radius, self.radius, size=self.num_sample)
        if self.sample_method == 'gaussian':
            return np.clip(np.random.normal(size=self.num_sample) / 3, -1, 1) * self.radius
        raise Exception('Unsupported sampling method {}!'.format(self.sample_method))

    def transform(self, x, grad, **kwargs):
        factors = np.linspace(-self.radius, self.radius, num=self.num_sample)       # emifgsm.py:57: always linear
        if not isinstance(grad, torch.Tensor):                                       # first iteration: g_bar = 0
            grad = torch.zeros_like(x)
        return _SampleStack.apply(x, grad.contiguous(), [float(f * self.alpha) for f in factors])

    def get_loss(self, logits, label):
        label = label.repeat(self.num_sample)
        return -self.loss(logits, label) if self.targeted else self.loss(logits, label)

    def forward(self, data, label, **kwargs):
        data, label = self._to_device(data, label)
        delta = self.init_delta(data)
        momentum, bar_grad = 0, 0
        fused = self._can_fuse_update()
        for _ in range(self.epoch):
            logits = self.get_logits(self.transform(data + delta, grad=bar_grad))
            grad = self.get_grad(self.get_loss(logits, label), delta).contiguous()
            bar_grad = self.l1_normalize(grad)
            if fused:
                momentum = self._fused_update(grad, momentum, delta, data)
            else:
                momentum = self.get_momentum(grad, momentum)
                delta = self.update_delta(delta, data, momentum, self.alpha)
        return delta.detach()
