"""FGSRA (CIKM 2024) -- frequency-guided sample relevance: each iteration draws ``max_iter`` neighbours of x+delta
in the DCT domain (uniform noise of radius beta*eps, then a random per-coefficient spectrum mask in [1-rho, 1+rho]),
weights their gradients by the per-image cosine between the neighbour and x+delta, blends the result with the current
gradient GRA-style, and shrinks the per-pixel step by 0.94 wherever the momentum's sign disagrees with the blended
gradient's -- so ``update_delta`` receives a TENSOR step.
Mirror of transferattack/gradient/fgsra.py:36-46 (constructor), :49-123 (DCT-II / inverse by Makhoul's FFT
factorisation, unnormalised), :163-215 (loop).  HIP: momentum, ``ta_update_delta_linf`` with the per-element step;
the FFTs are rocFFT through torch.fft, differentiable, as in the reference."""
import torch

from .. import spectrum
from ..attack import Attack
from ..spectrum import MakhoulDct


class FGSRA(Attack):
    """Official arguments: epsilon=16/255, alpha=1.6/255, rho=0.7, beta=2.0, max_iter=20, epoch=10, decay=1."""

    def __init__(self, model_name, epsilon=16/255, alpha=1.6/255, rho=0.7, beta=2.0, max_iter=20, epoch=10, decay=1.,
                 targeted=False, random_start=False, norm='linfty', loss='crossentropy', device=None, attack='FGSRA',
                 **kwargs):
        super().__init__(attack, model_name, epsilon, targeted, random_start, norm, loss, device)
        self._schedule(alpha, epoch, decay)
        self.rho, self.beta, self.max_iter = rho, beta, max_iter
        self.targeted = False                       # fgsra.py:46: the reference runs this attack untargeted only
        self._dct = MakhoulDct()

    # DCT-II / inverse over the last axis and the last two axes (fgsra.py:49-123): shared with SSM
    def dct(self, x, norm=None):
        return self._dct.dct(x, norm)

    def idct(self, X, norm=None):
        return self._dct.idct(X, norm)

    def dct_2d(self, x, norm=None):
        return self._dct.dct_2d(x, norm)

    def idct_2d(self, x, norm=None):
        return self._dct.idct_2d(x, norm)

    # ----------------------------------------------------------------------------------------------------- loop
    def _unit_uniform(self, x):
        """U[0, 1) shaped like x: device generator, or the test's injected CPU draws"""
        if self.noise_source is not None:
            return self.noise_source(x.shape, 0.0, 1.0).to(self.device)
        return torch.rand_like(x)

    @staticmethod
    def _cosine(a, b):
        """per-image cosine over C,H,W, kept as [N,1,1,1] (fgsra.py:198, 205-207)"""
        dims = [1, 2, 3]
        return (a * b).sum(dims, keepdim=True) / (torch.sqrt((a ** 2).sum(dims, keepdim=True)) *
                                                  torch.sqrt((b ** 2).sum(dims, keepdim=True)))

    def spectrum_neighbor(self, x):
        radius = self.epsilon * self.beta
        jitter = self._unit_uniform(x) * 2 * radius - radius
        mask = self._unit_uniform(x) * 2 * self.rho + 1 - self.rho
        return spectrum.spectrum_view(x, jitter, mask)         # idct_2d(dct_2d(x + jitter) * mask)

    def forward(self, data, label, **kwargs):
        data, label = self._to_device(data, label)
        delta = self.init_delta(data)
        step_scale = torch.ones_like(data) * 10 / 9.4
        momentum = 0
        for _ in range(self.epoch):
            x = data + delta
            current_grad = self.get_grad(self.get_loss(self.get_logits(x, momentum=momentum), label), delta)
            sample_grads, relevance = [], []
            for _s in range(self.max_iter):
                x_near = self.spectrum_neighbor(x)
                sample_grads.append(self.get_grad(self.get_loss(self.get_logits(x_near), label), delta))
                relevance.append(self._cosine(x, x_near))
            avg_grad = (torch.stack(sample_grads, dim=1) * torch.stack(relevance, dim=1)).sum(1)
            s = self._cosine(current_grad, avg_grad)
            current_grad = s * current_grad + (1 - s) * avg_grad
            momentum = self.get_momentum(current_grad, momentum)
            agree = (torch.sign(momentum) == torch.sign(current_grad)).float()
            step_scale = step_scale * (agree + (torch.ones_like(data) - agree) * 0.94)
            delta = self.update_delta(delta, data, momentum, self.alpha * step_scale)
        return delta.detach()
