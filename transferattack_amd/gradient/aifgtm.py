"""AI-FGTM (Zou et al., AAAI 2022) -- Adam-style first / second moments of the raw gradient, a per-iteration step size
that sums to epsilon, and tanh instead of sign.  Mirror of transferattack/gradient/aifgtm.py:34-95.  The moments and the
tanh step are the method's own arithmetic (elementwise torch ops on the device); the eps-ball / image-box projection
is the base class's."""
import math

import torch

from ..attack import Attack
from ..utils import clamp, img_max, img_min


class AIFGTM(Attack):
    """Official arguments: epsilon=16/255, alpha=1.6/255, epoch=10, decay=1., beta_1=0.9, beta_2=0.99, lam=1.3, mu_1=1.5,
    mu_2=1.9."""

    def __init__(self, model_name, epsilon=16/255, alpha=1.6/255, epoch=10, decay=1., targeted=False, random_start=False,
                 norm='linfty', loss='crossentropy', device=None, attack='AI-FGTM', beta_1=0.9, beta_2=0.99, lam=1.3,
                 mu_1=1.5, mu_2=1.9, **kwargs):
        super().__init__(attack, model_name, epsilon, targeted, random_start, norm, loss, device)
        self._schedule(alpha, epoch, decay)
        self.beta_1, self.beta_2, self.lam, self.mu_1, self.mu_2 = beta_1, beta_2, lam, mu_1, mu_2

    def get_alpha(self, T, t_):
        res = 0
        for t in range(T):
            res += (1 - self.beta_1 ** (t + 1)) / math.sqrt(1 - self.beta_2 ** (t + 1))
        return self.epsilon / res * (1 - self.beta_1 ** (t_ + 1)) / math.sqrt(1 - self.beta_2 ** (t_ + 1))

    def update_delta(self, delta, data, grad, alpha, **kwargs):
        if self.norm == 'linfty':
            delta = torch.clamp(delta + alpha * grad.tanh(), -self.epsilon, self.epsilon)
        else:
            grad_norm = torch.norm(grad.view(grad.size(0), -1), dim=1).view(-1, 1, 1, 1)
            scaled_grad = grad / (grad_norm + 1e-20)
            delta = (delta + scaled_grad * alpha).view(delta.size(0), -1).renorm(p=2, dim=0, maxnorm=self.epsilon).view_as(delta)
        return clamp(delta, img_min - data, img_max - data)

    def forward(self, data, label, **kwargs):
        data, label = self._to_device(data, label)
        delta = self.init_delta(data)
        momentum, v = 0, 0
        for it in range(self.epoch):
            grad = self.get_grad(self.get_loss(self.get_logits(data + delta), label), delta)
            momentum = momentum + self.mu_1 * grad
            v = v + self.mu_2 * grad * grad
            alpha = self.get_alpha(self.epoch, it)
            delta = self.update_delta(delta, data, self.lam * momentum / (torch.sqrt(v) + 1e-20), alpha)
        return delta.detach()
