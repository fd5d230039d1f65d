"""AI-FGTM (Zou et al., AAAI 2022) -- Adam-style first / second moments of the raw gradient, a per-iteration step size
that sums to epsilon, and tanh instead of sign.  Mirror of transferattack/gradient/aifgtm.py:34-95.

    m_t = m_{t-1} + mu_1 g          v_t = v_{t-1} + mu_2 g g          direction = lam m_t / (sqrt(v_t) + 1e-20)
    alpha_t = epsilon * w_t / sum_k w_k,    w_k = (1 - beta_1^(k+1)) / sqrt(1 - beta_2^(k+1))
    delta <- box( clamp(delta + alpha_t tanh(direction), +-epsilon) )

The moments and the tanh step are the method's own arithmetic (elementwise torch ops on the device, same order of
operations as the reference); the weights w_k are host doubles, evaluated exactly as the reference evaluates them (the
sum is accumulated k = 0 .. T-1 on every call there; here once per forward, which gives the same double).  Like the
reference's, ``update_delta`` returns the moved tensor itself: the next iteration differentiates with respect to it."""
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

    def _weight(self, k):
        return (1 - self.beta_1 ** (k + 1)) / math.sqrt(1 - self.beta_2 ** (k + 1))

    def get_alpha(self, T, t_):
        """step size of iteration ``t_`` out of ``T`` (aifgtm.py:47-51)"""
        total = 0
        for k in range(T):
            total += self._weight(k)
        return self.epsilon / total * (1 - self.beta_1 ** (t_ + 1)) / math.sqrt(1 - self.beta_2 ** (t_ + 1))

    def update_delta(self, delta, data, grad, alpha, **kwargs):
        if self.norm == 'linfty':
            moved = torch.clamp(delta + alpha * grad.tanh(), -self.epsilon, self.epsilon)
        else:
            per_image = torch.norm(grad.view(grad.size(0), -1), dim=1).view(-1, 1, 1, 1)
            stepped = (delta + grad / (per_image + 1e-20) * alpha).view(delta.size(0), -1)
            moved = stepped.renorm(p=2, dim=0, maxnorm=self.epsilon).view_as(delta)
        return clamp(moved, img_min - data, img_max - data)

    def forward(self, data, label, **kwargs):
        data, label = self._to_device(data, label)
        delta = self.init_delta(data)
        steps = [self.get_alpha(self.epoch, it) for it in range(self.epoch)]
        first_moment, second_moment = 0, 0
        for alpha in steps:
            grad = self.get_grad(self.get_loss(self.get_logits(data + delta), label), delta)
            first_moment = first_moment + self.mu_1 * grad
            second_moment = second_moment + self.mu_2 * grad * grad
            direction = self.lam * first_moment / (torch.sqrt(second_moment) + 1e-20)
            delta = self.update_delta(delta, data, direction, alpha)
        return delta.detach()
