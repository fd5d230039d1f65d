"""AdaMSI-FGM (Long et al., 2024) -- an Adam-like update without the sign: the gradient is divided by the root of a
running mean of its square, the momentum adds a look-back term along the last step (weighted by the ratio of successive
l1 gradient masses), and delta moves by alpha times that direction itself.
Mirror of transferattack/gradient/adamsi_fgm.py:31-82.  The moments are the method's own elementwise arithmetic
(torch ops on the device, in the reference's operation order); as in the reference the look-back term is
``x0 + delta - x_prev`` with ``x_prev`` never advanced, i.e. the current delta."""
import math

import torch

from .mifgsm import MIFGSM
from ..utils import clamp, img_max, img_min


class AdaMSI_FGM(MIFGSM):
    """Official arguments: epsilon=16/255, alpha=1.6/255, epoch=10, decay=1., lambda_=0.6."""

    def __init__(self, model_name, epsilon=16/255, alpha=1.6/255, epoch=10, decay=1.0, targeted=False, random_start=False,
                 norm='linfty', loss='crossentropy', device=None, attack='AdaMSI_FGM', lambda_=0.6, **kwargs):
        super().__init__(model_name, epsilon, alpha, epoch, decay, targeted, random_start, norm, loss, device, attack)
        self.lambda_ = lambda_

    def get_momentum(self, grad, momentum, **kwargs):
        mass = grad.abs().view(grad.size(0), -1).sum(dim=1)
        s_t = self.lambda_ * (self.t ** 2) * mass
        beta1_t = self.s_prev / (s_t + 1.0)
        beta2_t = 1.0 - 1.0 / self.t
        self.v = beta2_t * self.v + (1.0 - beta2_t) * (grad * grad)
        v_hat = self.v.sqrt() + 1e-16 / math.sqrt(self.t)
        momentum = momentum * self.decay + beta1_t.view(-1, 1, 1, 1) * (self.x0 + self.delta - self.x_prev)
        self.s_prev = s_t
        return grad / v_hat + momentum

    def update_delta(self, delta, data, grad, alpha, **kwargs):
        if self.norm == 'linfty':
            delta = torch.clamp(delta + alpha * grad, -self.epsilon, self.epsilon)
        else:
            grad_norm = torch.norm(grad.view(grad.size(0), -1), dim=1).view(-1, 1, 1, 1)
            stepped = (delta + grad / (grad_norm + 1e-20) * alpha).view(delta.size(0), -1)
            delta = stepped.renorm(p=2, dim=0, maxnorm=self.epsilon).view_as(delta)
        return clamp(delta, img_min - data, img_max - data).detach().requires_grad_(True)

    def forward(self, data, label, **kwargs):
        data = data.clone().detach().to(self.device)            # (no targeted-label unpacking in the reference either)
        label = label.clone().detach().to(self.device)
        self.x0 = data.clone().detach()
        self.x_prev = self.x0.clone()
        self.v = torch.zeros_like(self.x0)
        self.s_prev = torch.zeros(self.x0.size(0), device=self.x0.device)
        delta = self.delta = self.init_delta(data)
        momentum = 0
        for self.t in range(1, self.epoch + 1):
            logits = self.get_logits(self.transform(data + delta, momentum=momentum))
            grad = self.get_grad(self.get_loss(logits, label), delta)
            momentum = self.get_momentum(grad, momentum)
            delta = self.delta = self.update_delta(delta, data, momentum, self.alpha)
        return delta.detach()
