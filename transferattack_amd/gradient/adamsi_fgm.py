"""AdaMSI-FGM (Long et al., 2024) -- an Adam-like update without the sign: the gradient is divided by the root of a
running mean of its square, the momentum adds a look-back term along the last step (weighted by the ratio of successive
l1 gradient masses), and delta moves by alpha times that direction itself.
Mirror of transferattack/gradient/adamsi_fgm.py:31-82.

State kept between iterations (the reference hangs it on the attack object; so does this class, under the same names,
because ``get_momentum`` is a hook other code may call):
    t        iteration counter, from 1
    v        running mean of grad^2 with weight 1/t for the newest term   -> v_hat = sqrt(v) + 1e-16 / sqrt(t)
    s_prev   lambda * t^2 * ||grad||_1 of the previous iteration          -> beta1 = s_prev / (s_t + 1)
    x0, x_prev, delta   the look-back term is x0 + delta - x_prev; the reference never advances x_prev past x0, so the
             term is the current delta -- kept as written, bit for bit
The moments are the method's own elementwise arithmetic: torch ops on the device in the reference's operation order (the
division by v_hat, the product with beta1 and the final add each round once, as there).  The step has no sign, so it is
not the fused HIP update; projection onto the eps-ball and the image box follows."""
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

    def _reset(self, data):
        self.t = 0
        self.x0 = data.clone().detach()
        self.x_prev = self.x0.clone()
        self.v = torch.zeros_like(self.x0)
        self.s_prev = torch.zeros(self.x0.size(0), device=self.x0.device)

    def get_momentum(self, grad, momentum, **kwargs):
        n = grad.size(0)
        s_t = self.lambda_ * (self.t ** 2) * grad.abs().view(n, -1).sum(dim=1)          # l1 mass, scaled
        beta1 = (self.s_prev / (s_t + 1.0)).view(n, 1, 1, 1)
        self.s_prev = s_t
        newest = 1.0 / self.t                                                          # weight of grad^2 in the mean
        keep = 1.0 - newest
        self.v = keep * self.v + (1.0 - keep) * (grad * grad)
        v_hat = self.v.sqrt() + 1e-16 / math.sqrt(self.t)
        look_back = beta1 * (self.x0 + self.delta - self.x_prev)
        return grad / v_hat + (momentum * self.decay + look_back)

    def update_delta(self, delta, data, grad, alpha, **kwargs):
        if self.norm == 'linfty':
            moved = torch.clamp(delta + alpha * grad, -self.epsilon, self.epsilon)
        else:
            per_image = torch.norm(grad.view(grad.size(0), -1), dim=1).view(-1, 1, 1, 1)
            stepped = (delta + grad / (per_image + 1e-20) * alpha).view(delta.size(0), -1)
            moved = stepped.renorm(p=2, dim=0, maxnorm=self.epsilon).view_as(delta)
        return clamp(moved, img_min - data, img_max - data).detach().requires_grad_(True)

    def forward(self, data, label, **kwargs):
        # no targeted-label unpacking here in the reference either (adamsi_fgm.py:66-67)
        data, label = data.clone().detach().to(self.device), label.clone().detach().to(self.device)
        self._reset(data)
        self.delta = delta = self.init_delta(data)
        momentum = 0
        while self.t < self.epoch:
            self.t += 1
            loss = self.get_loss(self.get_logits(self.transform(data + delta, momentum=momentum)), label)
            momentum = self.get_momentum(self.get_grad(loss, delta), momentum)
            self.delta = delta = self.update_delta(delta, data, momentum, self.alpha)
        return delta.detach()
