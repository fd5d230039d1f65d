"""PI-FGSM (Gao et al., ECCV 2020) -- patch-wise step: the part of the amplified step that overshoots the eps-ball
("cut noise") is redistributed to the neighbouring pixels by a 3x3 uniform projection kernel (centre 0).
Mirror of transferattack/gradient/pifgsm.py:33-112 (decay=0: PI-FGSM, decay=1: MPI-FGSM).  The projection conv is
the same depthwise 'same' kernel as TIM (``ta_depthwise_conv2d_same``, k=3 instantiation)."""
import numpy as np
import torch

from ..attack import Attack
from .. import _hip
from ..utils import clamp, img_max, img_min


class PIFGSM(Attack):
    """Official arguments: epsilon=16/255, alpha=1.6/255, epoch=10, decay=0., kern_size=3, gamma=16.0, beta=10.0."""

    def __init__(self, model_name, epsilon=16.0/255, alpha=1.6/255, epoch=10, decay=0., kern_size=3, gamma=16.0,
                 beta=10.0, targeted=False, random_start=False, norm='linfty', loss='crossentropy', device=None,
                 attack='PI-FGSM', **kwargs):
        super().__init__(attack, model_name, epsilon, targeted, random_start, norm, loss, device)
        self.alpha, self.epoch, self.decay = alpha, epoch, decay
        self.kern_size = kern_size
        self.gamma = gamma / 255.0
        self.beta = beta

    def project_kern(self, kern_size):
        """Uniform k x k kernel with a hole in the middle (the overshoot goes to the NEIGHBOURS), one per plane."""
        kern = np.full((kern_size, kern_size), 1.0, dtype=np.float32) / (kern_size ** 2 - 1)
        centre = kern_size // 2
        kern[centre, centre] = 0.0
        stack = np.repeat(kern[None, None].astype(np.float32), 3, axis=0)
        return torch.tensor(stack).to(self.device), centre

    def project_noise(self, x, stack_kern, padding_size):
        x = x.contiguous()
        spread = torch.empty_like(x)
        _hip.depthwise_conv2d_same(x, spread, stack_kern[0, 0].contiguous())   # padding k//2 == 'same' for odd k
        return spread

    def update_delta(self, delta, data, grad, alpha, projection, **kwargs):
        if self.norm == 'linfty':
            stepped = delta + alpha * grad.sign() + projection
            delta = torch.clamp(stepped, -self.epsilon, self.epsilon)
        else:
            flat_norm = torch.norm(grad.view(grad.size(0), -1), dim=1).view(-1, 1, 1, 1)
            stepped = delta + grad / (flat_norm + 1e-20) * alpha + projection
            delta = stepped.view(delta.size(0), -1).renorm(p=2, dim=0, maxnorm=self.epsilon).view_as(delta)
        return clamp(delta, img_min - data, img_max - data)

    def forward(self, data, label, **kwargs):
        data, label = self._to_device(data, label)
        delta = self.init_delta(data)
        stack_kern, padding_size = self.project_kern(self.kern_size)
        step = self.beta * self.alpha                                   # amplified step size
        momentum, amplification = 0.0, 0.0
        for _ in range(self.epoch):
            grad = self.get_grad(self.get_loss(self.get_logits(self.transform(data + delta)), label), delta)
            momentum = self.get_momentum(grad, momentum)
            amplification = amplification + step * momentum.sign()
            overshoot = torch.clamp(abs(amplification) - self.epsilon, 0, 10000.0) * torch.sign(amplification)
            projection = self.gamma * torch.sign(self.project_noise(overshoot, stack_kern, padding_size))
            amplification = amplification + projection
            delta = self.update_delta(delta.detach(), data, momentum, step, projection).requires_grad_(True)
        return delta.detach()
