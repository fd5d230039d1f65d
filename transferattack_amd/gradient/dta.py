"""DTA (Yang et al., 2023) -- direction tuning: every outer iteration runs ``K`` small inner MI-FGSM steps from the
current point (with a running look-ahead ``gt`` normalised by the BATCH-wide L1 norm) and feeds the mean inner
gradient, plus decay * the outer gradient, to the outer momentum.  Mirror of transferattack/gradient/dta.py:33-91."""
import torch

from ..attack import Attack


class DTA(Attack):
    """Official arguments: epsilon=16/255, alpha=1.6/255, beta=1.5, K=10, u=0.8, epoch=10, decay=1."""

    def __init__(self, model_name, epsilon=16/255, alpha=1.6/255, beta=1.5, K=10, u=0.8, epoch=10, decay=1.,
                 targeted=False, random_start=False, norm='linfty', loss='crossentropy', device=None, attack='DTA',
                 **kwargs):
        super().__init__(attack, model_name, epsilon, targeted, random_start, norm, loss, device)
        self.alpha, self.epoch, self.decay, self.K, self.u = alpha, epoch, decay, K, u
        self.radius = beta * epsilon

    def forward(self, data, label, **kwargs):
        data, label = self._to_device(data, label)
        delta = self.init_delta(data)
        momentum = 0
        for _ in range(self.epoch):
            logits = self.get_logits(self.transform(data + delta, momentum=momentum))
            t_grad = self.get_grad(self.get_loss(logits, label), delta)
            gt = t_grad.clone().detach()
            delta_tk = delta.clone().detach().requires_grad_(True)
            gtk, momentum_tk = 0., 0.
            for _k in range(self.K):
                logits = self.get_logits(self.transform(data + delta_tk + gt, momentum=momentum))
                grad = self.get_grad(self.get_loss(logits, label), delta_tk)
                gt = self.u * gt + grad / torch.norm(grad, p=1)
                gtk = gtk + grad
                momentum_tk = self.get_momentum(grad, momentum_tk)
                delta_tk = self.update_delta(delta_tk, data, momentum_tk, self.alpha)
            grad = self.decay * t_grad + gtk / self.K
            momentum = self.get_momentum(grad, momentum)
            delta = self.update_delta(delta, data, momentum, self.alpha)
        return delta.detach()
