"""MIG (Ma et al., ICCV 2023) -- momentum integrated gradients: the gradient of the mean true-class probability over
``s_factor`` points on the straight path from a black baseline to x, times (x - baseline) / s_factor.
Mirror of transferattack/gradient/mig.py:35-90 (the step size is epsilon/epoch whatever ``alpha`` says, mig.py:37).
HIP: fused momentum + projected step on the integrated gradient."""
import torch
import torch.nn.functional as F

from .mifgsm import MIFGSM


class MIG(MIFGSM):
    """Official arguments: epsilon=16/255, alpha=epsilon/epoch, epoch=10, decay=1., s_factor=20."""

    def __init__(self, model_name, epsilon=16/255, alpha=1.6/255, epoch=10, decay=1., s_factor=20, targeted=False,
                 random_start=False, norm='linfty', loss='crossentropy', device=None, attack='MIG', **kwargs):
        super().__init__(model_name, epsilon, epsilon / epoch, epoch, decay, targeted, random_start, norm, loss, device,
                         attack, **kwargs)
        self.s_factor = s_factor

    def transform(self, data, **kwargs):
        x_base = torch.zeros_like(data).to(self.device)
        return torch.cat([x_base + i / self.s_factor * (data - x_base) for i in range(1, self.s_factor + 1)], dim=0)

    def get_loss(self, logits, label):
        loss = torch.mean(logits.gather(1, label.view(-1, 1)))
        return loss if self.targeted else -loss

    def forward(self, data, label, **kwargs):
        data, label = self._to_device(data, label)
        delta = self.init_delta(data)
        momentum = 0
        x_base = torch.zeros_like(data).to(self.device)
        fused = self._can_fuse_update()
        for _ in range(self.epoch):
            probs = F.softmax(self.get_logits(self.transform(data + delta)), dim=1)
            grad = self.get_grad(self.get_loss(probs, label.repeat(self.s_factor)), delta)
            i_grad = (data + delta - x_base) * grad / self.s_factor
            if fused:
                momentum = self._fused_update(i_grad.detach(), momentum, delta, data)
            else:
                momentum = self.get_momentum(i_grad, momentum)
                delta = self.update_delta(delta, data, momentum, self.alpha)
        return delta.detach()
