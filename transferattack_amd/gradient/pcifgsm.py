"""PC-I-FGSM (Wan et al., 2021) -- prediction-correction: a full-epsilon FGSM "prediction" step from the current
point, whose gradient corrects the (batch-L1-normalised) current gradient before the momentum update.
Mirror of transferattack/gradient/pcifgsm.py:31-83 (K = 1 prediction step, as the reference hard-codes)."""
import torch

from ..attack import Attack


class PCIFGSM(Attack):
    """Official arguments: epsilon=16/255, alpha=1.6/255, epoch=10, decay=1."""

    def __init__(self, model_name, epsilon=16/255, alpha=1.6/255, epoch=10, decay=1., targeted=False, random_start=False,
                 norm='linfty', loss='crossentropy', device=None, attack='PC-FGSM', **kwargs):
        super().__init__(attack, model_name, epsilon, targeted, random_start, norm, loss, device)
        self.alpha, self.epoch, self.decay, self.K = alpha, epoch, decay, 1

    def forward(self, data, label, **kwargs):
        data, label = self._to_device(data, label)
        delta = self.init_delta(data)
        momentum = 0
        for _ in range(self.epoch):
            delta_pre = self.init_delta(data)
            logits = self.get_logits(self.transform(data + delta, momentum=momentum))
            grad = self.get_grad(self.get_loss(logits, label), delta)
            g_pre = self.decay * torch.zeros_like(delta) + grad / torch.norm(grad, p=1)
            for _k in range(self.K):
                logits = self.get_logits(self.transform(data + delta + delta_pre, momentum=momentum))
                grad = self.get_grad(self.get_loss(logits, label), delta_pre)
                g_pre = self.decay * g_pre + grad / (self.K * torch.norm(grad, p=1))
                delta_pre = self.update_delta(delta_pre, data, grad, self.epsilon)
            momentum = self.get_momentum(g_pre, momentum)
            delta = self.update_delta(delta, data, momentum, self.alpha)
        return delta.detach()
