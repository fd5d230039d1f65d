"""SMI-FGRM (Han et al., 2023) -- sampling-based momentum with gradient RESCALING instead of the bare sign: the
gradient is averaged along a random walk of ``num_neighbor`` samples, accumulated into the momentum, and the
momentum is then replaced by rescale_factor * sign(m) * sigmoid(z-score of log2|m|) before the step.
Mirror of transferattack/gradient/smifgrm.py:33-102.  HIP: gradient accumulation, momentum, update_delta."""
import torch

from ..attack import Attack
from .. import _hip


class SMIFGRM(Attack):
    """Official arguments: epsilon=16/255, alpha=1.6/255, beta=1.5, num_neighbor=12, rescale_factor=2, epoch=10, decay=1."""

    def __init__(self, model_name, epsilon=16/255, alpha=1.6/255, beta=1.5, num_neighbor=12, rescale_factor=2, epoch=10,
                 decay=1., targeted=False, random_start=False, norm='linfty', loss='crossentropy', device=None,
                 attack='SMI-FGRM', **kwargs):
        super().__init__(attack, model_name, epsilon, targeted, random_start, norm, loss, device)
        self.alpha, self.epoch, self.decay = alpha, epoch, decay
        self.radius = beta * epsilon
        self.num_neighbor = num_neighbor
        self.rescale_factor = rescale_factor

    def get_sampled_grad(self, data, delta, label, momentum, **kwargs):
        acc = None
        samples = data + delta
        for i in range(self.num_neighbor):
            loss = self.get_loss(self.get_logits(self.transform(samples)), label)
            grad = self.get_grad(loss, delta).contiguous()
            if acc is None:
                acc = torch.empty_like(grad)
            _hip.grad_accumulate(acc, grad, first=(i == 0))
            step = self._uniform_like(data, self.radius)
            if step is None:
                step = torch.zeros_like(data).uniform_(-self.radius, self.radius)
            samples = samples + step                                  # random walk: the noise accumulates
        return acc / self.num_neighbor

    def rescale_grad(self, grad, **kwargs):
        log_abs = grad.abs().log2()
        mean = torch.mean(log_abs, dim=(1, 2, 3), keepdim=True)
        std = torch.std(log_abs, dim=(1, 2, 3), keepdim=True)
        return self.rescale_factor * grad.sign() * torch.sigmoid((log_abs - mean) / std)

    def forward(self, data, label, **kwargs):
        data = data.clone().detach().to(self.device)              # the reference skips the targeted unpacking here
        label = label.clone().detach().to(self.device)
        delta = self.init_delta(data)
        momentum = 0
        for _ in range(self.epoch):
            grad = self.get_sampled_grad(data, delta, label, momentum)
            momentum = self.rescale_grad(self.get_momentum(grad, momentum))
            delta = self.update_delta(delta, data, momentum, self.alpha)
        return delta.detach()
