"""GRA (Zhu et al., ICCV 2023) -- gradient relevance: the current gradient and the mean gradient of ``num_neighbor``
samples are blended by their per-image cosine similarity, and a per-pixel decay indicator M shrinks the step where
the momentum sign flips, so ``update_delta`` receives a TENSOR step M*alpha.
Mirror of transferattack/gradient/gra.py:31-151.  HIP: neighbour sampling, gradient accumulation, momentum,
``ta_update_delta_linf`` with the per-element step operand."""
import torch

from ..attack import Attack
from .. import _hip
from ..transforms import Neighbor


class GRA(Attack):
    """Official arguments: epsilon=16/255, alpha=1.6/255, beta=3.5, num_neighbor=20, epoch=10, decay=1."""

    def __init__(self, model_name, epsilon=16/255, alpha=1.6/255, beta=3.5, num_neighbor=20, epoch=10, decay=1.,
                 targeted=False, random_start=False, norm='linfty', loss='crossentropy', device=None, attack='GRA',
                 **kwargs):
        super().__init__(attack, model_name, epsilon, targeted, random_start, norm, loss, device)
        self.alpha, self.epoch, self.decay, self.num_neighbor = alpha, epoch, decay, num_neighbor
        self.radius = beta * epsilon

    def get_average_gradient(self, data, delta, label, momentum, **kwargs):
        acc = None
        for i in range(self.num_neighbor):
            x_near = Neighbor.apply(delta, data, self.radius, self.rng_seed, self._next_offset(),
                                    self._uniform_like(data, self.radius))
            loss = self.get_loss(self.get_logits(self.transform(x_near, momentum=momentum)), label)
            grad = self.get_grad(loss, delta).contiguous()
            if acc is None:
                acc = torch.empty_like(grad)
            _hip.grad_accumulate(acc, grad, first=(i == 0))
        return acc / self.num_neighbor

    def get_cosine_similarity(self, cur_grad, sam_grad, **kwargs):
        cur = cur_grad.view(cur_grad.size(0), -1)
        sam = sam_grad.view(sam_grad.size(0), -1)
        cos = torch.sum(cur * sam, dim=1) / (torch.sqrt(torch.sum(cur ** 2, dim=1)) * torch.sqrt(torch.sum(sam ** 2, dim=1)))
        return cos.view(-1, 1, 1, 1)

    def get_decay_indicator(self, M, delta, cur_noise, last_noise, eta, **kwargs):
        if not isinstance(last_noise, torch.Tensor):
            last_noise = torch.full(cur_noise.shape, last_noise, device=cur_noise.device)
        same = (last_noise.sign() == cur_noise.sign()).float()
        return M * (same + (torch.ones_like(delta) - same) * eta)

    def forward(self, data, label, **kwargs):
        data, label = self._to_device(data, label)
        delta = self.init_delta(data)
        eta = 0.94
        M = torch.full_like(delta, 1 / eta)
        momentum = 0
        for _ in range(self.epoch):
            grad = self.get_grad(self.get_loss(self.get_logits(self.transform(data + delta, momentum=momentum)), label),
                                 delta)
            samgrad = self.get_average_gradient(data, delta, label, momentum)
            s = self.get_cosine_similarity(grad, samgrad)
            current_grad = s * grad + (1 - s) * samgrad
            last_momentum = momentum
            momentum = self.get_momentum(current_grad, momentum)
            M = self.get_decay_indicator(M, delta, momentum, last_momentum, eta)
            delta = self.update_delta(delta, data, momentum, M * self.alpha)
        return delta.detach()
