"""VMI-FGSM (Wang & He, CVPR 2021) -- variance tuning: the momentum uses grad + v where
v = mean_i grad(x + delta + U(-beta*eps, beta*eps)) - grad over ``num_neighbor`` samples.
Mirror of transferattack/gradient/vmifgsm.py:28-97.

HIP path: neighbour sampling is one kernel (``ta_vmi_neighbor``: x + delta + Philox noise, nothing but the
sample itself is written), accumulation/finalisation are ``ta_grad_accumulate`` / ``ta_variance_finalize``
and ``grad + variance`` is folded into the fused update (its ``v`` operand), so no grad-sized temporary is
ever materialised besides the accumulator.
"""
import torch

from ..attack import Attack
from .. import _hip
from ..transforms import Neighbor


class VMIFGSM(Attack):
    """Official arguments: epsilon=16/255, alpha=epsilon/epoch=1.6/255, beta=1.5, num_neighbor=20, epoch=10, decay=1."""

    def __init__(self, model_name, epsilon=16/255, alpha=1.6/255, beta=1.5, num_neighbor=20, epoch=10, decay=1.,
                 targeted=False, random_start=False, norm='linfty', loss='crossentropy', device=None,
                 attack='VMI-FGSM', **kwargs):
        super().__init__(attack, model_name, epsilon, targeted, random_start, norm, loss, device)
        self.alpha = alpha
        self.radius = beta * epsilon
        self.epoch = epoch
        self.decay = decay
        self.num_neighbor = num_neighbor

    def get_variance(self, data, delta, label, cur_grad, momentum, **kwargs):
        """mean of the neighbour gradients minus the current gradient (vmifgsm.py:42-58)."""
        acc = torch.empty_like(cur_grad)
        for i in range(self.num_neighbor):
            noise = None
            if self.noise_source is not None:
                noise = self.noise_source(data.shape, -self.radius, self.radius).to(self.device).contiguous()
            x_near = Neighbor.apply(delta, data, self.radius, self.rng_seed, self._next_offset(), noise)
            logits = self.get_logits(self.transform(x_near, momentum=momentum))
            loss = self.get_loss(logits, label)
            _hip.grad_accumulate(acc, self.get_grad(loss, delta).contiguous(), first=(i == 0))
        variance = torch.empty_like(cur_grad)
        _hip.variance_finalize(acc, cur_grad.contiguous(), variance, self.num_neighbor)
        return variance

    def forward(self, data, label, **kwargs):
        if self.targeted:
            assert len(label) == 2
            label = label[1]
        data = data.clone().detach().to(self.device)
        label = label.clone().detach().to(self.device)
        delta = self.init_delta(data)
        momentum, variance = 0, 0
        fused = self._can_fuse_update()
        for _ in range(self.epoch):
            logits = self.get_logits(self.transform(data + delta, momentum=momentum))
            loss = self.get_loss(logits, label)
            grad = self.get_grad(loss, delta).contiguous()
            if fused:
                # momentum first (the neighbours' look-ahead uses the NEW momentum, vmifgsm.py:89-92), the
                # delta step only after the variance of the OLD delta has been sampled
                m_new = torch.empty_like(grad)
                _hip.momentum(grad, momentum if isinstance(momentum, torch.Tensor) else None, m_new, self.decay,
                              variance=variance if isinstance(variance, torch.Tensor) else None)
                momentum = m_new
                variance = self.get_variance(data, delta, label, grad, momentum)
                delta = self.update_delta(delta, data, momentum, self.alpha)
            else:
                momentum = self.get_momentum(grad + variance, momentum)
                variance = self.get_variance(data, delta, label, grad, momentum)
                delta = self.update_delta(delta, data, momentum, self.alpha)
        return delta.detach()
