"""MI-FGSM with the tricks of 'Bag of tricks to boost the adversarial transferability' (Bai et al., 2024):
RGI (random global momentum initialisation), the dual example, and the ensemble of dual examples.
Mirror of transferattack/gradient/mifgsm_with_tricks.py:15-266 -- same hooks, same draw order (every random start is
one ``init_delta`` draw), including the reference's quirks: RGI's pre-convergence momentum runs through all directions
and is divided by their number; the dual example's own delta is re-drawn every iteration, only the dual delta moves.
HIP: ``get_momentum`` / ``update_delta`` are the base class's kernels (``ta_momentum``, ``ta_update_delta_linf``),
random starts ``ta_init_delta_uniform``."""
import torch

from ..attack import Attack
from .. import _hip


class _TrickBase(Attack):
    def __init__(self, attack, model_name, epsilon, alpha, epoch, decay, targeted, random_start, norm, loss, device):
        super().__init__(attack, model_name, epsilon, targeted, random_start, norm, loss, device)
        self._schedule(alpha, epoch, decay)

    def _random_delta(self, data):
        """a fresh uniform start regardless of the constructor's flag (the reference toggles ``random_start`` around the
        call, e.g. mifgsm_with_tricks.py:141-144)"""
        keep, self.random_start = self.random_start, True
        try:
            return self.init_delta(data).to(self.device)
        finally:
            self.random_start = keep

    def _grad_at(self, data, delta, label, momentum):
        logits = self.get_logits(self.transform(data + delta, momentum=momentum))
        return self.get_grad(self.get_loss(logits, label), delta)


class RGMIFGSM(_TrickBase):
    """Official arguments: epsilon=16/255, alpha=1.6/255, epoch=10, decay=1., pre_epoch=5, s=10 (num_directions=5)."""

    def __init__(self, model_name, epsilon=16/255, alpha=1.6/255, epoch=10, decay=1., targeted=False, random_start=False,
                 norm='linfty', loss='crossentropy', device=None, attack='RGMIFGSM', pre_epoch=5, s=10,
                 num_directions=5, **kwargs):
        super().__init__(attack, model_name, epsilon, alpha, epoch, decay, targeted, random_start, norm, loss, device)
        self.pre_epoch, self.s, self.num_directions = pre_epoch, s, num_directions

    def forward(self, data, label, **kwargs):
        data, label = self._to_device(data, label)
        momentum = 0.
        self._random_delta(data)                              # the reference draws (and drops) one start before the directions
        for _ in range(self.num_directions):
            delta = self._random_delta(data)
            for _ in range(self.pre_epoch):
                momentum = self.get_momentum(self._grad_at(data, delta, label, momentum), momentum)
                delta = self.update_delta(delta, data, momentum, self.alpha * self.s)
        momentum = momentum / self.num_directions
        self.random_start = False                             # left that way by the reference as well (:63)
        delta = self.init_delta(data).to(self.device)
        for _ in range(self.epoch):
            momentum = self.get_momentum(self._grad_at(data, delta, label, momentum), momentum)
            delta = self.update_delta(delta, data, momentum, self.alpha)
        return delta.detach()


class DualMIFGSM(_TrickBase):
    """Official arguments: epsilon=16/255, alpha=1.6/255, epoch=10, decay=1."""

    def __init__(self, model_name, epsilon=16/255, alpha=1.6/255, epoch=10, decay=1., targeted=False, random_start=False,
                 norm='linfty', loss='crossentropy', device=None, attack='DualMIFGSM', **kwargs):
        super().__init__(attack, model_name, epsilon, alpha, epoch, decay, targeted, random_start, norm, loss, device)

    def forward(self, data, label, **kwargs):
        data, label = self._to_device(data, label)
        delta_dual = self.init_delta(data).clone().detach().to(self.device)
        momentum, momentum_dual = 0., 0.
        for _ in range(self.epoch):
            delta = self._random_delta(data)
            grad = self._grad_at(data, delta, label, momentum)
            self.random_start = False                         # (:144) -- later batches start the dual example from zero
            momentum = self.get_momentum(grad, momentum)
            momentum_dual = self.get_momentum(grad, momentum_dual, decay=self.decay)
            delta_dual = self.update_delta(delta_dual, data, momentum_dual, self.alpha)
        return delta_dual.detach()


class Ens_FGSM_MIFGSM(_TrickBase):
    """Official arguments: epsilon=16/255, alpha=1.6/255, epoch=10, decay=1. (num_d=5 random starts per iteration)."""

    def __init__(self, model_name, epsilon=16/255, alpha=1.6/255, epoch=10, decay=1., targeted=False, random_start=False,
                 norm='linfty', loss='crossentropy', device=None, attack='Ens_DualMIFGSM', num_d=5, **kwargs):
        super().__init__(attack, model_name, epsilon, alpha, epoch, decay, targeted, random_start, norm, loss, device)
        self.num_directions = num_d

    def forward(self, data, label, **kwargs):
        data, label = self._to_device(data, label)
        delta_dual = self.init_delta(data).clone().detach().to(self.device)
        momentum, momentum_dual = 0., 0.
        for _ in range(self.epoch):
            acc = torch.empty_like(data)
            for nd in range(self.num_directions):
                grad = self._grad_at(data, self._random_delta(data), label, momentum)
                self.random_start = False
                _hip.grad_accumulate(acc, grad.contiguous(), first=(nd == 0))
            grad = acc / self.num_directions
            momentum = self.get_momentum(grad, momentum)
            momentum_dual = self.get_momentum(grad, momentum_dual, decay=self.decay)
            delta_dual = self.update_delta(delta_dual, data, momentum_dual, self.alpha)
        return delta_dual.detach()
