"""US-MM (Wang et al., 2023) -- uniform scales and mix-mask: ``num_scale`` scalings between ``scale_low`` and
``scale_high`` of the input, each multiplied by a mask built from a shuffled batch member
((1 - r) + 2 r x[perm]), ``num_mix`` times; the copies are clamped to [0, 1] and cut off the graph, the gradient is
taken with respect to the COPIES and summed over them (so neither the scale nor the mask enters the chain rule).
Mirror of transferattack/input_transformation/usmm.py:34-99.  The copies are elementwise device ops; the copy-sum is
``ta_sum_members`` (which also leaves the |g| tile sums for the fused update)."""
import torch

from ..gradient.mifgsm import MIFGSM
from .. import _hip


class USMM(MIFGSM):
    """Official arguments: epsilon=16/255, alpha=1.6/255, epoch=10, decay=1., scale_low=0.1, scale_high=0.75, num_scale=5,
    num_mix=3, mix_range=0.5."""

    def __init__(self, model_name, epsilon=16/255, alpha=1.6/255, epoch=10, decay=1., scale_low=0.1, scale_high=0.75,
                 num_scale=5, num_mix=3, mix_range=0.5, targeted=False, random_start=False, norm='linfty',
                 loss='crossentropy', device=None, attack='USMM', **kwargs):
        super().__init__(model_name, epsilon, alpha, epoch, decay, targeted, random_start, norm, loss, device, attack)
        self.scale_low, self.scale_high, self.num_scale = scale_low, scale_high, num_scale
        self.num_mix, self.mix_range = num_mix, mix_range

    def transform(self, x, **kwargs):
        span = self.scale_high - self.scale_low
        scaled = [x * (self.scale_low + span * i / (self.num_scale - 1)) for i in range(self.num_scale)]
        copies = []
        for _ in range(self.num_mix):
            for one in scaled:                        # one host permutation per copy (usmm.py:48), scales innermost
                mask = (1 - self.mix_range) * torch.ones_like(x) + 2 * self.mix_range * x[torch.randperm(x.size(0))].detach()
                copies.append(one * mask)
        return torch.clamp(torch.cat(copies, dim=0), 0, 1)

    def get_loss(self, logits, label):
        label = label.repeat(self.num_scale * self.num_mix)
        return -self.loss(logits, label) if self.targeted else self.loss(logits, label)

    @staticmethod
    def _sum_ascending(chunks, like):
        """((c0 + c1) + c2) + ... -- the order of the reference's ``torch.sum(torch.stack(...), dim=0)`` (usmm.py:88; ATen
        adds the rows of a short outer reduction one after another).  ``ta_sum_members`` adds its LAST pointer first and
        takes eight operands, so the copies go in reversed, eight at a time, the running sum leading the next call."""
        acc, pending = None, list(chunks)
        while pending:
            room = 8 - (acc is not None)
            head, pending = pending[:room], pending[room:]
            operands = ([acc] if acc is not None else []) + [c.contiguous() for c in head]
            acc = torch.empty_like(like)
            _hip.sum_members(operands[::-1], acc)
        return acc

    def forward(self, data, label, **kwargs):
        data, label = self._to_device(data, label)
        delta = self.init_delta(data)
        momentum = 0
        fused = self._can_fuse_update()
        for _ in range(self.epoch):
            x_trans = self.transform(data + delta, momentum=momentum).clone().detach().to(self.device)
            x_trans.requires_grad = True
            grad_copies = self.get_grad(self.get_loss(self.get_logits(x_trans), label), x_trans).contiguous()
            grad = self._sum_ascending(grad_copies.split(data.shape[0]), data)
            if fused:
                momentum = self._fused_update(grad, momentum, delta, data)
            else:
                momentum = self.get_momentum(grad, momentum)
                delta = self.update_delta(delta, data, momentum, self.alpha)
        return delta.detach()
