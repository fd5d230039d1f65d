"""VMI-FGSM (Wang & He, CVPR 2021) -- variance tuning: the momentum uses grad + v where
v = mean_i grad(x + delta + U(-beta*eps, beta*eps)) - grad over ``num_neighbor`` samples.
Mirror of transferattack/gradient/vmifgsm.py:28-97.

HIP path: neighbour sampling is one kernel (``ta_vmi_neighbor``: x + delta + Philox noise, nothing but the
sample itself is written), accumulation/finalisation are ``ta_grad_accumulate`` / ``ta_variance_finalize``
and ``grad + variance`` is folded into the fused update (its ``v`` operand), so no grad-sized temporary is
ever materialised besides the accumulator.

When nothing sits between the sample and the backbone but the surrogate's own Normalize (the plain class on a 224-pixel
surrogate: no look-ahead transform, no overridden hooks, no module hooks on the preprocessing layer), the chain per
neighbour shrinks from four passes (sample 12 + normalize 8 ... normalize backward 8 + accumulate 12 = 40 B/element) to
two (``ta_vmi_neighbor_normalized`` 12 + ``ta_normalize_bwd_accumulate`` 12 = 24): the sample is written already
normalised and handed to ``self.model[1]``, the backbone's input gradient goes straight into the accumulator.  The
current gradient's Normalize backward adds the |grad + variance| tile sums, and momentum + step are ONE ``ta_mi_update``
after the neighbours (the variance is sampled around the old delta either way, vmifgsm.py:89-95), so the update makes no
pass of its own over the gradient.  Rounding points are those of the separate kernels: same bits.
"""
import os

import torch
import torch.nn as nn

from ..attack import Attack
from .. import _hip
from ..transforms import Neighbor
from ..utils import PreprocessingModel


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

    def _normalize_folds(self, data):
        """(mean, std) of the surrogate's Normalize if the neighbour chain may bypass ``self.model[0]`` -- i.e. if that layer
        is the only thing between the sample and the backbone and nobody observes it -- else None."""
        cls = type(self)
        if not (cls.transform is Attack.transform and cls.get_logits is Attack.get_logits and cls.get_grad is Attack.get_grad
                and cls.get_variance is VMIFGSM.get_variance and self._can_fuse_update()):
            return None
        model = self.model
        if not (isinstance(model, nn.Sequential) and len(model) == 2 and isinstance(model[0], PreprocessingModel)):
            return None
        pre = model[0]
        if pre.resize.size != data.shape[-1] or data.shape[-1] != data.shape[-2] or data.dim() != 4 or data.dtype != torch.float32:
            return None
        for mod in (model, pre, pre.resize, pre.normalize):
            if mod._forward_hooks or mod._forward_pre_hooks or mod._backward_hooks or getattr(mod, "_backward_pre_hooks", None):
                return None
        return pre.normalize.mean.reshape(-1).contiguous(), pre.normalize.std.reshape(-1).contiguous()

    def _backbone_grad(self, y, label, stack=1):
        """d loss / d y for an already normalised input ``y`` (a fresh leaf) through ``self.model[1]``.  ``stack`` > 1: ``y``
        holds that many neighbour samples of the batch one after the other ([stack * N, ...]) and the loss is the SUM of the
        neighbours' own batch-mean losses (``get_loss`` per slice, as vmifgsm.py:50-53 calls it per neighbour), so every slice
        of the result is that neighbour's gradient with the reference's 1 / N scaling."""
        y.requires_grad_(True)
        logits = self.model[1](y)
        if stack == 1:
            loss = self.get_loss(logits, label)
        else:
            n = y.shape[0] // stack
            loss = self.get_loss(logits[:n], label)
            for j in range(1, stack):
                loss = loss + self.get_loss(logits[j * n:(j + 1) * n], label)
        return torch.autograd.grad(loss, y, retain_graph=False, create_graph=False)[0].contiguous()

    def _neighbor_stack(self, n):
        """How many of the ``num_neighbor`` samples go through the surrogate in ONE evaluation (round 5).  The samples of an
        iteration are independent (vmifgsm.py:46-58 loops over them only because autograd.grad is called per sample): stacking
        k of them runs the surrogate on k * N images -- fewer, larger launches and 1 / k of the host-side dispatch work per
        image; slices are accumulated in neighbour order, so the variance keeps its rounding sequence.  ``TA_VMI_STACK=k`` sets
        it for any surrogate (1 = one evaluation per neighbour, the reference's shape).  Default: stacking only where it was
        measured -- a transformer surrogate (configs[3]: ViT-B/16, +15 %, profiles/r05/bench_vmi_vit_b32_stack*.json), the largest
        divisor of num_neighbor with k * N <= 160 images; every other surrogate keeps the reference's one-by-one shape (its
        libraries may pick other algorithms for a k * N batch, and the activations of k * N images must fit)."""
        want = os.environ.get("TA_VMI_STACK", "")
        if want.isdigit() and int(want) >= 1:
            k = min(int(want), self.num_neighbor)
        elif "VisionTransformer" in type(self.model[1]).__name__:
            k = max(1, min(self.num_neighbor, 160 // max(n, 1)))
        else:
            k = 1
        while self.num_neighbor % k:
            k -= 1
        return k

    def _forward_folded(self, data, label, mean, std):
        delta = self.init_delta(data).detach()
        momentum, variance, x_adv = None, None, None
        for it in range(self.epoch):
            x_in = data + delta if x_adv is None else x_adv
            y = torch.empty_like(data)
            _hip.normalize_fwd(x_in.contiguous(), y, mean, std)
            gy = self._backbone_grad(y, label)
            grad = torch.empty_like(data)
            _hip.normalize_bwd(gy, grad, std, variance=variance)          # leaves the tile sums of |grad + variance|
            if self.grad_probe is not None:                               # test hook: sees what get_grad would return
                self.grad_probe(it, grad)
            acc = torch.empty_like(data)
            n, k = data.shape[0], self._neighbor_stack(data.shape[0])
            for base in range(0, self.num_neighbor, k):
                ys = torch.empty((k * n,) + tuple(data.shape[1:]), dtype=data.dtype, device=data.device)
                for j in range(k):                                  # draws in neighbour order: the Philox offsets of the
                    noise = None                                    # one-by-one loop
                    if self.noise_source is not None:
                        noise = self.noise_source(data.shape, -self.radius, self.radius).to(self.device).contiguous()
                    _hip.vmi_neighbor_normalized(data, delta, ys[j * n:(j + 1) * n], mean, std, self.radius, self.rng_seed,
                                                 self._next_offset(), noise)
                gys = self._backbone_grad(ys, label, stack=k)
                for j in range(k):                                  # accumulated in neighbour order
                    _hip.normalize_bwd_accumulate(gys[j * n:(j + 1) * n], acc, std, first=(base + j == 0))
            new_variance = torch.empty_like(data)
            _hip.variance_finalize(acc, grad, new_variance, self.num_neighbor)
            m_out = momentum if momentum is not None else (None if self.decay == 0 else torch.empty_like(data))
            if x_adv is None and it + 1 < self.epoch:
                x_adv = torch.empty_like(data)
            _hip.mi_update(grad, momentum, m_out, delta, data, self.decay, self.alpha, self.epsilon, variance=variance,
                           x_adv=x_adv if it + 1 < self.epoch else None, data_u8=self._byte_source_of(data))
            momentum, variance = m_out, new_variance
        return delta.detach()

    def forward(self, data, label, **kwargs):
        if self.targeted:
            assert len(label) == 2
            label = label[1]
        data = data.clone().detach().to(self.device)
        label = label.clone().detach().to(self.device)
        folds = self._normalize_folds(data)
        if folds is not None:
            data = data.contiguous()
            self._attach_byte_source(data)
            return self._forward_folded(data, label, *folds)
        self._attach_byte_source(data)
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
