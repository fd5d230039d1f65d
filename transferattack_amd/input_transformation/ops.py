"""OPS (Guo et al., CVPR 2025) -- operator-perturbation-based stochastic optimisation: the update direction is the mean
gradient over sampled neighbours of the adversarial point (uniform noise at several radii, drawn once per batch) and,
for every neighbour, over sampled COMPOSITIONS of two to four basic operators (flips, circular shifts, rotations by fixed
angles, scalings, resize-pad-resize at ten rates) -- 900 gradients per iteration at the official sample counts, plus
the plain one.  Mirror of transferattack/input_transformation/ops.py:33-220.

Draws follow the reference generator by generator: python ``random`` picks operators and samples neighbours /
compositions, numpy the shift steps, torch the resize-pad geometry (and, once per batch, the neighbour noise).  A
composition applies its LAST drawn operator first (the reference folds them with ``f(g(x))``).
HIP: every resize-pad-resize is ``ta_dim_fwd`` / ``ta_dim_bwd`` (rates up to 2.9), the gradients are summed with
``ta_grad_accumulate`` and the averaged gradient goes through the fused momentum / projected step; flips, rolls and the
nearest-neighbour rotations (torchvision's ``functional.rotate`` default) are device ops."""
import math
import random

import numpy as np
import torch
import torch.nn.functional as F

from .. import _hip
from ..attack import Attack
from ..transforms import DimResizePad
from .dem import dem_draw


def identity(x):
    return x


def vertical_flip(x):
    return x.flip(dims=(2,))


def horizontal_flip(x):
    return x.flip(dims=(3,))


def vertical_shift(x):
    return x.roll(int(np.random.randint(low=0, high=x.shape[2], dtype=np.int32)), dims=2)


def horizontal_shift(x):
    return x.roll(int(np.random.randint(low=0, high=x.shape[3], dtype=np.int32)), dims=3)


class scaling:
    def __init__(self, scale):
        self.scale = scale

    def __call__(self, x):
        return x / self.scale


class dim:
    """resize-pad-resize to ``resize_rate`` x the image size, always applied, one geometry per call (ops.py:186-205)"""

    def __init__(self, resize_rate=1.1, diversity_prob=0.5):
        self.resize_rate, self.diversity_prob = resize_rate, diversity_prob

    def __call__(self, x):
        return DimResizePad.apply(x, *dem_draw(x.shape[-1], self.resize_rate))


class rotate:
    """torchvision.transforms.functional.rotate(x, angle) with its defaults: nearest neighbour, zero fill, no expansion.
    The sampling grid is built as torchvision 0.13 builds it (pixel centres relative to the image centre times the
    inverse rotation, normalised by half the size), once per (angle, size, device)."""

    def __init__(self, angle):
        self.angle = angle
        self._grids = {}

    def _grid(self, h, w, like):
        key = (h, w, str(like.device), like.dtype)
        if key not in self._grids:
            rot = math.radians(-self.angle)
            theta = torch.tensor([math.cos(rot), math.sin(rot), 0.0, -math.sin(rot), math.cos(rot), 0.0],
                                 dtype=like.dtype, device=like.device).reshape(1, 2, 3)
            base = torch.empty(1, h, w, 3, dtype=like.dtype, device=like.device)
            base[..., 0].copy_(torch.linspace(-w * 0.5 + 0.5, w * 0.5 + 0.5 - 1, steps=w, device=like.device))
            base[..., 1].copy_(torch.linspace(-h * 0.5 + 0.5, h * 0.5 + 0.5 - 1, steps=h, device=like.device).unsqueeze_(-1))
            base[..., 2].fill_(1)
            rescaled = theta.transpose(1, 2) / torch.tensor([0.5 * w, 0.5 * h], dtype=like.dtype, device=like.device)
            self._grids[key] = base.view(1, h * w, 3).bmm(rescaled).view(1, h, w, 2)
        return self._grids[key]

    def __call__(self, x):
        h, w = x.shape[-2], x.shape[-1]
        grid = self._grid(h, w, x).expand(x.shape[0], h, w, 2)
        return F.grid_sample(x, grid, mode="nearest", padding_mode="zeros", align_corners=False)


class OPS(Attack):
    """Official arguments: epsilon=16/255, alpha=epsilon/epoch, epoch=10, decay=1.0, beta=2., num_sample_neighbor=30,
    num_sample_operator=30, sample_levels=range(2, 5), sample_ratios=0.25 .. 1.5 in steps of 0.25."""

    def __init__(self, model_name, epsilon=16/255, beta=2., epoch=10, num_sample_neighbor=30, num_sample_operator=30,
                 sample_levels=range(2, 5), sample_ratios=np.arange(0., 1.5, 0.25) + 0.25, decay=1., targeted=False,
                 random_start=False, norm='linfty', loss='crossentropy', device=None, attack='OPS', **kwargs):
        super().__init__(attack, model_name, epsilon, targeted, random_start, norm, loss, device)
        self._schedule(epsilon / epoch, epoch, decay)
        self.using_sampling = (num_sample_operator * num_sample_neighbor > 0)
        if self.using_sampling:
            self.num_sample_operator, self.num_sample_neighbor = num_sample_operator, num_sample_neighbor
            self.basic_ops = ([identity, vertical_flip, horizontal_flip, vertical_shift, horizontal_shift]
                              + [rotate(a) for a in (5, -5, 15, -15, 45, -45, 90, -90, 180)]
                              + [scaling(s) for s in range(2, 9)]
                              + [dim(r) for r in (1.1, 1.3, 1.5, 1.7, 1.9, 2.1, 2.3, 2.5, 2.7, 2.9)])
            self.sample_levels = sample_levels
            self.op_list, self.eps_list = [], []
            self.num_extra_ops, self.num_extra_eps = len(self.basic_ops), num_sample_neighbor
            self.sample_radius = beta * epsilon * sample_ratios

    @property
    def op_num(self):
        return len(self.op_list)

    @property
    def eps_num(self):
        return len(self.eps_list)

    def get_new_ops(self, k=2):
        chosen = random.choices(self.basic_ops, k=k)

        def composed(x):
            for op in reversed(chosen):                    # f(g(x)) folded left to right: the last drawn runs first
                x = op(x)
            return x
        return composed

    def expand_op_list(self, k=2):
        for _ in range(self.num_extra_ops):
            self.op_list.append(self.get_new_ops(k=k))

    def init_op_list(self):
        self.op_list = []
        for level in self.sample_levels:
            if level == 1:
                self.op_list.append(self.basic_ops.copy())
            else:
                self.expand_op_list(level)

    def expand_eps_list(self, delta, radius=1.):
        shape = (self.num_extra_eps, *delta.shape[1:])
        self.eps_list.extend(torch.zeros(shape).uniform_(-radius, radius).to(self.device))       # host draw (ops.py:85)

    def init_eps_list(self, delta):
        self.eps_list = []
        for radius in self.sample_radius:
            self.expand_eps_list(delta, radius)

    def get_surrogate_gradient(self, data, delta, label, **kwargs):
        return self.get_grad(self.get_loss(self.get_logits(data + delta), label), delta)

    def get_averaged_gradient(self, data, delta, label, **kwargs):
        total = self.get_surrogate_gradient(data, delta, label).contiguous()
        if not self.using_sampling:
            return total
        for eps in random.sample(self.eps_list, min(self.num_sample_neighbor, self.eps_num)):
            x_near = data + delta + eps
            self.init_op_list()
            for op in random.sample(self.op_list, min(self.num_sample_operator, self.op_num)):
                grad = self.get_grad(self.get_loss(self.get_logits(op(x_near)), label), delta)
                _hip.grad_accumulate(total, grad.contiguous(), first=False)
        return total / (self.num_sample_neighbor * self.num_sample_operator + 1)

    def forward(self, data, label, **kwargs):
        data, label = self._to_device(data, label)
        delta = self.init_delta(data)
        if self.using_sampling:
            self.init_eps_list(delta)
        momentum = 0
        fused = self._can_fuse_update()
        for _ in range(self.epoch):
            averaged = self.get_averaged_gradient(data, delta, label)
            if fused:
                momentum = self._fused_update(averaged, momentum, delta, data)
            else:
                momentum = self.get_momentum(averaged, momentum)
                delta = self.update_delta(delta, data, momentum, self.alpha)
        return delta.detach()
