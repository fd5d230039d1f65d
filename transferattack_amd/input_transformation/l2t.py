"""L2T (Zhu et al., CVPR 2024) -- learning to transform: every iteration draws ``num_scale`` pairs of operations from a
learned categorical distribution over 98 candidate transformations, applies each pair in sequence to the adversarial
batch (most operations return several copies, so the batch multiplies), averages the losses, and -- besides the MI-FGSM
step on delta -- moves the distribution's logits along the gradient of the probability-weighted losses (REINFORCE-like,
learning rate 0.01).  Mirror of transferattack/input_transformation/l2t.py:16-529.

The candidates and how each runs here:
    rotate(a, 5)       copies rotated by a / 2^i, nearest neighbour (torchvision ``functional.rotate`` defaults)   device gather
    sim(k)             x / 2^i, i < k                                                                            ``ta_scale_copies_fwd/bwd``
    dim(r)             resize-pad-resize at rate r, always applied                                                ``ta_dim_fwd/bwd``
    blockshuffle(b)    10 copies, b x b blocks of random sizes shuffled along both axes                           device split / cat
    admix(m, s)        x + s x[perm] for m permutations, then three scales                                        ``ta_admix_fwd/bwd``
    ide(p...)          one drop-out copy per probability, rescaled to the input's mean                           device op
    masked(b)          5 copies, one random block of the b x b lattice blanked in each                           device op
    ssm(rho)           10 spectrum-perturbed views                                                                ``ta_dct_pair`` (spectrum.spectrum_view)
    crop(r)            5 central crops from ratio r up to 1, resized to 224 (torchvision ``resized_crop``)         device bilinear resize
    affine(o)          5 sub-pixel translations, nearest neighbour (torchvision ``functional.affine``)            device gather
Draws follow the reference: ``torch.multinomial`` picks the pair (host), then each operation draws as its reference
counterpart does (python ``random`` / numpy / torch, in that order of appearance).  The torchvision sampling grids are
built as torchvision 0.13 builds them (see ``ops.rotate``)."""
import math
import random

import numpy as np
import torch
import torch.nn.functional as F

from .. import spectrum
from ..attack import Attack
from ..transforms import AdmixCopies, DimResizePad, ScaleCopies
from .dem import dem_draw

softmax = torch.nn.Softmax(dim=0)

# where the element-wise random tensors of the operations (ssm's spectrum mask, ide's drop-out mask) are drawn: None = on
# the input's device (product mode); 'cpu' while a seeded parity run is in progress (L2T.forward sets it when the
# attack's ``noise_source`` is set), so that the draws are the reference's host stream whatever device computes
_draw_device = None


def _rand_like(x):
    return torch.rand_like(x) if _draw_device is None else torch.rand(x.shape, device=_draw_device).to(x.device)


def select_op(op_params, num_ops):
    return torch.multinomial(softmax(op_params), num_ops, replacement=True).tolist()


def trace_prob(op_params, op_ids):
    probs = softmax(op_params)
    tp = 1
    for idx in op_ids:
        tp = tp * probs[idx]
    return tp


def _affine_sample(x, matrix, mode="nearest"):
    """torchvision 0.13 ``functional_tensor.affine`` / ``rotate``: the sampling grid of an inverse affine matrix given in
    pixels about the image centre, then grid_sample with zero padding"""
    h, w = x.shape[-2], x.shape[-1]
    theta = torch.tensor(matrix, dtype=x.dtype, device=x.device).reshape(1, 2, 3)
    base = torch.empty(1, h, w, 3, dtype=x.dtype, device=x.device)
    base[..., 0].copy_(torch.linspace(-w * 0.5 + 0.5, w * 0.5 + 0.5 - 1, steps=w, device=x.device))
    base[..., 1].copy_(torch.linspace(-h * 0.5 + 0.5, h * 0.5 + 0.5 - 1, steps=h, device=x.device).unsqueeze_(-1))
    base[..., 2].fill_(1)
    rescaled = theta.transpose(1, 2) / torch.tensor([0.5 * w, 0.5 * h], dtype=x.dtype, device=x.device)
    grid = base.view(1, h * w, 3).bmm(rescaled).view(1, h, w, 2)
    return F.grid_sample(x, grid.expand(x.shape[0], h, w, 2), mode=mode, padding_mode="zeros", align_corners=False)


def _rotate(x, angle):
    rot = math.radians(-angle)
    return _affine_sample(x, [math.cos(rot), math.sin(rot), 0.0, -math.sin(rot), math.cos(rot), 0.0])


def _translate(x, tx, ty):
    """``functional.affine(x, angle=0, translate=[tx, ty], scale=1, shear=0)``: torchvision's inverse matrix for that case"""
    a, b, c, d = 1.0, -0.0, 0.0, 1.0               # cos(0)/cos(0), -cos(0) tan(0)/cos(0) - sin(0), sin(0)/cos(0), ... + cos(0)
    matrix = [d, -b, 0.0, -c, a, 0.0]
    matrix[2] += matrix[0] * (-tx) + matrix[1] * (-ty)
    matrix[5] += matrix[3] * (-tx) + matrix[4] * (-ty)
    return _affine_sample(x, matrix)


def identity(x):
    return x


class rotate:
    def __init__(self, angle, num_scale):
        self.angle, self.num_scale = angle, num_scale

    def __call__(self, x):
        return torch.cat([_rotate(x, self.angle / (2 ** i)) for i in range(self.num_scale)])


class sim:
    def __init__(self, num_copy):
        self.num_copy = num_copy

    def __call__(self, x):
        return ScaleCopies.apply(x, self.num_copy)


class dim:
    def __init__(self, resize_rate=1.1, diversity_prob=0.5):
        self.resize_rate, self.diversity_prob = resize_rate, diversity_prob

    def __call__(self, x):
        return DimResizePad.apply(x, *dem_draw(x.shape[-1], self.resize_rate))


class blockshuffle:
    def __init__(self, num_block=3, num_scale=10):
        self.num_block, self.num_scale = num_block, num_scale

    def get_length(self, length):
        rand = np.random.uniform(size=self.num_block)
        cut = np.round(rand / rand.sum() * length).astype(np.int32)
        cut[cut.argmax()] += length - cut.sum()
        return tuple(int(v) for v in cut)

    def shuffle_single_dim(self, x, axis):
        strips = list(x.split(self.get_length(x.size(axis)), dim=axis))
        random.shuffle(strips)
        return strips

    def shuffle(self, x):
        axes = [2, 3]
        random.shuffle(axes)
        return torch.cat([torch.cat(self.shuffle_single_dim(strip, axes[1]), dim=axes[1])
                          for strip in self.shuffle_single_dim(x, axes[0])], dim=axes[0])

    def __call__(self, x, **kwargs):
        return torch.cat([self.shuffle(x) for _ in range(self.num_scale)])


class admix:
    def __init__(self, num_admix=3, admix_strength=0.2, num_scale=3):
        self.num_admix, self.admix_strength, self.num_scale = num_admix, admix_strength, num_scale

    def __call__(self, x):
        perm = torch.cat([torch.randperm(x.size(0)) for _ in range(self.num_admix)]).to(x.device)
        return AdmixCopies.apply(x, perm, self.num_admix, self.num_scale, self.admix_strength)


class ide:
    def __init__(self, dropout_prob=(0, 0.1, 0.2, 0.3, 0.4, 0.5)):
        self.dropout_prob = dropout_prob

    def __call__(self, x):
        if _draw_device is None:
            return torch.cat([F.dropout(x, p=prob, training=True) * (1 - prob) for prob in self.dropout_prob])
        # the same arithmetic with the keep / (1 - p) mask drawn on the host: dropout(x) is x * mask
        masks = [F.dropout(torch.ones(x.shape, device=_draw_device), p=prob, training=True).to(x.device) for prob in self.dropout_prob]
        return torch.cat([x * mask * (1 - prob) for mask, prob in zip(masks, self.dropout_prob)])


class masked:
    def __init__(self, num_block, num_scale=5):
        self.num_block, self.num_scale = num_block, num_scale

    def blockmask(self, x, choice=-1):
        first, second = x.shape[2], x.shape[3]
        assert first == second, "the reference defines the block lattice for square inputs only (l2t.py:205-207)"
        edges = [round(first / self.num_block * i) for i in range(self.num_block + 1)]
        blanked = x.clone()
        a, b = random.randint(0, self.num_block - 1), random.randint(0, self.num_block - 1)
        blanked[:, :, edges[a]:edges[a + 1], edges[b]:edges[b + 1]] = 0
        return blanked

    def __call__(self, x):
        return torch.cat([self.blockmask(x) for _ in range(self.num_scale)])


class ssm:
    def __init__(self, rho=0.5, num_spectrum=10):
        self.epsilon, self.rho, self.num_spectrum = 16 / 255, rho, num_spectrum

    def __call__(self, x):
        views = []
        for _ in range(self.num_spectrum):
            gauss = (torch.randn(x.size()[0], 3, 224, 224) * self.epsilon).to(x.device)          # host draw (l2t.py:337)
            mask = _rand_like(x) * 2 * self.rho + 1 - self.rho
            views.append(spectrum.spectrum_view(x, gauss, mask))
        return torch.cat(views)


class crop:
    def __init__(self, ratio, num_scale=5):
        self.ratio, self.num_scale = ratio, num_scale

    def crop(self, x, ratio):
        rows, cols = int(x.shape[3] * ratio), int(x.shape[2] * ratio)      # the reference's (height, width) of resized_crop
        top, left = (x.shape[3] - rows) // 2, (x.shape[2] - cols) // 2
        return F.interpolate(x[..., top:top + rows, left:left + cols], size=(224, 224), mode="bilinear", align_corners=False)

    def __call__(self, x):
        return torch.cat([self.crop(x, self.ratio + (1 - self.ratio) * (i + 1) / self.num_scale) for i in range(self.num_scale)])


class affine:
    def __init__(self, offset, num_scale=5):
        self.offset, self.num_scale = offset, num_scale

    def __call__(self, x):
        return torch.cat([_translate(x, self.offset * (i + 1) / self.num_scale, self.offset * (i + 1) / self.num_scale)
                          for i in range(self.num_scale)])


op_list = ([identity]
           + [rotate(a, 5) for a in range(30, 301, 30)]
           + [sim(k) for k in range(1, 11)]
           + [dim(r) for r in (1.1, 1.15, 1.2, 1.25, 1.3, 1.35, 1.4, 1.45, 1.5, 1.55)]
           + [blockshuffle(b) for b in range(3, 13)]
           + [admix(m, s) for s in (0.2, 0.4) for m in range(1, 6)]
           + [ide(p) for p in ([0.1], [0.1, 0.2], [0.1, 0.2, 0.3], [0.1, 0.2, 0.3, 0.4], [0.1, 0.2, 0.3, 0.4, 0.5],
                               [0.2, 0.3, 0.4, 0.5], [0.1, 0.3, 0.4, 0.5], [0.1, 0.2, 0.4, 0.5], [0.1, 0.2, 0.3, 0.5],
                               [0.1, 0.2, 0.3, 0.4])]
           + [masked(b) for b in (2, 4, 6, 8, 10, 3, 5, 7, 9, 11)]
           + [ssm(r) for r in (0.2, 0.4, 0.5, 0.6, 0.8, 0.1, 0.3, 0.7, 0.9)]
           + [crop(r) for r in (0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9)]
           + [affine(o) for o in (0.5, 0.55, 0.6, 0.65, 0.7, 0.75, 0.8, 0.85, 0.9)])


class RWAug_Search:
    def __init__(self, n, idxs):
        self.n, self.idxs, self.op_list = n, idxs, op_list

    def __call__(self, img):
        assert len(self.idxs) == self.n
        for idx in self.idxs:
            img = op_list[idx](img)
        return img


class L2T(Attack):
    """Official arguments: epsilon=16/255, alpha=1.6/255, epoch=10, decay=1, num_scale=3."""

    def __init__(self, model_name, epsilon=16/255, alpha=1.6/255, epoch=10, decay=1., num_scale=3, targeted=False,
                 random_start=False, norm='linfty', loss='crossentropy', device=None, attack='L2T', **kwargs):
        super().__init__(attack, model_name, epsilon, targeted, random_start, norm, loss, device)
        self._schedule(alpha, epoch, decay)
        self.num_scale = num_scale

    def get_loss(self, logits, label, num_copy):
        label = label.repeat(num_copy)
        return -self.loss(logits, label) if self.targeted else self.loss(logits, label)

    def transform(self, x, **kwargs):
        return kwargs['search'](x)

    def forward(self, data, label, **kwargs):
        global _draw_device
        keep, _draw_device = _draw_device, ('cpu' if self.noise_source is not None else None)
        try:
            return self._attack(data, label)
        finally:
            _draw_device = keep

    def _attack(self, data, label):
        data, label = self._to_device(data, label)
        ops_num, learning_rate = 2, 0.01
        aug_param = torch.nn.Parameter(torch.zeros(len(op_list), requires_grad=True), requires_grad=True)
        delta = self.init_delta(data)
        momentum = 0
        fused = self._can_fuse_update()
        for _ in range(self.epoch):
            aug_probs, losses = [], []
            for _ in range(self.num_scale):
                chosen = select_op(aug_param, ops_num)
                aug_probs.append(trace_prob(aug_param, chosen))
                logits = self.get_logits(self.transform(data + delta, search=RWAug_Search(ops_num, chosen)))
                copies = math.floor((len(logits) + 0.01) / len(label))
                losses.append(self.get_loss(logits, label, copies).reshape(1))
            grad = self.get_grad(torch.sum(torch.cat(losses)) / self.num_scale, delta)
            # the policy's logits move along d/d(aug_param) of the probability-weighted losses (the losses enter as values)
            weighted = torch.cat([aug_probs[i] * losses[i].reshape(1).to(aug_probs[i].device) for i in range(self.num_scale)])
            aug_grad = torch.autograd.grad(torch.sum(weighted) / self.num_scale, aug_param, retain_graph=False,
                                           create_graph=False)[0]
            aug_param = aug_param + learning_rate * aug_grad
            if fused:
                momentum = self._fused_update(grad, momentum, delta, data)
            else:
                momentum = self.get_momentum(grad, momentum)
                delta = self.update_delta(delta, data, momentum, self.alpha)
        return delta.detach()
