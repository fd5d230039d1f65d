"""SU (Wei et al., CVPR 2023) -- self-universality for targeted transfer: every iteration attacks the image AND a random
local crop of it (10 % of the area, resized to full size) with the same perturbation, maximises the target logit of
both, and pulls their intermediate features together (cosine similarity at one chosen layer, weight ``coef``); the
batch of 2N goes through a DI-style random resize-and-pad (nearest, probability 0.7), the gradient through a 5 x 5
Gaussian (TI).  Mirror of transferattack/input_transformation/su.py:39-182.

HIP: the TI smoothing is ``ta_depthwise_conv2d_same`` (which also leaves the |g| tile sums), momentum and the
projected step are the fused update.  The local crop is torchvision's ``RandomResizedCrop`` -- restated here as
torchvision 0.13 defines it: up to ten (area, log-uniform aspect) draws from torch's host generator, one crop for the
whole batch, bilinear resize without antialiasing -- and, like the DI step (numpy draws), a device op.  The feature
layer is resolved by surrogate name exactly as the reference does (``_target_layer``)."""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _hip
from ..gradient.mifgsm import MIFGSM
from ..utils import img_height
from .tim import _gaussian_profile


class LogitLoss(nn.Module):
    def forward(self, logits, labels):
        return (-1 * logits.gather(1, labels.unsqueeze(1)).squeeze(1)).mean()


class RandomResizedCrop:
    """torchvision.transforms.RandomResizedCrop(size, scale, ratio=(3/4, 4/3)) for float NCHW batches"""

    def __init__(self, size, scale=(0.08, 1.0), ratio=(3. / 4., 4. / 3.)):
        self.size = (size, size) if isinstance(size, int) else tuple(size)
        self.scale, self.ratio = scale, ratio

    @staticmethod
    def get_params(img, scale, ratio):
        height, width = img.shape[-2], img.shape[-1]
        area = height * width
        log_ratio = torch.log(torch.tensor(ratio))
        for _ in range(10):
            target_area = area * torch.empty(1).uniform_(scale[0], scale[1]).item()
            aspect_ratio = torch.exp(torch.empty(1).uniform_(log_ratio[0], log_ratio[1])).item()
            w = int(round(math.sqrt(target_area * aspect_ratio)))
            h = int(round(math.sqrt(target_area / aspect_ratio)))
            if 0 < w <= width and 0 < h <= height:
                i = torch.randint(0, height - h + 1, size=(1,)).item()
                j = torch.randint(0, width - w + 1, size=(1,)).item()
                return i, j, h, w
        in_ratio = float(width) / float(height)               # fall back to a central crop
        if in_ratio < min(ratio):
            w = width
            h = int(round(w / min(ratio)))
        elif in_ratio > max(ratio):
            h = height
            w = int(round(h * max(ratio)))
        else:
            w, h = width, height
        return (height - h) // 2, (width - w) // 2, h, w

    def __call__(self, img):
        i, j, h, w = self.get_params(img, self.scale, self.ratio)
        return F.interpolate(img[..., i:i + h, j:j + w], size=list(self.size), mode="bilinear", align_corners=False)


class SU(MIFGSM):
    """Official arguments: epsilon=16/255, alpha=2/255, epoch=300, decay=1, coef=0.001, scale=(0.1, 0.0), depth=3
    (layer3 of ResNet-50), targeted."""

    def __init__(self, model_name, epsilon=16/255, alpha=2/255, epoch=300, decay=1., coef=0.001, scale=(0.1, 0.0), depth=3,
                 targeted=True, random_start=False, norm='linfty', loss='crossentropy', device=None, attack='SU', **kwargs):
        super().__init__(model_name, epsilon, alpha, epoch, decay, targeted, random_start, norm, loss, device, attack)
        self.model_name = model_name
        self.start, self.interval = scale
        self.coef, self.depth = coef, depth
        self.local_transform = RandomResizedCrop(img_height, scale=(self.start, self.start + self.interval))
        self.gaussian_kernel = self._TI_kernel()
        self.resize_rate, self.diversity_prob = 1.1, 0.7
        self._register_forward()
        self.loss_fn = LogitLoss()

    def _target_layer(self, model_name, depth):
        """the module whose output is compared (su.py:62-78): 'resnet50', 'vgg16_bn', 'densenet121', 'inception_v3'"""
        net = self.model[1]
        if model_name == 'resnet50':
            return getattr(net, 'layer{}'.format(depth))[-1]
        if model_name == 'vgg16_bn':
            return net.features[{1: 12, 2: 22, 3: 32, 4: 42}[depth]]
        if model_name == 'densenet121':
            return getattr(net.features, 'denseblock{}'.format(depth))
        if model_name == 'inception_v3':
            return getattr(net, {1: 'Conv2d_4a_3x3', 2: 'Mixed_5d', 3: 'Mixed_6e', 4: 'Mixed_7c'}[depth])
        return None

    def _register_forward(self):
        self.activations = []

        def keep(module, inputs, output):
            self.activations += [output]

        self._target_layer(self.model_name, self.depth).register_forward_hook(keep)

    def _DI(self, X_in):
        """random nearest-neighbour resize to [224, 246) and zero padding to 246 x 246 with probability 0.7 (numpy
        draws, all four taken whether or not the branch is; su.py:92-110) -- the surrogate's own Resize brings it back"""
        img_resize = int(img_height * self.resize_rate)
        rnd = np.random.randint(img_height, img_resize, size=1)[0]
        rem = img_resize - rnd
        pad_top = np.random.randint(0, rem, size=1)[0]
        pad_left = np.random.randint(0, rem, size=1)[0]
        if np.random.rand(1) <= 0.7:
            return F.pad(F.interpolate(X_in, size=(rnd, rnd)), (pad_left, rem - pad_left, pad_top, rem - pad_top),
                         mode='constant', value=0)
        return X_in

    def _TI_kernel(self):
        profile = _gaussian_profile(5, 3)
        plane = np.outer(profile, profile)
        plane = (plane / plane.sum()).astype(np.float32)
        return torch.from_numpy(np.stack([plane, plane, plane])[:, None]).to(self.device)

    def forward(self, data, label, **kwargs):
        if self.targeted:
            assert len(label) == 2
            label = label[1]
            used_coef = -1
        else:
            used_coef = 1
        data = data.clone().detach().to(self.device)
        label = label.clone().detach().to(self.device)
        batch = data.shape[0]
        delta = self.init_delta(data)
        momentum = 0
        fused = self._can_fuse_update()
        for _ in range(self.epoch):
            self.activations = []
            local = self.local_transform(data)
            logits = self.model(self._DI(torch.cat([data + delta, local + delta], dim=0)))
            classifier_loss = self.loss_fn(logits, torch.cat([label, label], dim=0))
            feats = self.activations[0]
            fs_loss = torch.mean(F.cosine_similarity(feats[:batch].view(batch, -1), feats[-batch:].view(batch, -1)))
            raw = self.get_grad(-(classifier_loss + self.coef * used_coef * fs_loss), delta).contiguous()
            grad = torch.empty_like(raw)
            _hip.depthwise_conv2d_same(raw, grad, self.gaussian_kernel[0, 0].contiguous())     # conv2d(pad 2, groups 3)
            if fused:
                momentum = self._fused_update(grad, momentum, delta, data)
            else:
                momentum = self.get_momentum(grad, momentum)
                delta = self.update_delta(delta, data, momentum, self.alpha)
        return delta.detach()
