"""ANDA (Fang et al., 2024) -- asymptotically normal distribution attack: one image at a time, the gradient is taken on
``n_ens`` translated copies (a sqrt(n_ens) x sqrt(n_ens) lattice of affine shifts up to ``aug_max``), the running mean
of ALL gradients seen so far drives a sign step, and the deviations are kept so that a final perturbation can also be
SAMPLED from the fitted normal.  Mirror of transferattack/gradient/anda.py:45-210 (single-process ANDA; batch size 1).
The translations are ``affine_grid`` + ``grid_sample`` on the device, the statistics the method's own tensor arithmetic;
there is no momentum / projected-step hook on this path (the method clips to [0, 1] and the eps-ball itself)."""
import math

import numpy as np
import torch
import torch.nn.functional as F

from ..attack import Attack


class ANDA(Attack):
    """Official arguments: epsilon=16/255, alpha=1.6/255, epoch=10, n_ens=25, aug_max=0.3, sample=False."""

    def __init__(self, model_name, epsilon=16/255, alpha=1.6/255, epoch=10, n_ens=25, aug_max=0.3, sample=False,
                 targeted=False, random_start=False, norm='linfty', loss='crossentropy', device=None, attack='ANDA', **kwargs):
        super().__init__(attack, model_name, epsilon, targeted, random_start, norm, loss, device)
        self._schedule(alpha, epoch, 0)
        self.n_ens, self.aug_max, self.sample = n_ens, aug_max, sample
        side = int(math.sqrt(n_ens))
        assert side * side == n_ens, "n_ens must be square number."
        self.thetas = self.get_thetas(side, -aug_max, aug_max)

    def get_theta(self, i, j):
        return torch.tensor([[[1, 0, i], [0, 1, j]]], dtype=torch.float)

    def get_thetas(self, n, min_r=-0.5, max_r=0.5):
        shifts = torch.linspace(min_r, max_r, n)
        return torch.cat([self.get_theta(i, j) for i in shifts for j in shifts], dim=0)

    def transform(self, thetas, data):
        grids = F.affine_grid(thetas, data.size(), align_corners=False).to(data.device)
        return F.grid_sample(data, grids, align_corners=False)

    def get_loss(self, logits, labels):
        return F.cross_entropy(logits, labels, reduction="sum")

    def forward(self, data, label, **kwargs):
        assert data.shape[0] == 1, "ANDA currently only supports batchsize=1"
        assert (label.shape[1] if label.ndim == 2 else label.shape[0]) == 1, "ANDA currently only supports batchsize=1"
        data, label = self._to_device(data, label)
        xt = data.clone()
        lower, upper = data - self.epsilon, data + self.epsilon
        stat = ANDA_STATISTICS(data_shape=(1,) + tuple(data.shape[1:]), device=self.device)

        def project(x):
            return torch.max(torch.min(torch.clamp(x, 0.0, 1.0), upper), lower).detach()

        sampled = None
        for it in range(self.epoch):
            copies = xt.repeat(self.n_ens, 1, 1, 1).requires_grad_(True)
            logits = self.get_logits(self.transform(thetas=self.thetas, data=copies))
            stat.collect_stat(self.get_grad(self.get_loss(logits, label.repeat(self.n_ens)), copies))
            if self.sample and it == self.epoch - 1:
                drawn = stat.sample(n_sample=1, scale=1)
                sampled = project(self.alpha * drawn.squeeze().sign() + xt)
            xt = project(xt + self.alpha * stat.noise_mean.sign())
        adv = sampled if self.sample else xt
        return (adv.detach().clone() - data).detach()


class ANDA_STATISTICS:
    """running mean of the collected gradients and the stack of their deviations (a square root of the covariance)"""

    def __init__(self, device, data_shape=(1, 3, 224, 224)):
        self.data_shape, self.device = data_shape, device
        self.clear()

    def clear(self):
        self.n_models = 0
        self.noise_mean = torch.zeros(self.data_shape, dtype=torch.float).to(self.device)
        self.noise_cov_mat_sqrt = torch.empty((0, int(np.prod(self.data_shape))), dtype=torch.float).to(self.device)

    def collect_stat(self, noise):
        assert noise.device == self.noise_cov_mat_sqrt.device
        seen, new = self.n_models, noise.shape[0]
        mean = self.noise_mean * seen / (seen + new) + noise.data.sum(dim=0, keepdim=True) / (seen + new)
        self.noise_cov_mat_sqrt = torch.cat((self.noise_cov_mat_sqrt, (noise.data - mean).view(new, -1)), dim=0)
        self.noise_mean = mean
        self.n_models = seen + new

    def sample(self, n_sample=1, scale=0.0, seed=None):
        if seed is not None:
            torch.manual_seed(seed)
        if scale == 0.0:
            assert n_sample == 1
            return self.noise_mean.unsqueeze(0)
        assert scale == 1.0
        k = self.noise_cov_mat_sqrt.shape[0]
        draw = self.noise_cov_mat_sqrt.new_empty((n_sample, k), requires_grad=False).normal_().matmul(self.noise_cov_mat_sqrt)
        draw /= (k - 1) ** 0.5
        return (self.noise_mean.unsqueeze(0) + scale * draw.reshape(n_sample, *self.data_shape)).reshape(n_sample, *self.data_shape)
