"""SMER (Tang et al., CVPR 2024) -- stochastic mini-batch ensemble with learned member weights: each outer iteration
runs 4K inner MI steps, each on one member drawn from K-permutations (numpy generator); after every inner step the
per-member logit weights take one SGD step on -log(CE of the weighted logit mean).  The last inner momentum is the
"gradient" of the outer momentum.
Mirror of transferattack/ensemble/smer.py:36-138 (``Weight_Selection`` :130-138).  The weights and their optimiser
live on the attack object and persist across batches, as in the reference.

One difference in how the work is laid out, not in what is computed: the reference runs the drawn member twice per
inner step (once for the image gradient, once inside the all-member pass for the weight gradient).  The two passes
see the same input and weights, so here every member runs once and both gradients are taken from that graph."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..attack import Attack


class Weight_Selection(nn.Module):
    """one learnable scalar per ensemble member, initialised to 1 (smer.py:130-138)"""

    def __init__(self, weight_len):
        super().__init__()
        self.weight = nn.Parameter(torch.ones([weight_len]))

    def forward(self, x, index):
        return self.weight[index] * x


class SMER(Attack):
    """Official arguments: epsilon=16/255, alpha=1.6/255, epoch=10, decay=1.0, random_start=True."""

    def __init__(self, model_name, epsilon=16/255, alpha=1.6/255, epoch=10, decay=1.0, targeted=False,
                 random_start=True, norm='linfty', loss='crossentropy', device=None, attack='SMER', **kwargs):
        super().__init__(attack, model_name, epsilon, targeted, random_start, norm, loss, device)
        self._schedule(alpha, epoch, decay)
        self.num_model = len(model_name)
        self.m_smer = self.num_model * 4
        self.weight_selection = Weight_Selection(self.num_model).to(self.device)
        self.optimizer = torch.optim.SGD(self.weight_selection.parameters(), lr=2e-2, weight_decay=2e-3)

    def _draw_order(self):
        """m_smer/K shuffled permutations of the members, concatenated (smer.py:73-77)"""
        rounds = []
        for _ in range(self.m_smer // self.num_model):
            order = list(range(self.num_model))
            np.random.shuffle(order)
            rounds.append(order)
        return np.reshape(rounds, -1)

    def forward(self, data, label, **kwargs):
        data, label = self._to_device(data, label)
        members = self.model.models
        weight = self.weight_selection.weight
        delta = self.init_delta(data)
        momentum = 0.
        for _ in range(self.epoch):
            inner_delta = delta.detach()
            inner_momentum = torch.zeros_like(delta)
            for k in self._draw_order():
                inner_delta = inner_delta.detach().requires_grad_(True)
                logits = [self.weight_selection(member(data + inner_delta), m) for m, member in enumerate(members)]
                image_grad = torch.autograd.grad(F.cross_entropy(logits[int(k)], label), inner_delta, retain_graph=True)[0]
                group = 0
                for out in logits:
                    group = group + out / self.num_model
                weight.grad = torch.autograd.grad(-torch.log(F.cross_entropy(group, label)), weight)[0]
                self.optimizer.step()
                self.optimizer.zero_grad()
                inner_momentum = self.get_momentum(image_grad, inner_momentum)
                inner_delta = self.update_delta(inner_delta, data, inner_momentum, self.alpha)
            momentum = self.get_momentum(inner_momentum.clone(), momentum)
            delta = self.update_delta(delta, data, momentum, self.alpha)
        return delta.detach()
