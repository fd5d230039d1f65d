"""MEF (Qiu et al., 2024) -- maximin expected flatness: ``num_neighbor`` points per iteration, sampled uniformly around
delta and pushed along the previous iteration's (inverted, L1-normalised) gradients; the update uses the mean of the
gradients taken AT those points.  Mirror of transferattack/gradient/mef.py:35-128.  HIP: fused momentum + projected step
on the summed gradient; the sampling arithmetic is the method's own (elementwise torch ops)."""
import torch
import torch.nn as nn

from ..attack import Attack


class MEF(Attack):
    """Official arguments: epsilon=16/255, alpha=1.6/255, num_neighbor=20, gamma=2., kesai=0.15, epoch=20, inner_decay=0.9,
    decay=0.5."""

    def __init__(self, model_name, epsilon=16/255, alpha=1.6/255, num_neighbor=20, gamma=2., kesai=0.15, epoch=20,
                 inner_decay=0.9, decay=0.5, targeted=False, random_start=False, norm='linfty',
                 loss='crossentropy_no_reduction', device=None, attack='MEF', **kwargs):
        super().__init__(attack, model_name, epsilon, targeted, random_start, norm, loss, device)
        self._schedule(alpha, epoch, decay)
        self.kesai, self.gamma = kesai * epsilon, gamma * epsilon
        self.inner_decay, self.num_neighbor = inner_decay, num_neighbor

    def loss_function(self, loss):
        if loss == 'crossentropy':
            return nn.CrossEntropyLoss()
        if loss == 'crossentropy_no_reduction':
            return nn.CrossEntropyLoss(reduction='none')
        raise Exception("Unsupported loss {}".format(loss))

    def get_conditional_sampled_points(self, delta, grad_pgia):
        """the ``num_neighbor`` sampling points: uniform in the gamma-ball around delta, then pushed by kesai along the
        carried direction (mef.py:69-76); one draw covers all neighbours"""
        noise = self._uniform_like(grad_pgia, self.gamma)
        if noise is None:
            noise = torch.zeros_like(grad_pgia).uniform_(-self.gamma, self.gamma)
        return self.transform(self.transform(delta + noise) + self.kesai * grad_pgia)

    def get_points_gradient(self, data, delta, label, **kwargs):
        """gradient of the per-image mean loss taken AT each sampling point, scaled by 1 / num_neighbor (mef.py:78-89)"""
        stack = torch.zeros((self.num_neighbor,) + tuple(data.shape)).to(self.device)
        for i in range(self.num_neighbor):
            point = self.transform(data + delta[i])
            stack[i] = self.get_grad(self.get_loss(self.get_logits(point), label).mean(), point)
        return (1 / self.num_neighbor) * stack

    def forward(self, data, label, **kwargs):
        data, label = self._to_device(data, label)
        delta = self.init_delta(data)
        momentum = 0
        carried = torch.zeros((self.num_neighbor,) + tuple(data.shape)).to(self.device)      # the reference's grad_pgia
        fused = self._can_fuse_update()
        for _ in range(self.epoch):
            gradient = self.get_points_gradient(data, self.get_conditional_sampled_points(delta, carried), label)
            per_point_mean = torch.mean(torch.abs(gradient), (2, 3, 4), keepdim=True)
            carried = (gradient / per_point_mean).detach() - self.inner_decay * carried
            summed = gradient.sum(0)
            if fused:
                momentum = self._fused_update(summed, momentum, delta, data)
            else:
                momentum = self.get_momentum(summed, momentum)
                delta = self.update_delta(delta, data, momentum, self.alpha)
        return delta.detach()
