"""RAP (Qin et al., NeurIPS 2022) -- reverse adversarial perturbation: after a late start (``transpoint``), every
iteration first finds the perturbation ``n_rap`` (eps_n-ball around the CURRENT adversarial point) that makes that point
look least adversarial -- ``adv_steps`` sign steps down the attack's loss from a uniform random start -- and takes the
MI-FGSM gradient at ``x + delta + n_rap``.  Mirror of transferattack/gradient/rap.py:42-147.

HIP: the inner search is ``ta_init_delta_uniform`` + ``ta_update_delta_linf`` with the eps_n-ball and the image box
taken around x + delta; the outer step is the fused momentum / projection update.  As in the reference the first inner
gradient is taken with respect to a start that depends on x + delta only through the box clamp."""
import torch

from ..attack import Attack
from .. import _hip


class RAP(Attack):
    """Official arguments: epsilon=16/255, alpha=2/255, epoch=400, transpoint=100, epsilon_n=16/255, alpha_n=2/255,
    adv_steps=8."""

    def __init__(self, model_name, epsilon=16/255, alpha=2/255, epoch=400, transpoint=100, epsilon_n=16/255, alpha_n=2/255,
                 adv_steps=8, targeted=False, random_start=False, norm='linfty', loss='crossentropy', device=None,
                 attack='RAP', **kwargs):
        super().__init__(attack, model_name, epsilon, targeted, random_start, norm, loss, device)
        self._schedule(alpha, epoch, 1.)
        self.alpha_n, self.adv_steps, self.transpoint, self.epsilon_n = alpha_n, adv_steps, transpoint, epsilon_n

    def get_logit_loss(self, logits, label):
        real = logits.gather(1, label.unsqueeze(1)).squeeze(1)
        return real.mean() if self.targeted else (-1 * real).mean()

    def init_n_rap(self, data, random_start, **kwargs):
        """uniform start in the eps_n-ball, clamped to the image box around ``data`` (rap.py:66-79; linfty)"""
        start = torch.zeros_like(data)
        if random_start:
            if self.norm != 'linfty':
                raise Exception("Unsupported norm {} for the reverse perturbation".format(self.norm))   # the reference's l2 branch indexes dim 10
            noise = None
            if self.noise_source is not None:
                noise = self.noise_source(data.shape, -self.epsilon_n, self.epsilon_n).to(self.device).contiguous()
            _hip.init_delta_uniform(start, data.detach().contiguous(), self.epsilon_n, self.rng_seed, self._next_offset(),
                                    noise=noise)
        return start.requires_grad_(True)

    def update_n_rap(self, delta, data, grad, alpha, **kwargs):
        src = delta.detach().contiguous()
        out = torch.empty_like(src)
        if self.norm == 'linfty':
            _hip.update_delta_linf(src, data.detach().contiguous(), grad.detach().contiguous(), alpha, self.epsilon_n, out)
        else:
            _hip.update_delta_l2(src, data.detach().contiguous(), grad.detach().contiguous(), alpha, self.epsilon, out)
        return out.requires_grad_(True)

    def get_n_rap(self, data, label):
        data = data.detach()
        n_rap = self.init_n_rap(data, random_start=True)
        for _ in range(self.adv_steps):
            loss = -self.get_loss(self.get_logits(self.transform(data + n_rap)), label)
            n_rap = self.update_n_rap(n_rap, data, self.get_grad(loss, n_rap), self.alpha_n)
        return n_rap.detach()

    def forward(self, data, label, **kwargs):
        data, label = self._to_device(data, label)
        delta = self.init_delta(data)
        momentum = 0
        fused = self._can_fuse_update()
        n_rap = torch.zeros_like(data)
        for it in range(self.epoch):
            if it >= self.transpoint:                          # late start
                n_rap = self.get_n_rap(data + delta, label)
            logits = self.get_logits(self.transform(data + delta + n_rap, momentum=momentum))
            grad = self.get_grad(self.get_loss(logits, label), delta)
            if fused:
                momentum = self._fused_update(grad, momentum, delta, data)
            else:
                momentum = self.get_momentum(grad, momentum)
                delta = self.update_delta(delta, data, momentum, self.alpha)
        return delta.detach()
