"""GNP (Wu et al., ICASSP 2023) -- gradient-norm penalty: a second gradient at x + r * g/mean|g| and the update
direction (1+beta)*g1 + beta*g2.  Mirror of transferattack/gradient/gnp.py:31-85 (note the reference's sign: it adds
beta*g2).  HIP: L1 normalisation (``ta_momentum`` with no history), look-ahead (``ta_axpy``), fused update."""
from ..attack import Attack
from ..transforms import LookAhead


class GNP(Attack):
    """Official arguments: epsilon=16/255, alpha=1.6/255, epoch=10, decay=1., r=0.01, beta=0.8."""

    def __init__(self, model_name, epsilon=16/255, alpha=1.6/255, epoch=10, decay=1., r=0.01, beta=0.8, targeted=False,
                 random_start=False, norm='linfty', loss='crossentropy', device=None, attack='GNP', **kwargs):
        super().__init__(attack, model_name, epsilon, targeted, random_start, norm, loss, device)
        self.alpha, self.epoch, self.decay, self.r, self.beta = alpha, epoch, decay, r, beta

    def forward(self, data, label, **kwargs):
        data, label = self._to_device(data, label)
        delta = self.init_delta(data)
        momentum = 0
        fused = self._can_fuse_update()
        for _ in range(self.epoch):
            g1 = self.get_grad(self.get_loss(self.get_logits(self.transform(data + delta)), label), delta)
            probe = LookAhead.apply(data + delta, self.l1_normalize(g1), self.r)
            g2 = self.get_grad(self.get_loss(self.get_logits(self.transform(probe)), label), delta)
            direction = (1 + self.beta) * g1 + self.beta * g2
            if fused:
                momentum = self._fused_update(direction, momentum, delta, data)
            else:
                momentum = self.get_momentum(direction, momentum)
                delta = self.update_delta(delta, data, momentum, self.alpha)
        return delta.detach()
