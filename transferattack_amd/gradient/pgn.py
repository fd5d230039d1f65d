"""PGN (Ge et al., NeurIPS 2023) -- penalising the gradient norm: per iteration ``num_neighbor`` samples
x' = x + delta + U(-zeta, zeta); for each, g1 at x' and g2 at x' - alpha * g1/mean|g1|, averaged as
(1-gamma)*g1 + gamma*g2.  Mirror of transferattack/gradient/pgn.py:31-108.  HIP: neighbour sampling
(``ta_vmi_neighbor``), L1 normalisation, look-ahead axpy, fused update."""
from ..attack import Attack
from ..transforms import LookAhead, Neighbor


class PGN(Attack):
    """Official arguments: epsilon=16/255, alpha=epsilon/epoch, beta=3.0, gamma=0.5, num_neighbor=20, epoch=10, decay=1."""

    def __init__(self, model_name, epsilon=16/255, alpha=1.6/255, beta=3.0, gamma=0.5, num_neighbor=20, epoch=10,
                 decay=1., targeted=False, random_start=False, norm='linfty', loss='crossentropy', device=None,
                 attack='PGN', **kwargs):
        super().__init__(attack, model_name, epsilon, targeted, random_start, norm, loss, device)
        self.alpha = epsilon / epoch                     # the reference ignores its alpha argument (pgn.py:35)
        self.zeta = beta * epsilon
        self.gamma, self.epoch, self.decay, self.num_neighbor = gamma, epoch, decay, num_neighbor

    def get_averaged_gradient(self, data, delta, label, **kwargs):
        total = 0
        for _ in range(self.num_neighbor):
            x_near = self.transform(Neighbor.apply(delta, data, self.zeta, self.rng_seed, self._next_offset(),
                                                   self._uniform_like(data, self.zeta)))
            g_1 = self.get_grad(self.get_loss(self.get_logits(x_near), label), delta)
            x_next = self.transform(LookAhead.apply(x_near, self.l1_normalize(g_1), -self.alpha))
            g_2 = self.get_grad(self.get_loss(self.get_logits(x_next), label), delta)
            total = total + ((1 - self.gamma) * g_1 + self.gamma * g_2)
        return total / self.num_neighbor

    def forward(self, data, label, **kwargs):
        data, label = self._to_device(data, label)
        delta = self.init_delta(data)
        momentum = 0
        fused = self._can_fuse_update()
        for _ in range(self.epoch):
            direction = self.get_averaged_gradient(data, delta, label).contiguous()
            if fused:
                momentum = self._fused_update(direction, momentum, delta, data)
            else:
                momentum = self.get_momentum(direction, momentum)
                delta = self.update_delta(delta, data, momentum, self.alpha)
        return delta.detach()
