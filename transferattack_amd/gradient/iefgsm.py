"""IE-FGSM (Peng et al., 2023) -- improved-Euler step: average of the normalised gradient at x and at the Euler
look-ahead x + alpha * g/mean|g|, accumulated with decay 1.  Mirror of transferattack/gradient/iefgsm.py:31-90."""
from ..attack import Attack
from ..transforms import LookAhead


class IEFGSM(Attack):
    """Official arguments: epsilon=16/255, alpha=1.6/255, epoch=10."""

    def __init__(self, model_name, epsilon=16/255, alpha=1.6/255, epoch=10, targeted=False, random_start=False,
                 norm='linfty', loss='crossentropy', device=None, **kwargs):
        super().__init__('IE-FGSM', model_name, epsilon, targeted, random_start, norm, loss, device)
        self.alpha, self.epoch, self.decay = alpha, epoch, 1.0

    def forward(self, data, label, **kwargs):
        data, label = self._to_device(data, label)
        delta = self.init_delta(data)
        momentum = 0
        for _ in range(self.epoch):
            grad = self.get_grad(self.get_loss(self.get_logits(self.transform(data + delta)), label), delta)
            g_p = self.l1_normalize(grad)
            ahead = LookAhead.apply(data + delta, g_p, self.alpha)
            grad = self.get_grad(self.get_loss(self.get_logits(self.transform(ahead)), label), delta)
            g_a = self.l1_normalize(grad)
            momentum = self.decay * momentum + (g_p + g_a) / 2            # no second normalisation (iefgsm.py:84)
            delta = self.update_delta(delta, data, momentum, self.alpha)
        return delta.detach()
