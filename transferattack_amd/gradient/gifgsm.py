"""GI-FGSM (Wang et al., 2022) -- global momentum initialisation: ``pre_epoch`` look-ahead iterations with an
``s``-times larger step only warm up the momentum; the perturbation is then reset and the usual MI-FGSM loop runs
from that momentum.  Mirror of transferattack/gradient/gifgsm.py:31-81.  Both phases run on the fused update."""
from ..attack import Attack


class GIFGSM(Attack):
    """Official arguments: epsilon=16/255, alpha=1.6/255, epoch=10, decay=1., pre_epoch=5, s=10."""

    def __init__(self, model_name, epsilon=16/255, alpha=1.6/255, epoch=10, decay=1., targeted=False, random_start=False,
                 norm='linfty', loss='crossentropy', device=None, attack='GI-FGSM', pre_epoch=5, s=10, **kwargs):
        super().__init__(attack, model_name, epsilon, targeted, random_start, norm, loss, device, **kwargs)
        self.alpha, self.epoch, self.decay, self.pre_epoch, self.s = alpha, epoch, decay, pre_epoch, s

    def forward(self, data, label, **kwargs):
        data, label = self._to_device(data, label)
        momentum = 0.
        fused = self._can_fuse_update()
        for steps, step_size in ((self.pre_epoch, self.alpha * self.s), (self.epoch, self.alpha)):
            delta = self.init_delta(data).to(self.device)           # the second phase restarts from a fresh delta
            for _ in range(steps):
                logits = self.get_logits(self.transform(data + delta, momentum=momentum))
                grad = self.get_grad(self.get_loss(logits, label), delta)
                if fused:
                    momentum = self._fused_update(grad, momentum, delta, data, alpha=step_size)
                else:
                    momentum = self.get_momentum(grad, momentum)
                    delta = self.update_delta(delta, data, momentum, step_size)
        return delta.detach()
