"""SIM (Lin et al., ICLR 2020) -- ``num_scale`` copies x / 2^i stacked on the batch axis.
Mirror of transferattack/input_transformation/sim.py:29-46; forward (r 4, w 4*S B/elem) and backward
(sum_i g_i / 2^i) are single HIP kernels (``ta_scale_copies_fwd/bwd``)."""
from ..gradient.mifgsm import MIFGSM
from ..transforms import ScaleCopies


class SIM(MIFGSM):
    """Official arguments: epsilon=16/255, alpha=1.6/255, epoch=10, decay=1., num_scale=5."""

    def __init__(self, model_name, epsilon=16/255, alpha=1.6/255, epoch=10, decay=1., num_scale=5, targeted=False,
                 random_start=False, norm='linfty', loss='crossentropy', device=None, attack='SIM', **kwargs):
        super().__init__(model_name, epsilon, alpha, epoch, decay, targeted, random_start, norm, loss, device, attack)
        self.num_scale = num_scale

    def transform(self, x, **kwargs):
        return ScaleCopies.apply(x, self.num_scale)

    def get_loss(self, logits, label):
        label = label.repeat(self.num_scale)
        return -self.loss(logits, label) if self.targeted else self.loss(logits, label)
