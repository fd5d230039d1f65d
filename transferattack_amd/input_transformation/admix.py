"""Admix (Wang et al., ICCV 2021) -- ``num_admix`` copies x + strength * x[randperm] (mixed-in image
detached), each at ``num_scale`` scales.  Mirror of transferattack/input_transformation/admix.py:32-51.
The permutations are drawn on the CPU generator (reference order) and uploaded; mixing + scaling, and the
backward sum over the 15 copies, are single HIP kernels (``ta_admix_fwd/bwd``)."""
import torch

from ..gradient.mifgsm import MIFGSM
from ..transforms import AdmixCopies


class Admix(MIFGSM):
    """Official arguments: epsilon=16/255, alpha=1.6/255, epoch=10, decay=1., num_scale=5, num_admix=3, admix_strength=0.2."""

    def __init__(self, model_name, epsilon=16/255, alpha=1.6/255, epoch=10, decay=1., num_scale=5, num_admix=3,
                 admix_strength=0.2, targeted=False, random_start=False, norm='linfty', loss='crossentropy',
                 device=None, attack='Admix', **kwargs):
        super().__init__(model_name, epsilon, alpha, epoch, decay, targeted, random_start, norm, loss, device, attack)
        self.num_scale = num_scale
        self.num_admix = num_admix
        self.admix_strength = admix_strength

    def transform(self, x, **kwargs):
        perm = torch.cat([torch.randperm(x.size(0)) for _ in range(self.num_admix)])      # CPU generator
        perm = perm.to(x.device, non_blocking=True)
        return AdmixCopies.apply(x, perm, self.num_admix, self.num_scale, self.admix_strength)

    def get_loss(self, logits, label):
        label = label.repeat(self.num_scale * self.num_admix)
        return -self.loss(logits, label) if self.targeted else self.loss(logits, label)
