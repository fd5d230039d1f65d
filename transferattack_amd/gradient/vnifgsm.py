"""VNI-FGSM (Wang & He, CVPR 2021) -- VMI-FGSM with the Nesterov look-ahead.
Mirror of transferattack/gradient/vnifgsm.py:31-41."""
import torch

from .vmifgsm import VMIFGSM
from ..transforms import LookAhead


class VNIFGSM(VMIFGSM):
    """Official arguments: epsilon=16/255, alpha=1.6/255, beta=1.5, num_neighbor=20, epoch=10, decay=1."""

    def __init__(self, model_name, epsilon=16/255, alpha=1.6/255, beta=1.5, num_neighbor=20, epoch=10, decay=1.,
                 targeted=False, random_start=False, norm='linfty', loss='crossentropy', device=None,
                 attack='VNI-FGSM', **kwargs):
        super().__init__(model_name, epsilon, alpha, beta, num_neighbor, epoch, decay, targeted, random_start, norm,
                         loss, device, attack)

    def transform(self, x, momentum, **kwargs):
        if not isinstance(momentum, torch.Tensor):
            return x
        return LookAhead.apply(x, momentum, self.alpha * self.decay)
