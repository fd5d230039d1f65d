"""NI-FGSM (Lin et al., ICLR 2020) -- Nesterov look-ahead x + alpha*decay*momentum before the surrogate.
Mirror of transferattack/gradient/nifgsm.py:31-39; the look-ahead is one HIP axpy (``ta_axpy``)."""
import torch

from .mifgsm import MIFGSM
from ..transforms import LookAhead


class NIFGSM(MIFGSM):
    """Official arguments: epsilon=16/255, alpha=epsilon/epoch=1.6/255, epoch=10, decay=1."""

    def __init__(self, model_name, epsilon=16/255, alpha=1.6/255, epoch=10, decay=1., targeted=False,
                 random_start=False, norm='linfty', loss='crossentropy', device=None, attack='NI-FGSM', **kwargs):
        super().__init__(model_name, epsilon, alpha, epoch, decay, targeted, random_start, norm, loss, device, attack)

    def transform(self, x, momentum, **kwargs):
        if not isinstance(momentum, torch.Tensor):      # first iteration: x + alpha*decay*0
            return x
        return LookAhead.apply(x, momentum, self.alpha * self.decay)
