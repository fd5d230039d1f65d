"""MI-FGSM (Dong et al., CVPR 2018) -- momentum on the L1-normalised gradient, decay 1.
Mirror of transferattack/gradient/mifgsm.py:31-36.  The whole per-iteration update of this class is the
fused HIP path of ``Attack.forward``."""
from ..attack import Attack


class MIFGSM(Attack):
    """Official arguments: epsilon=16/255, alpha=epsilon/epoch=1.6/255, epoch=10, decay=1.
    """

    def __init__(self, model_name, epsilon=16/255, alpha=1.6/255, epoch=10, decay=1., targeted=False,
                 random_start=False, norm='linfty', loss='crossentropy', device=None, attack='MI-FGSM', **kwargs):
        super().__init__(attack, model_name, epsilon, targeted, random_start, norm, loss, device)
        self._schedule(alpha, epoch, decay)
