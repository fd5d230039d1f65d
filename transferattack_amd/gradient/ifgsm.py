"""I-FGSM (Kurakin et al., ICLR-W 2017) -- K steps of size alpha, decay 0 (the normalised gradient still
goes through the momentum path).  Mirror of transferattack/gradient/ifgsm.py:30-35."""
from ..attack import Attack


class IFGSM(Attack):
    """Official arguments: epsilon=16/255, alpha=epsilon/epoch=1.6/255, epoch=10.
    """

    def __init__(self, model_name, epsilon=16/255, alpha=1.6/255, epoch=10, targeted=False, random_start=False,
                 norm='linfty', loss='crossentropy', device=None, attack='I-FGSM', **kwargs):
        super().__init__(attack, model_name, epsilon, targeted, random_start, norm, loss, device)
        self._schedule(alpha, epoch, decay=0)                    # no history: sign of the current gradient
