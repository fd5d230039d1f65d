"""FGSM (Goodfellow et al., ICLR 2015) -- one step of size epsilon, no momentum.
Mirror of transferattack/gradient/fgsm.py:28-33."""
from ..attack import Attack


class FGSM(Attack):
    """Official arguments: epsilon=16/255.
    """

    def __init__(self, model_name, epsilon=16/255, targeted=False, random_start=False, norm='linfty',
                 loss='crossentropy', device=None, attack='FGSM', **kwargs):
        super().__init__(attack, model_name, epsilon, targeted, random_start, norm, loss, device)
        self._schedule(alpha=epsilon, epoch=1, decay=0)          # one full-budget step
