"""MaskBlock (Fan et al., 2022) -- the batch is replicated once per ``patch_size`` block of the image, each copy with its
block blanked; the loss runs over all copies.  Mirror of transferattack/input_transformation/maskblock.py:34-57
(note its axis naming: the FIRST spatial axis is cut by the width list, which only matters for non-square inputs).
The copies are plain device ops (one masked write per copy, gradient = the complementary mask); momentum and the
projected step are the fused HIP update of ``Attack.forward``."""
import torch

from ..gradient.mifgsm import MIFGSM


class MaskBlock(MIFGSM):
    """Official arguments: epsilon=16/255, alpha=2/255, epoch=10, decay=1., patch_size=56."""

    def __init__(self, model_name, epsilon=16/255, alpha=2/255, epoch=10, decay=1., patch_size=56, targeted=False,
                 random_start=False, norm='linfty', loss='crossentropy', device=None, attack='MaskBlock', **kwargs):
        super().__init__(model_name, epsilon, alpha, epoch, decay, targeted, random_start, norm, loss, device, attack)
        self.patch_size = patch_size
        self.num = 0

    def transform(self, x, **kwargs):
        first, second = x.shape[2], x.shape[3]
        cuts_first = list(range(0, first + 1, self.patch_size))
        cuts_second = list(range(0, second + 1, self.patch_size))
        copies = []
        for lo_a, hi_a in zip(cuts_first[:-1], cuts_first[1:]):
            for lo_b, hi_b in zip(cuts_second[:-1], cuts_second[1:]):
                blanked = x.clone()
                blanked[:, :, lo_a:hi_a, lo_b:hi_b] = 0
                copies.append(blanked)
        self.num = len(copies)
        return torch.cat(copies, dim=0)

    def get_loss(self, logits, label):
        label = label.repeat(self.num)
        return -self.loss(logits, label) if self.targeted else self.loss(logits, label)
