"""SIA (Wang et al., ICCV 2023) -- structure invariant attack: ``num_scale`` copies of the batch, each cut into
``num_block`` x ``num_block`` rectangles at random positions, each rectangle put through one of seven simple operations
(roll rows / columns, flip rows / columns, rotate 180, random scaling, uniform noise + clip).
Mirror of transferattack/input_transformation/sia.py:35-106.  The draws stay on the host in the reference's order
(``transforms.sia_draw``); the whole stack and its backward are one HIP gather kernel each (``ta_sia_fwd/bwd``) instead
of ~10 ATen launches per rectangle.  (The reference also builds a 3x3 blur kernel, sia.py:69-80, that none of its
seven operations uses; it is not reproduced.)"""
import torch

from ..gradient.mifgsm import MIFGSM
from ..transforms import SiaBlocks, sia_draw


class SIA(MIFGSM):
    """Official arguments: epsilon=16/255, alpha=1.6/255, epoch=10, decay=1., num_scale=20, num_block=3."""

    def __init__(self, model_name, epsilon=16/255, alpha=1.6/255, epoch=10, decay=1., num_scale=20, num_block=3,
                 targeted=False, random_start=False, norm='linfty', loss='crossentropy', device=None, attack='SIA',
                 **kwargs):
        super().__init__(model_name, epsilon, alpha, epoch, decay, targeted, random_start, norm, loss, device, attack)
        self.num_scale, self.num_block = num_scale, num_block

    def transform(self, x, **kwargs):
        plan, noise = sia_draw(tuple(x.shape), self.num_block, self.num_scale, self.noise_source)
        plan = torch.from_numpy(plan).to(x.device)
        if noise is not None:
            noise = noise.to(x.device).contiguous()
        return SiaBlocks.apply(x, plan, self.num_scale, self.num_block, self.rng_seed, self._next_offset(), noise)

    def get_loss(self, logits, label):
        label = label.repeat(self.num_scale)
        return -self.loss(logits, label) if self.targeted else self.loss(logits, label)
