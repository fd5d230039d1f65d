"""BSR (Wang et al., CVPR 2024) -- block shuffle and rotation: ``num_scale`` copies of the batch, each cut into
``num_block`` strips along a random axis, the strips shuffled, every strip rotated by a random angle in [-24, 24] degrees
about its own centre, cut into ``num_block`` blocks along the other axis and shuffled again.
Mirror of transferattack/input_transformation/bsr.py:35-73.  The draws stay on the host in the reference's order, from
the reference's three generators (``transforms.bsr_draw``: python ``random``, numpy, torch); the whole stack and its
backward are one HIP gather kernel each (``ta_bsr_fwd/bwd``) instead of ~12 ATen launches per strip.  The rotation is
torchvision's ``RandomRotation`` (affine grid + bilinear ``grid_sample``, zero fill) -- a dependency the reference does
not vendor; its published algorithm is restated in oracle/fgsm_oracle.py::rotate_tensor."""
import torch

from ..gradient.mifgsm import MIFGSM
from ..transforms import BsrBlocks, bsr_draw


class BSR(MIFGSM):
    """Official arguments: epsilon=16/255, alpha=1.6/255, epoch=10, decay=1., num_scale=20 (the class default; the paper
    text says 10), num_block=3."""

    def __init__(self, model_name, epsilon=16/255, alpha=1.6/255, epoch=10, decay=1., num_scale=20, num_block=3,
                 targeted=False, random_start=False, norm='linfty', loss='crossentropy', device=None, attack='BSR',
                 **kwargs):
        super().__init__(model_name, epsilon, alpha, epoch, decay, targeted, random_start, norm, loss, device, attack)
        self.num_scale, self.num_block = num_scale, num_block

    def transform(self, x, **kwargs):
        plan = torch.from_numpy(bsr_draw(tuple(x.shape), self.num_block, self.num_scale)).to(x.device)
        return BsrBlocks.apply(x, plan, self.num_scale, self.num_block)

    def get_loss(self, logits, label):
        label = label.repeat(self.num_scale)
        return -self.loss(logits, label) if self.targeted else self.loss(logits, label)
