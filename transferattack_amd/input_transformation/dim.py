"""DIM (Xie et al., CVPR 2019) -- with probability ``diversity_prob`` resize the batch to a random side in
[size, size*rate), zero-pad to size*rate at a random offset and resize back.
Mirror of transferattack/input_transformation/dim.py:31-68; the three ATen ops (and their three backward
kernels) are ONE HIP gather kernel each way (``ta_dim_fwd`` / ``ta_dim_bwd``)."""
from ..gradient.mifgsm import MIFGSM
from ..transforms import DimResizePad, dim_draw


class DIM(MIFGSM):
    """Official arguments: epsilon=16/255, alpha=1.6/255, epoch=10, decay=1, resize_rate=1.1, diversity_prob=0.5."""

    def __init__(self, model_name, epsilon=16/255, alpha=1.6/255, epoch=10, decay=1., resize_rate=1.1,
                 diversity_prob=0.5, targeted=False, random_start=False, norm='linfty', loss='crossentropy',
                 device=None, attack='DIM', **kwargs):
        super().__init__(model_name, epsilon, alpha, epoch, decay, targeted, random_start, norm, loss, device, attack)
        if resize_rate < 1:
            raise Exception("Error! The resize rate should be larger than 1.")
        self.resize_rate = resize_rate
        self.diversity_prob = diversity_prob

    def transform(self, x, **kwargs):
        geom = dim_draw(x.shape[-1], self.resize_rate, self.diversity_prob)     # CPU generator, reference order
        if geom is None:
            return x
        return DimResizePad.apply(x, *geom)
