"""TIM (Dong et al., CVPR 2019) -- smooth the input gradient with a k x k kernel (depthwise, 'same').
Mirror of transferattack/input_transformation/tim.py:35-74.  The convolution is ``ta_depthwise_conv2d_same``:
LDS-tiled, same row-major FMA chain as the reference's CPU path (bit-identical smoothing). The same kernel leaves the per-tile sums of |grad| for the
fused update (TIM.get_grad is the last kernel that writes the gradient), so the update reads the gradient once."""
import numpy as np
import torch

from ..gradient.mifgsm import MIFGSM
from .. import _hip


def _gaussian_profile(size, nsig):
    x = np.linspace(-nsig, nsig, size)
    return np.exp(-x ** 2 / 2.0) / np.sqrt(2 * np.pi)                     # == scipy.stats.norm.pdf(x)


def _linear_profile(size, nsig):
    ramp = np.linspace((-size + 1) // 2, (size - 1) // 2, size)
    return 1 - np.abs(ramp / (size ** 2))


_PROFILE_1D = {'gaussian': _gaussian_profile, 'linear': _linear_profile}


class TIM(MIFGSM):
    """Official arguments: epsilon=16/255, alpha=1.6/255, epoch=10, decay=1., kernel_type='gaussian', kernel_size=15."""

    def __init__(self, model_name, epsilon=16/255, alpha=1.6/255, epoch=10, decay=1., kernel_type='gaussian',
                 kernel_size=15, targeted=False, random_start=False, norm='linfty', loss='crossentropy',
                 device=None, attack='TIM', **kwargs):
        super().__init__(model_name, epsilon, alpha, epoch, decay, targeted, random_start, norm, loss, device, attack)
        self.kernel = self.generate_kernel(kernel_type, kernel_size)

    def generate_kernel(self, kernel_type, kernel_size, nsig=3):
        """[3, 1, k, k] fp32 smoothing kernel (tim.py:42-66).  'gaussian' and 'linear' are separable profiles: the
        outer product of a 1-D profile with itself, normalised to unit sum in fp64 before the cast; 'uniform' is the
        box filter."""
        kind = kernel_type.lower()
        if kind not in _PROFILE_1D and kind != 'uniform':
            raise Exception("Unspported kernel type {}".format(kernel_type))
        if kind == 'uniform':
            plane = np.ones((kernel_size, kernel_size)) / (kernel_size ** 2)
        else:
            profile = _PROFILE_1D[kind](kernel_size, nsig)
            plane = np.outer(profile, profile)
            plane = plane / plane.sum()
        planes = np.stack([plane] * 3)[:, None]                          # one identical kernel per colour plane
        return torch.from_numpy(planes.astype(np.float32)).to(self.device)

    def get_grad(self, loss, delta, **kwargs):
        grad = torch.autograd.grad(loss, delta, retain_graph=False, create_graph=False)[0].contiguous()
        out = torch.empty_like(grad)
        _hip.depthwise_conv2d_same(grad, out, self.kernel[0, 0].contiguous())
        return out
