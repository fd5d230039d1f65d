"""TIM (Dong et al., CVPR 2019) -- smooth the input gradient with a k x k kernel (depthwise, 'same').
Mirror of transferattack/input_transformation/tim.py:35-74.  The convolution is ``ta_depthwise_conv2d_same``:
LDS-tiled, same row-major FMA chain as the reference's CPU path (bit-identical smoothing)."""
import numpy as np
import torch

from ..gradient.mifgsm import MIFGSM
from .. import _hip


class TIM(MIFGSM):
    """Official arguments: epsilon=16/255, alpha=1.6/255, epoch=10, decay=1., kernel_type='gaussian', kernel_size=15."""

    def __init__(self, model_name, epsilon=16/255, alpha=1.6/255, epoch=10, decay=1., kernel_type='gaussian',
                 kernel_size=15, targeted=False, random_start=False, norm='linfty', loss='crossentropy',
                 device=None, attack='TIM', **kwargs):
        super().__init__(model_name, epsilon, alpha, epoch, decay, targeted, random_start, norm, loss, device, attack)
        self.kernel = self.generate_kernel(kernel_type, kernel_size)

    def generate_kernel(self, kernel_type, kernel_size, nsig=3):
        """[3, 1, k, k] fp32 kernel built in fp64 (tim.py:42-66): gaussian (normal pdf on linspace(-nsig, nsig)),
        uniform or linear, normalised to sum 1."""
        kind = kernel_type.lower()
        if kind == 'gaussian':
            x = np.linspace(-nsig, nsig, kernel_size)
            kern1d = np.exp(-x ** 2 / 2.0) / np.sqrt(2 * np.pi)          # == scipy.stats.norm.pdf(x)
            kernel = np.outer(kern1d, kern1d)
            kernel = kernel / kernel.sum()
        elif kind == 'uniform':
            kernel = np.ones((kernel_size, kernel_size)) / (kernel_size ** 2)
        elif kind == 'linear':
            kern1d = 1 - np.abs(np.linspace((-kernel_size + 1) // 2, (kernel_size - 1) // 2, kernel_size)
                                / (kernel_size ** 2))
            kernel = np.outer(kern1d, kern1d)
            kernel = kernel / kernel.sum()
        else:
            raise Exception("Unspported kernel type {}".format(kernel_type))
        stack = np.expand_dims(np.stack([kernel, kernel, kernel]), 1)
        return torch.from_numpy(stack.astype(np.float32)).to(self.device)

    def get_grad(self, loss, delta, **kwargs):
        grad = torch.autograd.grad(loss, delta, retain_graph=False, create_graph=False)[0].contiguous()
        out = torch.empty_like(grad)
        _hip.depthwise_conv2d_same(grad, out, self.kernel[0, 0].contiguous())
        return out
