"""SSM with the tricks of 'Bag of tricks to boost the adversarial transferability' (Bai et al., 2024).
Mirror of transferattack/input_transformation/ssm_with_tricks.py:17-470.

* SSM_H -- high-frequency perturbation: the spectrum mask of SSM is applied everywhere but the lowest 5 % x 5 % corner
  of the spectrum, which passes unchanged; gradient taken at the view, averaged over ``num_spectrum`` views.
* SSM_P -- block perturbation: ``num_scale`` views per iteration; in each, one randomly chosen operation (a random
  scale, a multiplicative uniform mask, or channel drop-out) is applied to the three spectrum blocks outside the lowest
  10 % x 10 % corner, each block with its own draw; the gradient goes back THROUGH the views to delta.

Every one of those spectrum edits is a pointwise multiplication, so a view is idct2(dct2(x + gauss) * M) with M assembled
from the draws: ONE ``spectrum.spectrum_view`` (two launches of the fp32-MFMA kernel ``ta_dct_pair``, forward and
backward) instead of the reference's FFT-path chains plus slice assignments.  Draw order is the reference's: Gaussian
(host), then the mask draws in block order."""
import numpy as np
import torch
import torch.nn.functional as F

from .. import spectrum
from ..gradient.mifgsm import MIFGSM
from .ssm import SSM


class SSM_H(SSM):
    """Official arguments: epsilon=16/255, alpha=1.6/255, epoch=10, decay=1, num_spectrum=20, rho=0.5."""

    low_ratio = 0.05

    def transform(self, x, **kwargs):
        gauss = (self._draw((x.size()[0], 3, 224, 224), True) * self.epsilon).to(self.device)
        mask = (self._draw(tuple(x.shape), False) * 2 * self.rho + 1 - self.rho).to(self.device)
        mask[:, :, :int(x.shape[2] * self.low_ratio), :int(x.shape[3] * self.low_ratio)] = 1     # low frequencies pass
        return spectrum.spectrum_view(x, gauss, mask)


class SSM_P(MIFGSM):
    """Official arguments: epsilon=16/255, alpha=1.6/255, epoch=10, decay=1, num_scale=20, rho=0.5."""

    def __init__(self, model_name, epsilon=16/255, alpha=1.6/255, epoch=10, decay=1., rho=0.5, num_scale=20,
                 targeted=False, random_start=False, norm='linfty', loss='crossentropy', device=None, **kwargs):
        super().__init__(model_name, epsilon, alpha, epoch, decay, targeted, random_start, norm, loss, device)
        self.num_scale, self.rho = num_scale, rho
        self.num_block = 3
        self.ops = [self.scale, self.add_noise, self.drop_out]

    # each operation returns the FACTOR it multiplies a spectrum block of the given shape with
    def _draw_device(self):
        """seeded parity runs (``noise_source`` set) take every draw from the host generator, as the reference does when
        it runs on the CPU; product mode draws masks on the device"""
        return 'cpu' if self.noise_source is not None else self.device

    def scale(self, shape):
        return torch.rand(2)[0].to(self.device).expand(shape)             # host draw in the reference too (:321-322)

    def add_noise(self, shape):
        # the reference draws with rand_like on a block of dct_2d's output, which is a TRANSPOSED view (:451): the host
        # generator fills it in memory order, i.e. column by column of the block
        n, c, first, second = shape
        draw = torch.rand((n, c, second, first), device=self._draw_device()).transpose(-1, -2)
        return (draw * 2 * self.rho + 1 - self.rho).to(self.device)

    def drop_out(self, shape):
        # F.dropout2d draws one Bernoulli(0.8) per (image, channel) and scales by 1 / 0.8, whatever the spatial size
        keep = F.dropout2d(torch.ones(shape[0], shape[1], 1, 1, device=self._draw_device()), p=0.2, training=True)
        return keep.to(self.device).expand(shape)

    def block_factors(self, shape, choice=-1):
        """the multiplier of the whole spectrum: ones in the low-frequency corner, the chosen operation's factor on the
        three other blocks, drawn block by block in the reference's order (ssm_with_tricks.py:342-354)"""
        n, c, first, second = shape
        pa, pb = int(first * 0.1), int(second * 0.1)
        chosen = choice if choice >= 0 else np.random.randint(0, high=len(self.ops), dtype=np.int32)
        factors = torch.ones(shape, device=self.device)
        factors[:, :, pa:, pb:] = self.ops[chosen]((n, c, first - pa, second - pb))
        factors[:, :, :pa, pb:] = self.ops[chosen]((n, c, pa, second - pb))
        factors[:, :, pa:, :pb] = self.ops[chosen]((n, c, first - pa, pb))
        return factors

    def dct_perturbation(self, x):
        gauss = torch.randn((x.size()[0], 3, 224, 224), device=self._draw_device())
        return spectrum.spectrum_view(x, (gauss * self.epsilon).to(self.device), self.block_factors(tuple(x.shape)))

    def transform(self, x, **kwargs):
        return torch.cat([self.dct_perturbation(x) for _ in range(self.num_scale)], dim=0)

    def get_loss(self, logits, label):
        return self.loss(logits, label.repeat(self.num_scale))          # no targeted negation in the reference (:469-470)
