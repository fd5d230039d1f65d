"""DEM (Zou et al., ECCV 2020) -- diversity-ensemble: the logits of five resize-pad-resize views (rates 1.14 .. 1.66, one
geometry per rate and iteration, always applied) are averaged before the loss; the step is epsilon itself.
Mirror of transferattack/input_transformation/dem.py:41-117.  HIP: every view is ``ta_dim_fwd`` / ``ta_dim_bwd`` (the DIM
kernels; rates above 1.5 take the table-driven gather backward), fused momentum + projected step."""
import torch

from ..gradient.mifgsm import MIFGSM
from ..transforms import DimResizePad


def dem_draw(img_size, resize_rate):
    """One geometry (dem.py:54-66): randint(rnd) -> randint(top) -> randint(left) on the CPU generator; unlike DIM there is
    no ``diversity_prob`` draw"""
    img_resize = int(img_size * resize_rate)
    rnd = int(torch.randint(low=min(img_size, img_resize), high=max(img_size, img_resize), size=(1,), dtype=torch.int32))
    rem = img_resize - rnd
    top = int(torch.randint(low=0, high=rem, size=(1,), dtype=torch.int32))
    left = int(torch.randint(low=0, high=rem, size=(1,), dtype=torch.int32))
    return img_resize, rnd, top, left


class DEM(MIFGSM):
    """Official arguments: epsilon=16/255, alpha=epsilon, epoch=10, decay=1., resize_rates=[1.14, 1.27, 1.4, 1.53, 1.66]."""

    def __init__(self, model_name, epsilon=16/255, alpha=1.6/255, epoch=10, decay=1.,
                 resize_rates=[1.14, 1.27, 1.4, 1.53, 1.66], targeted=False, random_start=False, norm='linfty',
                 loss='crossentropy', device=None, attack='DEM', **kwargs):
        super().__init__(model_name, epsilon, alpha, epoch, decay, targeted, random_start, norm, loss, device, attack)
        if not isinstance(resize_rates, list):
            raise Exception("Error! The resize rates should be a list.")
        for resize_rate in resize_rates:
            if resize_rate < 1:
                raise Exception("Error! The resize rate should be larger than 1.")
        self.resize_rates = resize_rates
        self.alpha = epsilon

    def transform(self, x, resize_rate, **kwargs):
        return DimResizePad.apply(x, *dem_draw(x.shape[-1], resize_rate))

    def forward(self, data, label, **kwargs):
        data, label = self._to_device(data, label)
        delta = self.init_delta(data)
        momentum = 0
        fused = self._can_fuse_update()
        for _ in range(self.epoch):
            logits_ensemble = 0
            for resize_rate in self.resize_rates:
                logits_ensemble += self.get_logits(self.transform(data + delta, resize_rate, momentum=momentum))
            logits_ensemble /= len(self.resize_rates)
            grad = self.get_grad(self.get_loss(logits_ensemble, label), delta)
            if fused:
                momentum = self._fused_update(grad, momentum, delta, data)
            else:
                momentum = self.get_momentum(grad, momentum)
                delta = self.update_delta(delta, data, momentum, self.alpha)
        return delta.detach()
