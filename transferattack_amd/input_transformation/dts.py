"""DTS-MI-FGSM: DIM + TIM + SIM combined (BASELINE.json configs[2]).

The reference has no such class (cooperative inheritance of its DIM/TIM/SIM does not work);
the composition follows its own precedents: transform = DIM.transform(SIM.transform(x)) -- one DIM geometry
per iteration shared by all scale copies (l2t.py:36-42 chaining; sasd_ws.py:118-133 DI-on-input + TI-on-grad),
get_grad = TIM.get_grad, get_loss = SIM.get_loss.
"""
from .tim import TIM
from ..transforms import DimResizePad, ScaleCopies, dim_draw


class DTS(TIM):
    def __init__(self, model_name, epsilon=16/255, alpha=1.6/255, epoch=10, decay=1., kernel_type='gaussian',
                 kernel_size=15, resize_rate=1.1, diversity_prob=0.5, num_scale=5, targeted=False,
                 random_start=False, norm='linfty', loss='crossentropy', device=None, attack='DTS-MI-FGSM', **kwargs):
        super().__init__(model_name, epsilon, alpha, epoch, decay, kernel_type, kernel_size, targeted, random_start,
                         norm, loss, device, attack)
        if resize_rate < 1:
            raise Exception("Error! The resize rate should be larger than 1.")
        self.resize_rate = resize_rate
        self.diversity_prob = diversity_prob
        self.num_scale = num_scale

    def transform(self, x, **kwargs):
        x = ScaleCopies.apply(x, self.num_scale)
        geom = dim_draw(x.shape[-1], self.resize_rate, self.diversity_prob)
        return x if geom is None else DimResizePad.apply(x, *geom)

    def get_loss(self, logits, label):
        label = label.repeat(self.num_scale)
        return -self.loss(logits, label) if self.targeted else self.loss(logits, label)
