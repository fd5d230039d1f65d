"""CWA (Chen et al., ICLR 2024) -- common-weakness attack: a reverse step of size r (NEGATIVE alpha through
update_delta), then one sharpness-aware inner step per member with a per-image L2-normalised momentum; the net
inner displacement acts as the "gradient" of the outer batch-L1-normalised momentum.
Mirror of transferattack/ensemble/cwa.py:31-96."""
import torch

from ..attack import Attack
from ..utils import clamp, img_max, img_min


class CWA(Attack):
    """Official arguments: epsilon=16/255, alpha=3.2/255, epoch=10, decay=1.0, beta=50, r_size=16/255/15,
    inner_step_size=250, random_start=True."""

    def __init__(self, model_name, epsilon=16/255, alpha=3.2/255, epoch=10, decay=1.0, beta=50, r_size=16/255/15,
                 inner_step_size=250, targeted=False, random_start=True, norm='linfty', loss='crossentropy',
                 device=None, attack='CWA', **kwargs):
        super().__init__(attack, model_name, epsilon, targeted, random_start, norm, loss, device)
        self.alpha, self.epoch, self.decay, self.beta = alpha, epoch, decay, beta
        self.r_size, self.inner_step_size = r_size, inner_step_size
        self.K = len(model_name)

    def get_logits_by_model_k(self, x, k):
        return self.model.models[k](x)

    def forward(self, data, label, **kwargs):
        data = data.clone().detach().to(self.device)
        label = label.clone().detach().to(self.device)
        inner_momentum, outer_momentum = 0., 0.
        delta = self.init_delta(data).to(self.device)
        n = data.shape[0]
        for _ in range(self.epoch):
            original_delta = delta.clone().detach()
            grad = self.get_grad(self.get_loss(self.get_logits(self.transform(data + delta)), label), delta)
            inner_delta = self.update_delta(delta.clone().detach(), data, grad, -self.r_size)      # reverse step
            for k in range(self.K):
                inner_delta = inner_delta.detach().requires_grad_(True)
                logits_k = self.get_logits_by_model_k(self.transform(data + inner_delta), k)
                grad_k = self.get_grad(self.get_loss(logits_k, label), inner_delta)
                inner_delta = inner_delta.detach()
                norm_k = torch.norm(grad_k.reshape(n, -1), p=2, dim=1).view(n, 1, 1, 1)
                inner_momentum = self.decay * inner_momentum + grad_k / norm_k
                inner_delta = torch.clamp(inner_delta + self.inner_step_size * inner_momentum, -self.epsilon, self.epsilon)
                inner_delta = clamp(inner_delta, img_min - data, img_max - data)
            fake_grad = inner_delta - original_delta
            outer_momentum = outer_momentum * self.decay + fake_grad / torch.norm(fake_grad, p=1)
            delta = self.update_delta(delta, data, outer_momentum, self.alpha)
        return delta.detach()
