"""VA-I-FGSM (Xiong et al., 2020) -- virtual step + auxiliary labels: every iteration takes the gradient of the loss on
the true label and of MINUS the loss on ``aux_num`` random other labels, and applies them one after another as
un-clipped sign steps (image box only); the eps-ball is enforced once, at the end.
Mirror of transferattack/gradient/vaifgsm.py:31-126.  HIP: each step is ``ta_update_delta_linf`` with an infinite ball."""
import numpy as np
import torch

from ..attack import Attack
from .. import _hip


class VAIFGSM(Attack):
    """Official arguments: epsilon=16/255, alpha=0.007, epoch=20, aux_num=3."""

    def __init__(self, model_name, epsilon=16/255, alpha=0.007, epoch=20, aux_num=3, targeted=False, random_start=False,
                 norm='linfty', loss='crossentropy', device=None, attack='VA-I-FGSM', **kwargs):
        super().__init__(attack, model_name, epsilon, targeted, random_start, norm, loss, device)
        self.alpha, self.epoch = alpha, epoch
        self.aux_num = aux_num
        self.num_classes = 1000

    def get_aux_labels(self, label):
        """``aux_num`` label tensors [N]: per image the head of a host permutation of the classes without the true one
        (vaifgsm.py:40-66; one ``randperm`` per image on the CPU generator)"""
        picks = np.empty((label.shape[0], self.aux_num), dtype=np.int64)
        for i, truth in enumerate(label.tolist()):
            order = [c for c in torch.randperm(self.num_classes).tolist() if c != truth]
            picks[i] = order[:self.aux_num]
        return [torch.from_numpy(np.ascontiguousarray(picks[:, k])).to(self.device) for k in range(self.aux_num)]

    def update_delta(self, delta, data, grad, alpha, **kwargs):
        src = delta.detach().contiguous()
        out = torch.empty_like(src)
        if self.norm == 'linfty':
            _hip.update_delta_linf(src, data.contiguous(), grad.detach().contiguous(), alpha, float('inf'), out)
        else:
            # the reference's l2 branch calls renorm without its max norm (vaifgsm.py:73) and therefore raises; so does this
            grad_norm = torch.norm(grad.view(grad.size(0), -1), dim=1).view(-1, 1, 1, 1)
            out = (src + grad / (grad_norm + 1e-20) * alpha).view(src.size(0), -1).renorm(p=2, dim=0).view_as(src)
        return out.requires_grad_(True)

    def forward(self, data, label, **kwargs):
        data, label = self._to_device(data, label)
        delta = self.init_delta(data)
        for _ in range(self.epoch):
            logits = self.get_logits(self.transform(data + delta))
            losses = [self.get_loss(logits, label)]
            losses += [-self.get_loss(logits, aux) for aux in self.get_aux_labels(label)]
            grads = [torch.autograd.grad(one, delta, retain_graph=True, create_graph=False)[0] for one in losses]
            for grad in grads:
                delta = self.update_delta(delta, data, grad, self.alpha)
        return torch.clamp(delta, -self.epsilon, self.epsilon).detach()
