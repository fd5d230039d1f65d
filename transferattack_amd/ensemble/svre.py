"""SVRE (Xiong et al., CVPR 2022) -- stochastic variance-reduced ensemble attack: per outer iteration, M = 4K inner
steps each pick ONE member k at random (numpy generator), and correct its gradient at the inner point by the
difference between its gradient and the ensemble gradient at the outer point.
Mirror of transferattack/ensemble/svre.py:31-95.  Rides on ``EnsembleModel.models[k]`` and the HIP hooks."""
import numpy as np

from ..attack import Attack


class SVRE(Attack):
    """Official arguments: epsilon=16/255, alpha=1.6/255, epoch=10, decay=1.0, random_start=True."""

    def __init__(self, model_name, epsilon=16/255, alpha=1.6/255, epoch=10, decay=1.0, targeted=False, random_start=True,
                 norm='linfty', loss='crossentropy', device=None, attack='SVRE', **kwargs):
        super().__init__(attack, model_name, epsilon, targeted, random_start, norm, loss, device)
        self.alpha, self.epoch, self.decay = alpha, epoch, decay
        self.K = len(model_name)
        self.M = 4 * self.K
        self.beta = alpha

    def get_logits_by_model_k(self, x, k):
        return self.model.models[k](x)

    def forward(self, data, label, **kwargs):
        data = data.clone().detach().to(self.device)
        label = label.clone().detach().to(self.device)
        momentum_G = 0.
        delta = self.init_delta(data).to(self.device)
        for _ in range(self.epoch):
            grad = self.get_grad(self.get_loss(self.get_logits(self.transform(data + delta)), label), delta)
            inner_G = 0.
            inner_delta = delta.clone().detach().requires_grad_(True)
            for _m in range(self.M):
                k = np.random.randint(self.K)
                inner_logits = self.get_logits_by_model_k(self.transform(data + inner_delta), k)
                inner_grad = self.get_grad(self.get_loss(inner_logits, label), inner_delta)
                outer_logits = self.get_logits_by_model_k(self.transform(data + delta), k)
                outer_grad = self.get_grad(self.get_loss(outer_logits, label), delta)
                inner_G = self.get_momentum(inner_grad - (outer_grad - grad), inner_G)
                inner_delta = self.update_delta(inner_delta, data, inner_G, self.beta)
            momentum_G = self.get_momentum(inner_G, momentum_G)
            delta = self.update_delta(delta, data, momentum_G, self.alpha)
        return delta.detach()
