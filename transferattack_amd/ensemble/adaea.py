"""AdaEA (Chen et al., ICCV 2023) -- adaptive ensemble: per iteration every member's gradient is taken, the logits
are fused with softmax weights that score how well each member's one-step example transfers to the others (AGM),
and pixels where the members' gradient directions disagree are masked out of the fused gradient (DRF).
Mirror of transferattack/ensemble/adaea.py:36-148.  Rides on ``EnsembleModel.models[k]``; momentum and the
projected step are the HIP hooks, the scoring passes run under no_grad (they carry no gradient in the reference)."""
import torch
import torch.nn.functional as F

from ..attack import Attack


class AdaEA(Attack):
    """Official arguments: epsilon=16/255, alpha=1.6/255, epoch=10, decay=1.0, beta=10, threshold=-0.3."""

    def __init__(self, model_name, epsilon=16/255, alpha=1.6/255, epoch=10, decay=1.0, targeted=False,
                 random_start=True, beta=10, threshold=-0.3, norm='linfty', loss='crossentropy', device=None,
                 attack='AdaEA', **kwargs):
        super().__init__(attack, model_name, epsilon, targeted, random_start, norm, loss, device)
        self._schedule(alpha, epoch, decay)
        self.num_model = len(model_name)
        self.beta, self.threshold = beta, threshold

    def _start(self, data):
        """adaea.py:60: small Gaussian start, 0.001 * N(0, 1) (independent of ``random_start``)"""
        if self.noise_source is not None:
            noise = self.noise_source(data.shape, None, None).to(self.device)
        else:
            noise = torch.randn(data.shape, device=self.device)
        return (torch.zeros_like(data) + 0.001 * noise).requires_grad_(True)

    def forward(self, data, label, **kwargs):
        data = data.clone().detach().to(self.device)
        label = label.clone().detach().to(self.device)
        members = self.model.models
        momentum = 0.
        delta = self._start(data)
        side_by_side = hasattr(self.model, "member_input_grads")      # dist.ShardedMembers: one member per GPU
        for _ in range(self.epoch):
            if side_by_side:
                logits = list(self.model.member_logits(delta + data).unbind(0))       # one forward per member ...
                grads = self.model.member_input_grads(lambda z: F.cross_entropy(z, label))   # ... differentiated twice
            else:
                logits = [member(delta + data) for member in members]
                grads = [torch.autograd.grad(F.cross_entropy(out, label), delta, retain_graph=True)[0] for out in logits]
            weights = self.agm(ori_data=data, cur_adv=data + delta, grad=grads, label=label)
            keep = self.drf(grads, data_size=tuple(data.shape))
            keep = torch.where(keep >= self.threshold, torch.ones_like(keep), torch.zeros_like(keep))
            fused = (torch.stack(logits, dim=0) * weights.view(self.num_model, 1, 1)).sum(dim=0)
            grad = torch.autograd.grad(F.cross_entropy(fused, label), delta)[0] * keep
            momentum = self.get_momentum(grad, momentum)
            delta = self.update_delta(delta, data, momentum, self.alpha)
        return delta.detach()

    @torch.no_grad()
    def agm(self, ori_data, cur_adv, grad, label):
        """Adaptive gradient modulation (adaea.py:89-113): member j is scored by the loss its one-step example
        draws from every OTHER member, relative to that member's loss on its own example; softmax over members."""
        members = self.model.models
        probes = [self.get_adv_example(ori_data=ori_data, adv_data=cur_adv, grad=g) for g in grad]
        if hasattr(self.model, "member_losses"):                      # every GPU scores its member on all M probes
            cross = self.model.member_losses(probes, lambda z: F.cross_entropy(z, label))       # cross[i][j]
        else:
            cross = torch.stack([torch.stack([F.cross_entropy(members[i](probes[j]), label) for j in range(self.num_model)])
                                 for i in range(self.num_model)])
        score = torch.zeros(self.num_model, device=self.device)
        for j in range(self.num_model):
            for i in range(self.num_model):
                if i != j:
                    score[j] += cross[i][j] / cross[i][i] * self.beta
        return torch.softmax(score, dim=0)

    @torch.no_grad()
    def drf(self, grads, data_size):
        """Disparity-reduced filter (adaea.py:115-136): per-pixel cosine (over channels) between the members'
        channel-normalised gradients, averaged per member over its pairs, then over members.  As in the reference the
        last member contributes a zero row to that final mean."""
        n, height, width = data_size[0], data_size[-2], data_size[-1]
        unit = [F.normalize(g, dim=1) for g in grads]
        pair = torch.zeros(self.num_model, self.num_model, n, height, width, device=self.device)
        for i in range(self.num_model):
            for j in range(i + 1, self.num_model):
                pair[i][j] = F.cosine_similarity(unit[i], unit[j], dim=1, eps=1e-8)
        per_member = torch.zeros(self.num_model, n, height, width, device=self.device)
        for i in range(self.num_model - 1):
            per_member[i] = (pair[i, :].sum(dim=0) + pair[:, i].sum(dim=0)) / (self.num_model - 1)
        return per_member.mean(dim=0).view(n, 1, height, width)

    def get_adv_example(self, ori_data, adv_data, grad):
        """one signed step from ``adv_data`` projected on the eps-ball around ``ori_data`` and [0, 1] (adaea.py:138-148)"""
        stepped = adv_data.detach() + grad.sign() * self.alpha
        shift = torch.clamp(stepped - ori_data.detach(), -self.epsilon, self.epsilon)
        return torch.clamp(ori_data.detach() + shift, min=0.0, max=1.0)
