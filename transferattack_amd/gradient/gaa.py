"""GAA (gradient aggregation attack) -- per iteration N random points in a 3.5*eps box around x_adv; at each, the gradient
g' and the gradient g_hat one L1-normalised step rho further; aggregate g_hat + (1-lambda) g' + (1+lambda) g_hat, average,
L1-normalise (sum, +1e-8), momentum, sign step of eps/epoch.  Mirror of transferattack/gradient/gaa.py:32-158.  HIP: the
projected step (``ta_update_delta_linf``); the aggregation arithmetic is the method's own (elementwise torch ops)."""
import torch

from ..attack import Attack


class GAA(Attack):
    """Official arguments: epsilon=16/255, alpha=1.6/255, epoch=10, decay=1., rho=1.6/255, lambda_param=0.2, N=20."""

    def __init__(self, model_name, epsilon=16/255, alpha=1.6/255, epoch=10, decay=1., targeted=False, random_start=False,
                 norm='linfty', loss='crossentropy', device=None, attack='GAA', rho=1.6/255, lambda_param=0.2, xi=0.1,
                 N=20, **kwargs):
        super().__init__(attack, model_name, epsilon, targeted, random_start, norm, loss, device)
        self._schedule(alpha, epoch, decay)
        self.rho, self.lambda_param, self.N = rho, lambda_param, N
        self.xi = 3.5 * epsilon                                          # gaa.py:42 (the ``xi`` argument is ignored there)

    def sample_random_example(self, data, delta):
        if self.norm == 'linfty':
            if self.noise_source is not None:      # tests: the reference's CPU draws -- rand_like draws (and discards)
                self.noise_source(data.shape, 0.0, 1.0)                   # one tensor before uniform_ (gaa.py:112)
            random_pert = self._uniform_like(data, self.xi)
            if random_pert is None:
                random_pert = torch.empty_like(data).uniform_(-self.xi, self.xi)
        else:
            random_pert = torch.randn_like(data) * self.xi
            pert_norm = torch.norm(random_pert.view(random_pert.size(0), -1), p=2, dim=1).view(-1, 1, 1, 1)
            random_pert = random_pert / (pert_norm + 1e-8) * self.xi
        return torch.clamp(data + delta + random_pert, 0, 1)

    def calculate_gradient(self, x, label):
        x.requires_grad_(True)
        loss = self.get_loss(self.model(x), label)
        return torch.autograd.grad(loss, x, retain_graph=False, create_graph=False)[0]

    def forward(self, data, label, **kwargs):
        data, label = self._to_device(data, label)
        delta = self.init_delta(data)
        momentum = torch.zeros_like(delta).to(self.device)
        alpha = self.epsilon / self.epoch
        for _t in range(self.epoch):
            g_bar = torch.zeros_like(delta).to(self.device)
            for _i in range(self.N):
                x_prime = self.sample_random_example(data, delta).detach()
                g_prime = self.calculate_gradient(x_prime, label)
                g_prime_norm = torch.norm(g_prime, p=1, dim=(1, 2, 3), keepdim=True)
                x_hat = (x_prime + self.rho * (g_prime / (g_prime_norm + 1e-8))).detach()
                g_hat = self.calculate_gradient(x_hat, label)
                g_bar += g_hat + (1 - self.lambda_param) * g_prime + (1 + self.lambda_param) * g_hat
            g_bar = g_bar / self.N
            g_bar_norm = torch.norm(g_bar, p=1, dim=(1, 2, 3), keepdim=True)
            momentum = self.decay * momentum + (g_bar / (g_bar_norm + 1e-8))
            delta = self.update_delta(delta, data, momentum, alpha)
        return delta.detach()
