"""Foolmix (Li et al., TIFS 2024) -- dual blending and direction update.  Per iteration:
  1. images whose true class has dropped out of the top k+1 logits are pulled back towards the decision boundary: delta
     moves against d = |f_y - mean top-k logit| / ||w||_1 * sign(w), w = grad f_y - grad(mean top-k logit), scaled so that
     its mean magnitude is gamma * alpha;
  2. pixel blending: ``n`` Gaussian pixel-blocks P_j (std 0.1) are added with strength zeta, at ``m`` scales 1 / 2^k;
  3. label blending: the gradient of the loss towards ``z`` random other classes on the same blended inputs (g_lens) is
     subtracted, beta-weighted, from every blended gradient;
  4. momentum on g / (||g||_1 + 1e-8) (the SUM of magnitudes, not the mean), sign step, projection.
Mirror of transferattack/gradient/foolmix.py:37-330: same draws in the same order (pixel blocks, then labels), the same
chunking of the blended batch (``grad_chunk_size`` images per surrogate pass, the loss averaged PER CHUNK as there), the
same order of accumulation.  What the reference builds image by image and block by block in Python lists is built here as
one broadcast expression per stage; the sign step and both projections are ``ta_update_delta_linf``.  The reference's
half-precision autocast (its ``use_amp``, active on CUDA) is not reproduced: this path computes in fp32 throughout."""
import time

import torch

from ..attack import Attack


class Foolmix(Attack):
    """Official arguments: epsilon=16/255, alpha=1.6/255, epoch=10, decay=1., m=5, n=3, z=1, k=5, zeta=0.2, beta=1.0,
    gamma=0.1."""

    def __init__(self, model_name, epsilon=16/255, alpha=1.6/255, epoch=10, decay=1., targeted=False, random_start=False,
                 norm='linfty', loss='crossentropy', device=None, attack='Foolmix', m=5, n=3, z=1, k=5, zeta=0.2, beta=1.0,
                 gamma=0.1, print_timing=True, use_amp=True, use_cache=False, grad_chunk_size=16, **kwargs):
        super().__init__(attack, model_name, epsilon, targeted, random_start, norm, loss, device)
        self._schedule(alpha, epoch, decay)
        self.m, self.n, self.z, self.k = m, n, z, k
        self.zeta, self.beta, self.gamma = zeta, beta, gamma
        self.print_timing, self.use_amp, self.use_cache, self.grad_chunk_size = print_timing, use_amp, use_cache, grad_chunk_size
        self.gradient_cache = {} if use_cache else None
        self.scaler = None                                   # fp32 only (see the module docstring)
        self._num_classes = None

    # ------------------------------------------------------------------------------------------- draws
    def _draw_device(self):
        """seeded parity runs (``noise_source`` set) draw from the host generator, as the reference does on the CPU"""
        return 'cpu' if self.noise_source is not None else self.device

    def generate_random_pixel_blocks(self, data):
        b, c, h, w = data.shape
        return (torch.randn(b, self.n, c, h, w, device=self._draw_device()) * 0.1).to(self.device)

    def generate_random_other_class_labels(self, data, P):
        if self._num_classes is None:                        # the reference asks the surrogate with a zero image every time
            with torch.no_grad():
                self._num_classes = self.model(torch.zeros(1, *data.shape[1:], device=self.device)).shape[1]
        return torch.randint(0, self._num_classes, (data.shape[0], self.n, self.z), device=self._draw_device()).to(self.device)

    # ------------------------------------------------------------------------------- gradients of blended inputs
    def calculate_gradient_batch(self, x, label):
        """input-gradient of the loss, ``grad_chunk_size`` images per surrogate pass; the loss is the mean over the CHUNK"""
        grads = []
        chunk = max(1, min(self.grad_chunk_size, x.size(0)))
        for lo in range(0, x.size(0), chunk):
            part = x[lo:lo + chunk].clone().detach().requires_grad_(True)
            loss = self.get_loss(self.model(part), label[lo:lo + chunk])
            grads.append(torch.autograd.grad(loss, part, retain_graph=False, create_graph=False)[0].detach())
        return torch.cat(grads, dim=0)

    def _blended_inputs(self, x, P, scales):
        """scale_k * (x_i + zeta * P_ij) for every image i, block j, scale k, flattened in that order"""
        mixed = x[:, None] + self.zeta * P                                   # [B, n, C, H, W]
        factors = torch.tensor([1.0 / (2 ** k) for k in range(scales)], device=x.device).view(1, 1, scales, 1, 1, 1)
        return (factors * mixed[:, :, None]).reshape(-1, *x.shape[1:])

    def calculate_integrated_gradient_batch(self, x, P, L):
        grads = self.calculate_gradient_batch(self._blended_inputs(x, P, self.z), L.reshape(-1))
        grads = grads.view(x.shape[0], self.n * self.z, *x.shape[1:])
        g_lens = torch.zeros_like(x)
        for step in range(self.n * self.z):
            g_lens = g_lens + grads[:, step]
        return g_lens / (self.n * self.z)

    def calculate_average_blended_gradient_batch(self, x, P, g_lens, label):
        labels = label.view(-1, 1).expand(-1, self.n * self.m).reshape(-1)
        grads = self.calculate_gradient_batch(self._blended_inputs(x, P, self.m), labels)
        grads = grads.view(x.shape[0], self.n * self.m, *x.shape[1:])
        g_mix = torch.zeros_like(x)
        for step in range(self.n * self.m):
            g_mix = g_mix + grads[:, step] - self.beta * g_lens
        return g_mix / (self.n * self.m)

    # --------------------------------------------------------------------------------------- direction update
    def _topk_mean_logit(self, logits, top_k_indices):
        if top_k_indices.dim() == 1:
            top_k_indices = top_k_indices.unsqueeze(0)
        return torch.mean(torch.gather(logits, 1, top_k_indices), dim=1, keepdim=True)

    def get_integrated_logits(self, x, top_k_indices):
        with torch.no_grad():
            return self._topk_mean_logit(self.model(x), top_k_indices)

    def get_class_gradient(self, x, label):
        point = x.clone().detach().requires_grad_(True)
        own = torch.gather(self.model(point), 1, label.unsqueeze(1))
        return torch.autograd.grad(own.sum(), point, retain_graph=False, create_graph=False)[0]

    def get_integrated_gradient(self, x, top_k_indices):
        point = x.clone().detach().requires_grad_(True)
        return torch.autograd.grad(self._topk_mean_logit(self.model(point), top_k_indices).sum(), point,
                                   retain_graph=False, create_graph=False)[0]

    def get_update_direction(self, f_topk, omega_y, omega_topk, label, x_adv):
        w = omega_y - omega_topk
        with torch.no_grad():
            f_y = torch.gather(self.model(x_adv), 1, label.unsqueeze(1))
        return (torch.abs(f_y - f_topk) / (torch.norm(w, p=1, dim=(1, 2, 3), keepdim=True) + 1e-8)) * torch.sign(w)

    def adjust_adversarial_example(self, delta, data, d_direction, alpha):
        scaling = alpha * torch.ones_like(d_direction) / (torch.mean(torch.abs(d_direction)) + 1e-8)
        return delta - self.gamma * d_direction * scaling

    def _pull_back_misclassified(self, data, delta, label, top_k_indices, alpha):
        missed = ~torch.any(top_k_indices == label.unsqueeze(1), dim=1)
        for i in missed.nonzero().flatten().tolist():
            point, own = data[i:i + 1] + delta[i:i + 1], label[i:i + 1]
            direction = self.get_update_direction(self.get_integrated_logits(point, top_k_indices[i]),
                                                  self.get_class_gradient(point, own),
                                                  self.get_integrated_gradient(point, top_k_indices[i]), own, point)
            moved = self.adjust_adversarial_example(delta[i:i + 1], data[i:i + 1], direction, alpha)
            delta = torch.cat([delta[:i], moved, delta[i + 1:]], dim=0)
        return delta

    # ------------------------------------------------------------------------------------------------ loop
    def forward(self, data, label, **kwargs):
        data, label = self._to_device(data, label)
        started = time.time()
        delta = self.init_delta(data)
        momentum = torch.zeros_like(delta).to(self.device)
        phases = {"top_k": 0.0, "pull_back": 0.0, "blend": 0.0, "update": 0.0}

        def lap(name, since):
            phases[name] += time.time() - since
            return time.time()

        for _ in range(self.epoch):
            mark = time.time()
            with torch.no_grad():
                top_k_indices = torch.topk(self.model(data + delta), self.k + 1, dim=1)[1]
            mark = lap("top_k", mark)
            if not self.targeted:
                delta = self._pull_back_misclassified(data, delta, label, top_k_indices, self.alpha)
            mark = lap("pull_back", mark)
            P = self.generate_random_pixel_blocks(data)
            L = self.generate_random_other_class_labels(data, P)
            g_lens = self.calculate_integrated_gradient_batch(data + delta, P, L)
            g_mix = self.calculate_average_blended_gradient_batch(data + delta, P, g_lens, label)
            mark = lap("blend", mark)
            momentum = self.decay * momentum + g_mix / (torch.norm(g_mix, p=1, dim=(1, 2, 3), keepdim=True) + 1e-8)
            delta = self.update_delta(delta, data, momentum, self.alpha)
            lap("update", mark)
        self.timing_stats = dict(phases, total_time=time.time() - started, iterations=self.epoch)
        if self.print_timing:
            self.print_timing_stats(self.timing_stats)
        return delta.detach()

    def print_timing_stats(self, timing_stats):
        print("Foolmix: %d iterations in %.3f s (" % (timing_stats["iterations"], timing_stats["total_time"])
              + ", ".join("%s %.3f s" % (k, timing_stats[k]) for k in ("top_k", "pull_back", "blend", "update")) + ")")

    def get_timing_stats(self):
        return getattr(self, "timing_stats", None)

    def clear_cache(self):
        if self.gradient_cache is not None:
            self.gradient_cache.clear()
