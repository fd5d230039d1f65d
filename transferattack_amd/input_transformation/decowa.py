"""DeCoWA (Lin et al., AAAI 2024) -- deformation-constrained warping: each of ``num_warping`` views per iteration warps
the adversarial image with a thin-plate spline whose interior control points are displaced by a noise map; the map is
first drawn at random and then moved one gradient step (``rho``) in the direction that LOWERS the attack's loss (the
hardest warp), and the MI-FGSM gradient is averaged over the views.
Mirror of transferattack/input_transformation/decowa.py:35-184.

Everything that does not depend on the noise map is built once per image size and kept on the device -- the control
points, the spline system matrix L, and the 50 176 x k radial-basis matrix of the sampling grid (the reference rebuilds
all three, including the logarithms, on every one of its 400 warps per batch); per warp what remains is one small solve,
two thin products and ``grid_sampler_2d``.  Accumulation and the update are the HIP kernels of the MI-FGSM path.
The arithmetic of the spline is the reference's, operation for operation, so the host tier reproduces its bytes."""
import torch

from .. import _hip
from ..gradient.mifgsm import MIFGSM


def radial_basis(a, b):
    """U(r) = r^2 log(r^2 + 1e-9) between two point sets [n, p, 2] and [n, q, 2]"""
    d2 = torch.pow(a[:, :, None, :] - b[:, None, :, :], 2).sum(-1)
    return d2 * torch.log(d2 + 1e-9)


def affine_basis(points):
    n, k = points.shape[:2]
    basis = torch.ones(n, k, 3, device=points.device)
    basis[:, :, 1:] = points
    return basis


def grid_points_2d(width, height, device):
    rows, cols = torch.meshgrid([torch.linspace(-1.0, 1.0, height, device=device),
                                 torch.linspace(-1.0, 1.0, width, device=device)], indexing='ij')
    return torch.stack([cols, rows], dim=-1).contiguous().view(-1, 2)


class DeCowA(MIFGSM):
    """Official arguments: epsilon=16/255, alpha=1.6/255, epoch=10, decay=1. (mesh 3 x 3, rho=0.01, num_warping=20,
    noise_scale=2)."""

    def __init__(self, model_name, epsilon=16/255, alpha=1.6/255, epoch=10, decay=1., mesh_width=3, mesh_height=3, rho=0.01,
                 num_warping=20, noise_scale=2, targeted=False, random_start=False, norm='linfty', loss='crossentropy',
                 device=None, attack='DeCowA', **kwargs):
        super().__init__(model_name, epsilon, alpha, epoch, decay, targeted, random_start, norm, loss, device, attack)
        self.num_warping, self.noise_scale = num_warping, noise_scale
        self.mesh_width, self.mesh_height, self.rho = mesh_width, mesh_height, rho
        self._spline = {}

    def _constants(self, first, second):
        """control points [k, 2], system matrix L [1, k+3, k+3], and the sampling grid's radial / affine bases for images
        whose last two sizes are (first, second) -- the reference samples a (second x first) lattice (decowa.py:45-47)"""
        key = (first, second, str(self.device))
        if key not in self._spline:
            dev = self.device
            ctrl = grid_points_2d(self.mesh_width, self.mesh_height, dev)
            k = ctrl.shape[0]
            pts = ctrl[None, ...]
            system = torch.zeros(1, k + 3, k + 3, device=dev)
            basis = affine_basis(pts)
            system[:, :k, :k] = radial_basis(pts, pts)
            system[:, :k, k:] = basis
            system[:, k:, :k] = basis.permute(0, 2, 1)
            h, w = second, first
            lattice = torch.ones(1, h, w, 2, device=dev)
            lattice[:, :, :, 0] = torch.linspace(-1, 1, w)
            lattice[:, :, :, 1] = torch.linspace(-1, 1, h)[..., None]
            lattice = lattice.view(-1, h * w, 2)
            self._spline[key] = (ctrl, system, radial_basis(lattice, pts), affine_basis(lattice), (h, w))
        return self._spline[key]

    def vwt(self, x, noise_map):
        ctrl, system, lattice_radial, lattice_affine, (h, w) = self._constants(x.shape[2], x.shape[3])
        k = ctrl.shape[0]
        shift = torch.zeros([self.mesh_height, self.mesh_width, 2], device=self.device)
        shift[1:self.mesh_height - 1, 1:self.mesh_width - 1, :] = noise_map           # edge points stay put
        target = torch.zeros(1, k + 3, 2, device=self.device)
        target[:, :k, :] = (ctrl + shift.reshape(-1, 2))[None, ...]
        coeff = torch.linalg.solve(system, target)
        grid = (lattice_affine @ coeff[:, k:] + lattice_radial @ coeff[:, :k]).view(-1, h, w, 2)
        return torch.grid_sampler_2d(x, grid.repeat(x.shape[0], 1, 1, 1), 0, 0, False)

    def update_noise_map(self, x, label):
        x.requires_grad = False
        noise_map = (torch.rand([self.mesh_height - 2, self.mesh_width - 2, 2]) - 0.5) * self.noise_scale    # host draw
        noise_map.requires_grad = True
        loss = self.get_loss(self.get_logits(self.vwt(x, noise_map)), label)
        return (noise_map.detach() - self.rho * self.get_grad(loss, noise_map)).detach()

    def forward(self, data, label, **kwargs):
        data, label = self._to_device(data, label)
        delta = self.init_delta(data)
        momentum = 0
        fused = self._can_fuse_update()
        for _ in range(self.epoch):
            grads = torch.empty_like(data)
            for view in range(self.num_warping):
                hardest = self.update_noise_map((data + delta).clone().detach(), label)
                loss = self.get_loss(self.get_logits(self.vwt(data + delta, hardest)), label)
                _hip.grad_accumulate(grads, self.get_grad(loss, delta).contiguous(), first=(view == 0))
            grads = grads / self.num_warping
            if fused:
                momentum = self._fused_update(grads, momentum, delta, data)
            else:
                momentum = self.get_momentum(grads, momentum)
                delta = self.update_delta(delta, data, momentum, self.alpha)
        return delta.detach()
