"""TEST INFRASTRUCTURE -- import shim for the *real* reference (read-only, /root/reference).

Only used by ``oracle/gen_golden.py`` (golden-vector generation in the build container) and by the
``cpu_baseline`` leg of ``bench.py`` (kind "reference", only where /root/reference exists -- not on the GPU box).
Nothing in the product package may import this file.

The reference cannot be imported as shipped: ``transferattack/utils.py:3-4,9`` import
``torchvision.models``, ``torchvision.transforms`` and ``timm`` (none installed, no network) and
``transferattack/attack.py:60`` hard-codes ``.cuda()``.  Two shims, both through extension points
the reference itself offers:

1. stub modules ``torchvision{,.models,.transforms}`` / ``timm`` registered in ``sys.modules``
   (``Resize``/``Normalize`` restated as ``nn.Module`` s with torchvision's published semantics);
2. ``load_model`` override (documented override point, ``transferattack/attack.py:40-43``) that
   wraps a caller-supplied CPU backbone with the reference's own ``wrap_model``.
"""
import os
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

REFERENCE_ROOT = os.environ.get("TA_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "transferattack"))


class _Resize(nn.Module):
    """torchvision.transforms.Resize(int) on an NCHW tensor whose H == W: identity if already that
    size, else bilinear (align_corners=False, antialias off for tensors when upsampling)."""

    def __init__(self, size):
        super().__init__()
        self.size = size

    def forward(self, x):
        if x.shape[-1] == self.size and x.shape[-2] == self.size:
            return x
        return F.interpolate(x, size=(self.size, self.size), mode="bilinear", align_corners=False)


class _Normalize(nn.Module):
    def __init__(self, mean, std):
        super().__init__()
        self.mean = list(mean)
        self.std = list(std)

    def forward(self, x):
        mean = torch.as_tensor(self.mean, dtype=x.dtype, device=x.device).view(-1, 1, 1)
        std = torch.as_tensor(self.std, dtype=x.dtype, device=x.device).view(-1, 1, 1)
        return (x - mean) / std


class InterpolationMode:
    """torchvision.transforms.InterpolationMode: only the members the in-scope reference files name"""
    NEAREST = "nearest"
    BILINEAR = "bilinear"


def rotate_tensor(img, angle, mode="bilinear"):
    """torchvision.transforms.functional.rotate, tensor path of torchvision 0.13 -- restated in oracle/fgsm_oracle.py"""
    import fgsm_oracle
    return fgsm_oracle.rotate_tensor(img, angle, mode)


def functional_rotate(img, angle, interpolation=InterpolationMode.NEAREST, expand=False, center=None, fill=None):
    """torchvision.transforms.functional.rotate with the defaults input_transformation/ops.py:219 relies on (nearest
    neighbour, no expansion, image centre, zero fill)"""
    assert not expand and center is None and fill is None
    return rotate_tensor(img, angle, interpolation)


def _inverse_affine_matrix(center, angle, translate, scale, shear):
    """torchvision 0.13 ``functional._get_inverse_affine_matrix`` (inverted=True), python doubles"""
    import math
    rot, sx, sy = math.radians(angle), math.radians(shear[0]), math.radians(shear[1])
    cx, cy = center
    tx, ty = translate
    a = math.cos(rot - sy) / math.cos(sy)
    b = -math.cos(rot - sy) * math.tan(sx) / math.cos(sy) - math.sin(rot)
    c = math.sin(rot - sy) / math.cos(sy)
    d = -math.sin(rot - sy) * math.tan(sx) / math.cos(sy) + math.cos(rot)
    matrix = [d, -b, 0.0, -c, a, 0.0]
    matrix = [x / scale for x in matrix]
    matrix[2] += matrix[0] * (-cx - tx) + matrix[1] * (-cy - ty)
    matrix[5] += matrix[3] * (-cx - tx) + matrix[4] * (-cy - ty)
    matrix[2] += cx
    matrix[5] += cy
    return matrix


def functional_affine(img, angle, translate, scale, shear, interpolation=InterpolationMode.NEAREST, fill=None, center=None):
    """torchvision.transforms.functional.affine, tensor path of torchvision 0.13 (l2t.py:368): inverse matrix about the
    image centre, ``_gen_affine_grid`` (the construction of rotate_tensor) and grid_sample with zero padding"""
    assert fill is None and center is None
    shear = [float(shear), 0.0] if not isinstance(shear, (list, tuple)) else [float(s) for s in shear] + [0.0] * (2 - len(shear))
    matrix = _inverse_affine_matrix([0.0, 0.0], float(angle), [1.0 * t for t in translate], float(scale), shear)
    h, w = img.shape[-2], img.shape[-1]
    theta = torch.tensor(matrix, dtype=img.dtype).reshape(1, 2, 3)
    base = torch.empty(1, h, w, 3, dtype=img.dtype)
    base[..., 0].copy_(torch.linspace(-w * 0.5 + 0.5, w * 0.5 + 0.5 - 1, steps=w))
    base[..., 1].copy_(torch.linspace(-h * 0.5 + 0.5, h * 0.5 + 0.5 - 1, steps=h).unsqueeze_(-1))
    base[..., 2].fill_(1)
    rescaled = theta.transpose(1, 2) / torch.tensor([0.5 * w, 0.5 * h], dtype=img.dtype)
    grid = base.view(1, h * w, 3).bmm(rescaled).view(1, h, w, 2)
    return F.grid_sample(img, grid.expand(img.shape[0], h, w, 2), mode=interpolation, padding_mode="zeros", align_corners=False)


def functional_resized_crop(img, top, left, height, width, size, interpolation=InterpolationMode.BILINEAR):
    """torchvision.transforms.functional.resized_crop on a float tensor (0.13): slice, then
    ``interpolate(mode='bilinear', align_corners=False)`` without antialiasing"""
    assert top >= 0 and left >= 0 and top + height <= img.shape[-2] and left + width <= img.shape[-1]
    return F.interpolate(img[..., top:top + height, left:left + width], size=list(size), mode=interpolation, align_corners=False)


class RandomResizedCrop(nn.Module):
    """torchvision.transforms.RandomResizedCrop (0.13) for float NCHW tensors, as input_transformation/su.py:49 uses it:
    ``get_params`` -- up to ten (area, log-uniform aspect ratio) draws from torch's default generator, else a central
    crop -- then ``resized_crop`` (slice + bilinear interpolate, no antialiasing); one crop for the whole batch"""

    def __init__(self, size, scale=(0.08, 1.0), ratio=(3. / 4., 4. / 3.), interpolation=InterpolationMode.BILINEAR):
        super().__init__()
        self.size = (size, size) if isinstance(size, int) else tuple(size)
        self.scale, self.ratio, self.interpolation = scale, ratio, interpolation

    @staticmethod
    def get_params(img, scale, ratio):
        import math
        height, width = img.shape[-2], img.shape[-1]
        area = height * width
        log_ratio = torch.log(torch.tensor(ratio))
        for _ in range(10):
            target_area = area * torch.empty(1).uniform_(scale[0], scale[1]).item()
            aspect_ratio = torch.exp(torch.empty(1).uniform_(log_ratio[0], log_ratio[1])).item()
            w = int(round(math.sqrt(target_area * aspect_ratio)))
            h = int(round(math.sqrt(target_area / aspect_ratio)))
            if 0 < w <= width and 0 < h <= height:
                i = torch.randint(0, height - h + 1, size=(1,)).item()
                j = torch.randint(0, width - w + 1, size=(1,)).item()
                return i, j, h, w
        in_ratio = float(width) / float(height)
        if in_ratio < min(ratio):
            w = width
            h = int(round(w / min(ratio)))
        elif in_ratio > max(ratio):
            h = height
            w = int(round(h * max(ratio)))
        else:
            w, h = width, height
        return (height - h) // 2, (width - w) // 2, h, w

    def forward(self, img):
        i, j, h, w = self.get_params(img, self.scale, self.ratio)
        return functional_resized_crop(img, i, j, h, w, self.size, self.interpolation)


class RandomRotation(nn.Module):
    """torchvision.transforms.RandomRotation(degrees=(lo, hi), interpolation=...): one angle per call from torch's
    default generator (``torch.empty(1).uniform_(lo, hi)``), applied to the whole batch"""

    def __init__(self, degrees, interpolation=InterpolationMode.NEAREST, expand=False, center=None, fill=0):
        super().__init__()
        self.degrees = [float(d) for d in degrees]
        self.interpolation = interpolation
        assert not expand and center is None

    def forward(self, img):
        angle = float(torch.empty(1).uniform_(self.degrees[0], self.degrees[1]).item())
        return rotate_tensor(img, angle, self.interpolation)


def _install_stubs():
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tv_models = types.ModuleType("torchvision.models")
        tv_tf = types.ModuleType("torchvision.transforms")
        tv_tf.Resize = _Resize
        tv_tf.Normalize = _Normalize
        tv_tf.RandomRotation = RandomRotation
        tv_tf.RandomResizedCrop = RandomResizedCrop
        tv_tf.InterpolationMode = InterpolationMode
        tv_tff = types.ModuleType("torchvision.transforms.functional")
        tv_tff.rotate = functional_rotate
        tv_tff.affine = functional_affine
        tv_tff.resized_crop = functional_resized_crop
        tv_tff.InterpolationMode = InterpolationMode
        tv_tf.functional = tv_tff
        tv.models = tv_models
        tv.transforms = tv_tf
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.models"] = tv_models
        sys.modules["torchvision.transforms"] = tv_tf
        sys.modules["torchvision.transforms.functional"] = tv_tff
    if "timm" not in sys.modules:
        timm = types.ModuleType("timm")
        timm.list_models = lambda *a, **k: []
        sys.modules["timm"] = timm


def neutralise_cuda_calls():
    """A few reference classes hard-code ``.cuda()`` on freshly built tensors (e.g. gradient/pifgsm.py:52).  For
    golden generation on the CPU box make the call the identity; nothing else about the class is touched."""
    torch.Tensor.cuda = lambda self, *a, **k: self


def import_reference():
    """Return the reference's ``transferattack`` package (imported from REFERENCE_ROOT)."""
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    _install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import transferattack  # noqa: the reference package
    return transferattack


def make_reference_attack(name, backbone, **ctor_kwargs):
    """Instantiate the reference's own attack class ``name`` on CPU around ``backbone``.

    ``backbone`` is an ``nn.Module`` (or list of modules -> EnsembleModel); it is wrapped by the
    reference's ``wrap_model`` (utils.py:37-60) exactly as ``Attack.load_model`` would.
    """
    ta = import_reference()
    from transferattack.utils import wrap_model, EnsembleModel
    base = ta.load_attack_class(name)

    def load_model(self, model_name):
        if isinstance(backbone, (list, tuple)):
            return EnsembleModel([wrap_model(b.eval()) for b in backbone])
        return wrap_model(backbone.eval())

    cls = type("Ref_" + base.__name__, (base,), {"load_model": load_model})
    return cls(model_name="injected", **ctor_kwargs)
