"""TEST INFRASTRUCTURE -- import shim for the *real* reference (read-only, /root/reference).

Only used by ``oracle/gen_golden.py`` (golden-vector generation in the build container) and by
``tests/test_oracle_vs_reference.py`` (skipped when /root/reference is absent, e.g. on the GPU box).
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


def _install_stubs():
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tv_models = types.ModuleType("torchvision.models")
        tv_tf = types.ModuleType("torchvision.transforms")
        tv_tf.Resize = _Resize
        tv_tf.Normalize = _Normalize
        tv.models = tv_models
        tv.transforms = tv_tf
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.models"] = tv_models
        sys.modules["torchvision.transforms"] = tv_tf
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
