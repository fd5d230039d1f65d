"""Surrogate backbones the hot path differentiates through.

The reference pulls these from torchvision / timm with downloaded weights
(transferattack/attack.py:48-60); neither package nor network exists here, so the engine carries
its own definitions with the upstream parameter names.  ``create(name)`` loads
``$TA_WEIGHTS_DIR/<name>.pth`` (a plain ``state_dict``); without ``TA_WEIGHTS_DIR`` it initialises deterministically
from ``seed`` (the attack arithmetic does not depend on which).  If ``TA_WEIGHTS_DIR`` is set but the file is missing it
raises instead of silently attacking / evaluating a random network (``TA_ALLOW_RANDOM_INIT=1`` overrides).
"""
import os

import torch
import torch.nn as nn

from .resnet import resnet18, resnet34, resnet50, resnet101
from .vgg import vgg16, vgg19
from .inception import inception_v3
from .mobilenet import mobilenet_v2
from .vit import vit_base_patch16_224, vit_tiny_patch16_224
from .timm_victims import pit_b_224, visformer_small, swin_tiny_patch4_window7_224


class ToyCNN(nn.Module):
    """Four-conv classifier for fast parity tests (not a reference model)."""

    def __init__(self, num_classes=10, width=16):
        super().__init__()
        self.body = nn.Sequential(
            nn.Conv2d(3, width, 3, padding=1), nn.ReLU(),
            nn.Conv2d(width, width, 3, stride=2, padding=1), nn.ReLU(),
            nn.Conv2d(width, 2 * width, 3, stride=2, padding=1), nn.ReLU(),
            nn.Conv2d(2 * width, 2 * width, 3, padding=1), nn.ReLU())
        self.fc = nn.Linear(2 * width, num_classes)

    def forward(self, x):
        return self.fc(self.body(x).mean(dim=(2, 3)))


def toy_cnn(**kw):
    return ToyCNN(**kw)


# torchvision names first, then timm names (attack.py:52-57 resolves in that order)
TORCHVISION_ZOO = {
    "resnet18": resnet18, "resnet34": resnet34, "resnet50": resnet50, "resnet101": resnet101,
    "vgg16": vgg16, "vgg19": vgg19, "inception_v3": inception_v3, "mobilenet_v2": mobilenet_v2,
}
TIMM_ZOO = {
    "vit_base_patch16_224": vit_base_patch16_224, "vit_tiny_patch16_224": vit_tiny_patch16_224,
    # victims of the evaluation row only (utils.py:16-17)
    "pit_b_224": pit_b_224, "visformer_small": visformer_small,
    "swin_tiny_patch4_window7_224": swin_tiny_patch4_window7_224,
}
LOCAL_ZOO = {"toy_cnn": toy_cnn}


def available():
    return sorted(list(TORCHVISION_ZOO) + list(TIMM_ZOO) + list(LOCAL_ZOO))


def calibrate_batchnorm(model, seed=0, size=224, batch=4):
    """Give a randomly initialised network the activation statistics of a trained one: one seeded forward in
    train mode with BatchNorm momentum 1 sets every running mean/var to the batch statistics, so eval-mode
    activations stay O(1) through the depth instead of growing ~2x per residual block (which saturates the
    soft-max, makes input-gradients vanish and would turn throughput / parity measurements meaningless).
    No-op for networks without BatchNorm; never applied when a checkpoint is loaded."""
    bns = [m for m in model.modules() if isinstance(m, nn.modules.batchnorm._BatchNorm)]
    if not bns:
        return model
    saved = [(m.momentum, m.training) for m in bns]
    gen = torch.Generator().manual_seed(seed + 12345)
    x = torch.rand(batch, 3, size, size, generator=gen)
    x = (x - 0.45) / 0.225
    was_training = model.training
    model.train()
    for m in bns:
        m.momentum = 1.0
    if "Inc" in model.__class__.__name__:      # 299-pixel network (utils.py:49-53 of the reference)
        x = torch.nn.functional.interpolate(x, size=(299, 299), mode="bilinear", align_corners=False)
    with torch.no_grad():
        model(x)
    for m, (mom, _) in zip(bns, saved):
        m.momentum = mom
    model.train(was_training)
    return model


def fold_batchnorm(model):
    """Fold every eval-mode BatchNorm2d that directly follows a Conv2d into that convolution
    (w' = w * gamma/sqrt(var+eps), b' = beta + (b - mean) * gamma/sqrt(var+eps)) and replace the BatchNorm by
    nn.Identity, keeping all module names (hooks on '1.layer1.1' etc. keep working).  Algebraically exact in
    eval mode; numerically it changes the surrogate's rounding (one multiply less per activation), which is why
    it is opt-in (TA_FOLD_BN=1) and reported: it removes two memory-bound passes over every activation map in
    the forward and two in the backward.  Pairs are recognised by name: (convK, bnK), (conv, bn), and
    Sequential[i] = Conv2d followed by Sequential[i+1] = BatchNorm2d."""
    def fuse(conv, bn):
        with torch.no_grad():
            scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
            conv.weight.mul_(scale.view(-1, 1, 1, 1))
            bias = conv.bias if conv.bias is not None else torch.zeros_like(bn.running_mean)
            new_bias = bn.bias + (bias - bn.running_mean) * scale
            conv.bias = nn.Parameter(new_bias.detach().clone(), requires_grad=False)

    folded = 0
    for parent in model.modules():
        names = dict(parent.named_children())
        pairs = []
        for cname, child in names.items():
            if isinstance(child, nn.Conv2d):
                bname = cname.replace("conv", "bn")
                if bname != cname and isinstance(names.get(bname), nn.BatchNorm2d):
                    pairs.append((cname, bname))
        if isinstance(parent, nn.Sequential):
            keys = list(names)
            for a, b in zip(keys, keys[1:]):
                if isinstance(names[a], nn.Conv2d) and isinstance(names[b], nn.BatchNorm2d):
                    pairs.append((a, b))
        for cname, bname in pairs:
            bn = getattr(parent, bname)
            if isinstance(bn, nn.BatchNorm2d) and not bn.training:
                fuse(getattr(parent, cname), bn)
                setattr(parent, bname, nn.Identity())
                folded += 1
    from . import fused
    fused.mark_folded(model)            # a ResNet whose every BatchNorm is folded may run with fused glue (backbones/fused.py)
    return folded


def create(name, seed=0, verbose=True, **kw):
    """Build backbone ``name``; raises ValueError('Model {} not supported') like attack.py:59."""
    for zoo, origin in ((TORCHVISION_ZOO, "torchvision-compatible"), (TIMM_ZOO, "timm-compatible"),
                        (LOCAL_ZOO, "local")):
        if name in zoo:
            ctor = zoo[name]
            break
    else:
        raise ValueError('Model {} not supported'.format(name))
    gen_state = torch.random.get_rng_state()
    torch.manual_seed(seed)
    try:
        model = ctor(**kw)
    finally:
        torch.random.set_rng_state(gen_state)
    wdir = os.environ.get("TA_WEIGHTS_DIR", "")
    path = os.path.join(wdir, name + ".pth") if wdir else ""
    if path and os.path.isfile(path):
        model.load_state_dict(torch.load(path, map_location="cpu"))
        if verbose:
            print('=> Loading model {} ({}) with weights {}'.format(name, origin, path))
    elif path and name not in LOCAL_ZOO and os.environ.get("TA_ALLOW_RANDOM_INIT", "0") != "1":
        raise FileNotFoundError("TA_WEIGHTS_DIR is set but {} does not exist: refusing to fall back to a randomly "
                                "initialised {} (set TA_ALLOW_RANDOM_INIT=1 to allow it)".format(path, name))
    else:
        calibrate_batchnorm(model, seed)
        if verbose:
            print('=> Loading model {} ({}) with seeded random init (seed={}); set TA_WEIGHTS_DIR for '
                  'pretrained weights'.format(name, origin, seed))
    return model.eval()
