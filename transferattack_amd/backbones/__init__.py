"""Surrogate backbones the hot path differentiates through.

The reference pulls these from torchvision / timm with downloaded weights
(transferattack/attack.py:48-60); neither package nor network exists here, so the engine carries
its own definitions with the upstream parameter names.  ``create(name)`` loads
``$TA_WEIGHTS_DIR/<name>.pth`` (a plain ``state_dict``) when present and otherwise initialises
deterministically from ``seed`` -- the attack arithmetic does not depend on which.
"""
import os

import torch
import torch.nn as nn

from .resnet import resnet18, resnet34, resnet50, resnet101
from .vgg import vgg16, vgg19
from .inception import inception_v3
from .mobilenet import mobilenet_v2
from .vit import vit_base_patch16_224, vit_tiny_patch16_224


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
}
LOCAL_ZOO = {"toy_cnn": toy_cnn}


def available():
    return sorted(list(TORCHVISION_ZOO) + list(TIMM_ZOO) + list(LOCAL_ZOO))


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
    elif verbose:
        print('=> Loading model {} ({}) with seeded random init (seed={}); set TA_WEIGHTS_DIR for '
              'pretrained weights'.format(name, origin, seed))
    return model.eval()
