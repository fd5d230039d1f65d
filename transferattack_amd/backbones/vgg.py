"""VGG surrogates (Simonyan & Zisserman 2014), torchvision parameter names (``features.N``,
``classifier.N``) so standard checkpoints load.  Reference source of the model: torchvision via
transferattack/attack.py:52-55 (not vendored in the reference tree)."""
import torch
import torch.nn as nn

_PLANS = {
    "vgg16": (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"),
    "vgg19": (64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M",
              512, 512, 512, 512, "M"),
}


class VGG(nn.Module):
    def __init__(self, plan, num_classes=1000):
        super().__init__()
        layers, cin = [], 3
        for item in plan:
            if item == "M":
                layers.append(nn.MaxPool2d(2, 2))
            else:
                layers += [nn.Conv2d(cin, item, 3, padding=1), nn.ReLU(inplace=True)]
                cin = item
        self.features = nn.Sequential(*layers)
        self.avgpool = nn.AdaptiveAvgPool2d(7)
        self.classifier = nn.Sequential(
            nn.Linear(512 * 49, 4096), nn.ReLU(inplace=True), nn.Dropout(),
            nn.Linear(4096, 4096), nn.ReLU(inplace=True), nn.Dropout(),
            nn.Linear(4096, num_classes))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                nn.init.zeros_(m.bias)
            elif isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, 0, 0.01)
                nn.init.zeros_(m.bias)

    def forward(self, x):
        return self.classifier(torch.flatten(self.avgpool(self.features(x)), 1))


def vgg16(**kw):
    return VGG(_PLANS["vgg16"], **kw)


def vgg19(**kw):
    return VGG(_PLANS["vgg19"], **kw)
