"""MobileNet-v2 surrogate (Sandler et al. 2018), torchvision parameter names
(``features.N.conv.M...``).  Member of the reference's default ensemble
(transferattack/ensemble/ens.py:27); source model: torchvision via attack.py:52-55."""
import torch
import torch.nn as nn


def _cna(cin, cout, k=3, stride=1, groups=1):
    return nn.Sequential(nn.Conv2d(cin, cout, k, stride, (k - 1) // 2, groups=groups, bias=False),
                         nn.BatchNorm2d(cout), nn.ReLU6(inplace=True))


class InvertedResidual(nn.Module):
    def __init__(self, cin, cout, stride, expand):
        super().__init__()
        hidden = int(round(cin * expand))
        self.residual = stride == 1 and cin == cout
        layers = []
        if expand != 1:
            layers.append(_cna(cin, hidden, k=1))
        layers += [_cna(hidden, hidden, stride=stride, groups=hidden),
                   nn.Conv2d(hidden, cout, 1, bias=False), nn.BatchNorm2d(cout)]
        self.conv = nn.Sequential(*layers)

    def forward(self, x):
        return x + self.conv(x) if self.residual else self.conv(x)


class MobileNetV2(nn.Module):
    def __init__(self, num_classes=1000):
        super().__init__()
        plan = ((1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1),
                (6, 160, 3, 2), (6, 320, 1, 1))
        feats, cin = [_cna(3, 32, stride=2)], 32
        for t, c, n, s in plan:
            for i in range(n):
                feats.append(InvertedResidual(cin, c, s if i == 0 else 1, t))
                cin = c
        feats.append(_cna(cin, 1280, k=1))
        self.features = nn.Sequential(*feats)
        self.classifier = nn.Sequential(nn.Dropout(0.2), nn.Linear(1280, num_classes))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out")
            elif isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, 0, 0.01)
                nn.init.zeros_(m.bias)

    def forward(self, x):
        x = nn.functional.adaptive_avg_pool2d(self.features(x), 1)
        return self.classifier(torch.flatten(x, 1))


def mobilenet_v2(**kw):
    return MobileNetV2(**kw)
