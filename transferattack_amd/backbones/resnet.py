"""ResNet surrogates (He et al. 2015), written for this engine.

The reference takes its surrogates from ``torchvision.models`` (transferattack/attack.py:52-55),
which is not part of the reference tree and is not installed here.  These definitions keep
torchvision's parameter names (``conv1``, ``layer2.0.downsample.1`` ...) so a standard torchvision
checkpoint loads with ``load_state_dict`` when one is supplied; the v1.5 stride placement (stride on
the 3x3 of a bottleneck) is the one torchvision ships.
"""
import torch
import torch.nn as nn


def _conv_bn(cin, cout, k, stride=1):
    return nn.Conv2d(cin, cout, k, stride=stride, padding=k // 2, bias=False), nn.BatchNorm2d(cout)


class _Shortcut(nn.Sequential):
    pass


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, cin, width, stride):
        super().__init__()
        self.conv1, self.bn1 = _conv_bn(cin, width, 3, stride)
        self.conv2, self.bn2 = _conv_bn(width, width, 3)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if stride != 1 or cin != width:
            self.downsample = _Shortcut(*_conv_bn(cin, width, 1, stride))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return self.relu(y + idt)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, cin, width, stride):
        super().__init__()
        cout = width * self.expansion
        self.conv1, self.bn1 = _conv_bn(cin, width, 1)
        self.conv2, self.bn2 = _conv_bn(width, width, 3, stride)
        self.conv3, self.bn3 = _conv_bn(width, cout, 1)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = _Shortcut(*_conv_bn(cin, cout, 1, stride))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return self.relu(y + idt)


class ResNet(nn.Module):
    def __init__(self, block, depths, num_classes=1000):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        cin = 64
        for i, (width, depth) in enumerate(zip((64, 128, 256, 512), depths)):
            blocks = []
            for j in range(depth):
                blocks.append(block(cin, width, stride=2 if (j == 0 and i > 0) else 1))
                cin = width * block.expansion
            setattr(self, "layer%d" % (i + 1), nn.Sequential(*blocks))
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(cin, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def forward(self, x):
        if torch.is_grad_enabled() and x.requires_grad:
            from . import fused                  # the attack path: same convolutions, fused glue (backbones/fused.py)
            if fused.usable(self, x):
                return fused.forward(self, x)
            # the module path (e.g. the reference-literal arrangement): only the stem's input gradient leaves MIOpen
            x = self.maxpool(self.relu(self.bn1(fused.module_path_stem(self, x))))
        else:
            x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))


def resnet18(**kw):
    return ResNet(BasicBlock, (2, 2, 2, 2), **kw)


def resnet34(**kw):
    return ResNet(BasicBlock, (3, 4, 6, 3), **kw)


def resnet50(**kw):
    return ResNet(Bottleneck, (3, 4, 6, 3), **kw)


def resnet101(**kw):
    return ResNet(Bottleneck, (3, 4, 23, 3), **kw)
