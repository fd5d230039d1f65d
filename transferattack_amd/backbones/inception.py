"""Inception-v3 surrogate (Szegedy et al. 2015) with torchvision's module names so its checkpoint
loads.  The class is deliberately called ``Inception3``: the reference's ``wrap_model`` keys the
299-pixel resize and the 0.5/0.5 normalisation on ``'Inc' in model.__class__.__name__``
(transferattack/utils.py:49-53).  ``transform_input=True`` reproduces what torchvision forces for its
pretrained weights (the model re-normalises from ImageNet statistics to 0.5/0.5 internally); eval mode
returns logits only (the auxiliary head is kept for checkpoint compatibility, never evaluated).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class BasicConv2d(nn.Module):
    def __init__(self, cin, cout, k, stride=1, padding=0):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, stride=stride, padding=padding, bias=False)
        self.bn = nn.BatchNorm2d(cout, eps=0.001)

    def forward(self, x):
        return F.relu(self.bn(self.conv(x)), inplace=True)


def _seq(x, *mods):
    for m in mods:
        x = m(x)
    return x


class InceptionA(nn.Module):
    def __init__(self, cin, pool_features):
        super().__init__()
        self.branch1x1 = BasicConv2d(cin, 64, 1)
        self.branch5x5_1 = BasicConv2d(cin, 48, 1)
        self.branch5x5_2 = BasicConv2d(48, 64, 5, padding=2)
        self.branch3x3dbl_1 = BasicConv2d(cin, 64, 1)
        self.branch3x3dbl_2 = BasicConv2d(64, 96, 3, padding=1)
        self.branch3x3dbl_3 = BasicConv2d(96, 96, 3, padding=1)
        self.branch_pool = BasicConv2d(cin, pool_features, 1)

    def forward(self, x):
        return torch.cat([
            self.branch1x1(x),
            _seq(x, self.branch5x5_1, self.branch5x5_2),
            _seq(x, self.branch3x3dbl_1, self.branch3x3dbl_2, self.branch3x3dbl_3),
            self.branch_pool(F.avg_pool2d(x, 3, stride=1, padding=1))], 1)


class InceptionB(nn.Module):
    def __init__(self, cin):
        super().__init__()
        self.branch3x3 = BasicConv2d(cin, 384, 3, stride=2)
        self.branch3x3dbl_1 = BasicConv2d(cin, 64, 1)
        self.branch3x3dbl_2 = BasicConv2d(64, 96, 3, padding=1)
        self.branch3x3dbl_3 = BasicConv2d(96, 96, 3, stride=2)

    def forward(self, x):
        return torch.cat([
            self.branch3x3(x),
            _seq(x, self.branch3x3dbl_1, self.branch3x3dbl_2, self.branch3x3dbl_3),
            F.max_pool2d(x, 3, stride=2)], 1)


class InceptionC(nn.Module):
    def __init__(self, cin, c7):
        super().__init__()
        self.branch1x1 = BasicConv2d(cin, 192, 1)
        self.branch7x7_1 = BasicConv2d(cin, c7, 1)
        self.branch7x7_2 = BasicConv2d(c7, c7, (1, 7), padding=(0, 3))
        self.branch7x7_3 = BasicConv2d(c7, 192, (7, 1), padding=(3, 0))
        self.branch7x7dbl_1 = BasicConv2d(cin, c7, 1)
        self.branch7x7dbl_2 = BasicConv2d(c7, c7, (7, 1), padding=(3, 0))
        self.branch7x7dbl_3 = BasicConv2d(c7, c7, (1, 7), padding=(0, 3))
        self.branch7x7dbl_4 = BasicConv2d(c7, c7, (7, 1), padding=(3, 0))
        self.branch7x7dbl_5 = BasicConv2d(c7, 192, (1, 7), padding=(0, 3))
        self.branch_pool = BasicConv2d(cin, 192, 1)

    def forward(self, x):
        return torch.cat([
            self.branch1x1(x),
            _seq(x, self.branch7x7_1, self.branch7x7_2, self.branch7x7_3),
            _seq(x, self.branch7x7dbl_1, self.branch7x7dbl_2, self.branch7x7dbl_3,
                 self.branch7x7dbl_4, self.branch7x7dbl_5),
            self.branch_pool(F.avg_pool2d(x, 3, stride=1, padding=1))], 1)


class InceptionD(nn.Module):
    def __init__(self, cin):
        super().__init__()
        self.branch3x3_1 = BasicConv2d(cin, 192, 1)
        self.branch3x3_2 = BasicConv2d(192, 320, 3, stride=2)
        self.branch7x7x3_1 = BasicConv2d(cin, 192, 1)
        self.branch7x7x3_2 = BasicConv2d(192, 192, (1, 7), padding=(0, 3))
        self.branch7x7x3_3 = BasicConv2d(192, 192, (7, 1), padding=(3, 0))
        self.branch7x7x3_4 = BasicConv2d(192, 192, 3, stride=2)

    def forward(self, x):
        return torch.cat([
            _seq(x, self.branch3x3_1, self.branch3x3_2),
            _seq(x, self.branch7x7x3_1, self.branch7x7x3_2, self.branch7x7x3_3, self.branch7x7x3_4),
            F.max_pool2d(x, 3, stride=2)], 1)


class InceptionE(nn.Module):
    def __init__(self, cin):
        super().__init__()
        self.branch1x1 = BasicConv2d(cin, 320, 1)
        self.branch3x3_1 = BasicConv2d(cin, 384, 1)
        self.branch3x3_2a = BasicConv2d(384, 384, (1, 3), padding=(0, 1))
        self.branch3x3_2b = BasicConv2d(384, 384, (3, 1), padding=(1, 0))
        self.branch3x3dbl_1 = BasicConv2d(cin, 448, 1)
        self.branch3x3dbl_2 = BasicConv2d(448, 384, 3, padding=1)
        self.branch3x3dbl_3a = BasicConv2d(384, 384, (1, 3), padding=(0, 1))
        self.branch3x3dbl_3b = BasicConv2d(384, 384, (3, 1), padding=(1, 0))
        self.branch_pool = BasicConv2d(cin, 192, 1)

    def forward(self, x):
        a = self.branch3x3_1(x)
        b = self.branch3x3dbl_2(self.branch3x3dbl_1(x))
        return torch.cat([
            self.branch1x1(x),
            self.branch3x3_2a(a), self.branch3x3_2b(a),
            self.branch3x3dbl_3a(b), self.branch3x3dbl_3b(b),
            self.branch_pool(F.avg_pool2d(x, 3, stride=1, padding=1))], 1)


class InceptionAux(nn.Module):
    def __init__(self, cin, num_classes):
        super().__init__()
        self.conv0 = BasicConv2d(cin, 128, 1)
        self.conv1 = BasicConv2d(128, 768, 5)
        self.fc = nn.Linear(768, num_classes)

    def forward(self, x):
        x = self.conv1(self.conv0(F.avg_pool2d(x, 5, stride=3)))
        return self.fc(torch.flatten(F.adaptive_avg_pool2d(x, 1), 1))


class Inception3(nn.Module):
    def __init__(self, num_classes=1000, transform_input=True):
        super().__init__()
        self.transform_input = transform_input
        self.Conv2d_1a_3x3 = BasicConv2d(3, 32, 3, stride=2)
        self.Conv2d_2a_3x3 = BasicConv2d(32, 32, 3)
        self.Conv2d_2b_3x3 = BasicConv2d(32, 64, 3, padding=1)
        self.Conv2d_3b_1x1 = BasicConv2d(64, 80, 1)
        self.Conv2d_4a_3x3 = BasicConv2d(80, 192, 3)
        self.Mixed_5b = InceptionA(192, 32)
        self.Mixed_5c = InceptionA(256, 64)
        self.Mixed_5d = InceptionA(288, 64)
        self.Mixed_6a = InceptionB(288)
        self.Mixed_6b = InceptionC(768, 128)
        self.Mixed_6c = InceptionC(768, 160)
        self.Mixed_6d = InceptionC(768, 160)
        self.Mixed_6e = InceptionC(768, 192)
        self.AuxLogits = InceptionAux(768, num_classes)
        self.Mixed_7a = InceptionD(768)
        self.Mixed_7b = InceptionE(1280)
        self.Mixed_7c = InceptionE(2048)
        self.fc = nn.Linear(2048, num_classes)
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                nn.init.trunc_normal_(m.weight, std=0.1, a=-0.2, b=0.2)

    def forward(self, x):
        if self.transform_input:
            scale = x.new_tensor([0.229 / 0.5, 0.224 / 0.5, 0.225 / 0.5]).view(1, 3, 1, 1)
            shift = x.new_tensor([(0.485 - 0.5) / 0.5, (0.456 - 0.5) / 0.5, (0.406 - 0.5) / 0.5]).view(1, 3, 1, 1)
            x = x * scale + shift
        x = _seq(x, self.Conv2d_1a_3x3, self.Conv2d_2a_3x3, self.Conv2d_2b_3x3)
        x = F.max_pool2d(x, 3, stride=2)
        x = _seq(x, self.Conv2d_3b_1x1, self.Conv2d_4a_3x3)
        x = F.max_pool2d(x, 3, stride=2)
        x = _seq(x, self.Mixed_5b, self.Mixed_5c, self.Mixed_5d, self.Mixed_6a, self.Mixed_6b,
                 self.Mixed_6c, self.Mixed_6d, self.Mixed_6e, self.Mixed_7a, self.Mixed_7b, self.Mixed_7c)
        x = torch.flatten(F.adaptive_avg_pool2d(x, 1), 1)
        return self.fc(F.dropout(x, 0.5, self.training))


def inception_v3(**kw):
    return Inception3(**kw)
