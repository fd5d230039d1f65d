"""The three remaining victim networks of the reference's evaluation row (transferattack/utils.py:16-17:
``pit_b_224``, ``visformer_small``, ``swin_tiny_patch4_window7_224``; built by ``timm.create_model(name, pretrained=True)``
at utils.py:29-34 and only ever run forward, in ``main.py --eval``).

timm is neither vendored in the reference nor installed here, so the architectures are restated from the published
definitions with timm's module / parameter names (the layout of timm 0.6.x, the version range the reference's README
names): a checkpoint saved by that timm loads with ``strict=True``.  What can be checked offline is checked in
tests/test_backbones.py: the parameter totals against the published model cards (Swin-T 28 288 354 exactly; PiT-B 73.76 M,
Visformer-S 40.22 M to the digits the cards print) and the key / shape rules of the three layouts.  Normalisation statistics: timm's ``default_cfg`` of all three is the ImageNet
default, read by ``wrap_model`` (utils.py:44-47)."""
import math

import torch
import torch.nn as nn

from .vit import Block as VitBlock, Mlp

IMAGENET_CFG = {"mean": (0.485, 0.456, 0.406), "std": (0.229, 0.224, 0.225)}


def _init_linear(module):
    for m in module.modules():
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.zeros_(m.bias)


# ------------------------------------------------------------------------------------------------ PiT
class ConvEmbedding(nn.Module):
    def __init__(self, cin, cout, patch, stride, padding):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, patch, stride=stride, padding=padding, bias=True)

    def forward(self, x):
        return self.conv(x)


class ConvHeadPooling(nn.Module):
    """depthwise 3x3 stride-2 convolution on the token grid, a Linear on the class token (pit.py)"""

    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, stride + 1, padding=stride // 2, stride=stride, groups=cin)
        self.fc = nn.Linear(cin, cout)

    def forward(self, x, cls_token):
        return self.conv(x), self.fc(cls_token)


class PitTransformer(nn.Module):
    def __init__(self, base_dim, depth, heads, ratio, pool=None):
        super().__init__()
        dim = base_dim * heads
        self.blocks = nn.Sequential(*[VitBlock(dim, heads, ratio) for _ in range(depth)])
        self.pool = pool

    def forward(self, x, cls_tokens):
        b, c, h, w = x.shape
        tokens = torch.cat([cls_tokens, x.flatten(2).transpose(1, 2)], dim=1)
        tokens = self.blocks(tokens)
        cls_tokens = tokens[:, :cls_tokens.shape[1]]
        x = tokens[:, cls_tokens.shape[1]:].transpose(1, 2).reshape(b, c, h, w)
        if self.pool is not None:
            x, cls_tokens = self.pool(x, cls_tokens)
        return x, cls_tokens


class PoolingVisionTransformer(nn.Module):
    def __init__(self, img=224, patch=14, stride=7, base_dims=(64, 64, 64), depth=(3, 6, 4), heads=(4, 8, 16), ratio=4.0,
                 num_classes=1000):
        super().__init__()
        self.default_cfg = dict(IMAGENET_CFG, input_size=(3, img, img))
        side = math.floor((img - patch) / stride + 1)
        self.pos_embed = nn.Parameter(torch.randn(1, base_dims[0] * heads[0], side, side) * 0.02)
        self.patch_embed = ConvEmbedding(3, base_dims[0] * heads[0], patch, stride, 0)
        self.cls_token = nn.Parameter(torch.randn(1, 1, base_dims[0] * heads[0]) * 0.02)
        stages = []
        for s in range(len(depth)):
            pool = None
            if s < len(depth) - 1:
                pool = ConvHeadPooling(base_dims[s] * heads[s], base_dims[s + 1] * heads[s + 1], stride=2)
            stages.append(PitTransformer(base_dims[s], depth[s], heads[s], ratio, pool))
        self.transformers = nn.Sequential(*stages)          # timm: SequentialTuple -- same child names
        self.norm = nn.LayerNorm(base_dims[-1] * heads[-1], eps=1e-6)
        self.head = nn.Linear(base_dims[-1] * heads[-1], num_classes)
        _init_linear(self)

    def forward(self, x):
        x = self.patch_embed(x) + self.pos_embed
        cls_tokens = self.cls_token.expand(x.shape[0], -1, -1)
        for stage in self.transformers:
            x, cls_tokens = stage(x, cls_tokens)
        return self.head(self.norm(cls_tokens)[:, 0])


def pit_b_224(**kw):
    return PoolingVisionTransformer(**kw)


# ------------------------------------------------------------------------------------------- Visformer
class SpatialMlp(nn.Module):
    def __init__(self, dim, hidden, group, spatial_conv):
        super().__init__()
        self.spatial_conv = spatial_conv
        if spatial_conv:
            hidden = dim * 5 // 6 if group < 2 else dim * 2
        self.conv1 = nn.Conv2d(dim, hidden, 1, bias=False)
        self.act1 = nn.GELU()
        if spatial_conv:
            self.conv2 = nn.Conv2d(hidden, hidden, 3, padding=1, groups=group, bias=False)
            self.act2 = nn.GELU()
        self.conv3 = nn.Conv2d(hidden, dim, 1, bias=False)

    def forward(self, x):
        x = self.act1(self.conv1(x))
        if self.spatial_conv:
            x = self.act2(self.conv2(x))
        return self.conv3(x)


class ConvAttention(nn.Module):
    def __init__(self, dim, heads, head_dim_ratio):
        super().__init__()
        self.heads = heads
        self.head_dim = round(dim // heads * head_dim_ratio)
        self.scale = self.head_dim ** -0.5
        self.qkv = nn.Conv2d(dim, self.head_dim * heads * 3, 1, bias=False)
        self.proj = nn.Conv2d(self.head_dim * heads, dim, 1, bias=False)

    def forward(self, x):
        b, c, h, w = x.shape
        q, k, v = self.qkv(x).reshape(b, 3, self.heads, self.head_dim, -1).permute(1, 0, 2, 4, 3)
        attn = ((q * self.scale) @ k.transpose(-2, -1)).softmax(dim=-1)
        return self.proj((attn @ v).permute(0, 1, 3, 2).reshape(b, -1, h, w))


class VisformerBlock(nn.Module):
    def __init__(self, dim, heads, head_dim_ratio, ratio, group, attn_disabled, spatial_conv):
        super().__init__()
        self.attn_disabled = attn_disabled
        if not attn_disabled:
            self.norm1 = nn.BatchNorm2d(dim)
            self.attn = ConvAttention(dim, heads, head_dim_ratio)
        self.norm2 = nn.BatchNorm2d(dim)
        self.mlp = SpatialMlp(dim, int(dim * ratio), group, spatial_conv)

    def forward(self, x):
        if not self.attn_disabled:
            x = x + self.attn(self.norm1(x))
        return x + self.mlp(self.norm2(x))


class ConvPatchEmbed(nn.Module):
    def __init__(self, cin, cout, patch):
        super().__init__()
        self.proj = nn.Conv2d(cin, cout, patch, stride=patch)
        self.norm = nn.BatchNorm2d(cout)

    def forward(self, x):
        return self.norm(self.proj(x))


class Visformer(nn.Module):
    def __init__(self, img=224, init_channels=32, dim=384, depth=(7, 4, 4), heads=6, ratio=4.0, group=8,
                 attn_stage="011", spatial_conv="100", num_classes=1000):
        super().__init__()
        self.default_cfg = dict(IMAGENET_CFG, input_size=(3, img, img))
        self.stem = nn.Sequential(nn.Conv2d(3, init_channels, 7, stride=2, padding=3, bias=False),
                                  nn.BatchNorm2d(init_channels), nn.ReLU(inplace=True))
        dims = (dim // 2, dim, dim * 2)
        sides = (img // 8, img // 16, img // 32)
        patches, cins = (4, 2, 2), (init_channels, dims[0], dims[1])
        ratios = (0.5, 1.0, 1.0)
        for s in range(3):
            setattr(self, "patch_embed%d" % (s + 1), ConvPatchEmbed(cins[s], dims[s], patches[s]))
            setattr(self, "pos_embed%d" % (s + 1), nn.Parameter(torch.randn(1, dims[s], sides[s], sides[s]) * 0.02))
            setattr(self, "stage%d" % (s + 1), nn.Sequential(*[
                VisformerBlock(dims[s], heads, ratios[s], ratio, group, attn_stage[s] == "0", spatial_conv[s] == "1")
                for _ in range(depth[s])]))
        self.norm = nn.BatchNorm2d(dims[2])
        self.head = nn.Linear(dims[2], num_classes)
        _init_linear(self)

    def forward(self, x):
        x = self.stem(x)
        for s in (1, 2, 3):
            x = getattr(self, "patch_embed%d" % s)(x) + getattr(self, "pos_embed%d" % s)
            x = getattr(self, "stage%d" % s)(x)
        return self.head(self.norm(x).mean(dim=(2, 3)))


def visformer_small(**kw):
    return Visformer(**kw)


# ------------------------------------------------------------------------------------------------ Swin
def _window_partition(x, ws):
    b, h, w, c = x.shape
    x = x.view(b, h // ws, ws, w // ws, ws, c)
    return x.permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, c)


def _window_reverse(windows, ws, h, w):
    b = windows.shape[0] // ((h // ws) * (w // ws))
    x = windows.view(b, h // ws, w // ws, ws, ws, -1)
    return x.permute(0, 1, 3, 2, 4, 5).reshape(b, h, w, -1)


class WindowAttention(nn.Module):
    def __init__(self, dim, ws, heads):
        super().__init__()
        self.heads, self.scale = heads, (dim // heads) ** -0.5
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) ** 2, heads))
        coords = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")).flatten(1)
        rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
        rel[:, :, 0] += ws - 1
        rel[:, :, 1] += ws - 1
        rel[:, :, 0] *= 2 * ws - 1
        self.register_buffer("relative_position_index", rel.sum(-1))
        self.qkv = nn.Linear(dim, dim * 3)
        self.proj = nn.Linear(dim, dim)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)

    def forward(self, x, mask=None):
        bw, n, c = x.shape
        q, k, v = self.qkv(x).reshape(bw, n, 3, self.heads, c // self.heads).permute(2, 0, 3, 1, 4)
        attn = (q * self.scale) @ k.transpose(-2, -1)
        bias = self.relative_position_bias_table[self.relative_position_index.view(-1)].view(n, n, -1)
        attn = attn + bias.permute(2, 0, 1).unsqueeze(0)
        if mask is not None:
            nw = mask.shape[0]
            attn = attn.view(bw // nw, nw, self.heads, n, n) + mask.unsqueeze(1).unsqueeze(0)
            attn = attn.view(-1, self.heads, n, n)
        return self.proj((attn.softmax(dim=-1) @ v).transpose(1, 2).reshape(bw, n, c))


class SwinBlock(nn.Module):
    def __init__(self, dim, res, heads, ws, shift, ratio):
        super().__init__()
        if res <= ws:                                   # window covers the whole map: no shift, no mask (last stage)
            ws, shift = res, 0
        self.res, self.ws, self.shift = res, ws, shift
        self.norm1 = nn.LayerNorm(dim)
        self.attn = WindowAttention(dim, ws, heads)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(dim, int(dim * ratio))
        mask = None
        if shift > 0:
            img_mask = torch.zeros(1, res, res, 1)
            cnt = 0
            for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
                for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
                    img_mask[:, hs, wsl, :] = cnt
                    cnt += 1
            mw = _window_partition(img_mask, ws).view(-1, ws * ws)
            mask = mw.unsqueeze(1) - mw.unsqueeze(2)
            mask = mask.masked_fill(mask != 0, float(-100.0)).masked_fill(mask == 0, float(0.0))
        self.register_buffer("attn_mask", mask)         # None is not saved, like timm

    def forward(self, x):
        b, l, c = x.shape
        shortcut = x
        x = self.norm1(x).view(b, self.res, self.res, c)
        if self.shift > 0:
            x = torch.roll(x, shifts=(-self.shift, -self.shift), dims=(1, 2))
        windows = self.attn(_window_partition(x, self.ws), self.attn_mask)
        x = _window_reverse(windows, self.ws, self.res, self.res)
        if self.shift > 0:
            x = torch.roll(x, shifts=(self.shift, self.shift), dims=(1, 2))
        x = shortcut + x.view(b, l, c)
        return x + self.mlp(self.norm2(x))


class PatchMerging(nn.Module):
    def __init__(self, res, dim):
        super().__init__()
        self.res = res
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = nn.LayerNorm(4 * dim)

    def forward(self, x):
        b, l, c = x.shape
        x = x.view(b, self.res, self.res, c)
        x = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], dim=-1)
        return self.reduction(self.norm(x.view(b, -1, 4 * c)))


class SwinLayer(nn.Module):
    def __init__(self, dim, res, depth, heads, ws, ratio, downsample):
        super().__init__()
        self.blocks = nn.Sequential(*[SwinBlock(dim, res, heads, ws, 0 if i % 2 == 0 else ws // 2, ratio)
                                      for i in range(depth)])
        self.downsample = PatchMerging(res, dim) if downsample else None

    def forward(self, x):
        x = self.blocks(x)
        return self.downsample(x) if self.downsample is not None else x


class SwinPatchEmbed(nn.Module):
    def __init__(self, patch, dim):
        super().__init__()
        self.proj = nn.Conv2d(3, dim, patch, stride=patch)
        self.norm = nn.LayerNorm(dim)

    def forward(self, x):
        return self.norm(self.proj(x).flatten(2).transpose(1, 2))


class SwinTransformer(nn.Module):
    def __init__(self, img=224, patch=4, dim=96, depths=(2, 2, 6, 2), heads=(3, 6, 12, 24), ws=7, ratio=4.0,
                 num_classes=1000):
        super().__init__()
        self.default_cfg = dict(IMAGENET_CFG, input_size=(3, img, img))
        self.patch_embed = SwinPatchEmbed(patch, dim)
        res = img // patch
        self.layers = nn.Sequential(*[
            SwinLayer(dim * 2 ** i, res // 2 ** i, depths[i], heads[i], ws, ratio, downsample=i < len(depths) - 1)
            for i in range(len(depths))])
        self.norm = nn.LayerNorm(dim * 2 ** (len(depths) - 1))
        self.head = nn.Linear(dim * 2 ** (len(depths) - 1), num_classes)
        _init_linear(self)

    def forward(self, x):
        x = self.layers(self.patch_embed(x))
        return self.head(self.norm(x).mean(dim=1))


def swin_tiny_patch4_window7_224(**kw):
    return SwinTransformer(**kw)
