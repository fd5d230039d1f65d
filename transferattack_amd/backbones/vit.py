"""ViT-B/16 surrogate (Dosovitskiy et al. 2020) with timm's parameter names (``cls_token``,
``pos_embed``, ``patch_embed.proj``, ``blocks.N.attn.qkv`` ...) so a timm checkpoint loads.

The reference obtains it with ``timm.create_model('vit_base_patch16_224', pretrained=True)``
(transferattack/attack.py:56-57); ``wrap_model`` reads ``default_cfg['mean'/'std']`` from the model
(transferattack/utils.py:44-47) -- for the augreg weights timm ships for this name that is 0.5/0.5,
made explicit here.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class PatchEmbed(nn.Module):
    def __init__(self, patch, dim):
        super().__init__()
        self.proj = nn.Conv2d(3, dim, patch, stride=patch)

    def forward(self, x):
        return self.proj(x).flatten(2).transpose(1, 2)


class Attention(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.heads = heads
        self.qkv = nn.Linear(dim, 3 * dim)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        b, t, d = x.shape
        q, k, v = self.qkv(x).view(b, t, 3, self.heads, d // self.heads).permute(2, 0, 3, 1, 4)
        y = F.scaled_dot_product_attention(q, k, v)
        return self.proj(y.transpose(1, 2).reshape(b, t, d))


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class Block(nn.Module):
    def __init__(self, dim, heads, ratio):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = Attention(dim, heads)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = Mlp(dim, int(dim * ratio))

    def forward(self, x):
        x = x + self.attn(self.norm1(x))
        return x + self.mlp(self.norm2(x))


class VisionTransformer(nn.Module):
    def __init__(self, img=224, patch=16, dim=768, depth=12, heads=12, ratio=4.0, num_classes=1000):
        super().__init__()
        self.default_cfg = {"mean": (0.5, 0.5, 0.5), "std": (0.5, 0.5, 0.5), "input_size": (3, img, img)}
        self.patch_embed = PatchEmbed(patch, dim)
        tokens = (img // patch) ** 2 + 1
        self.cls_token = nn.Parameter(torch.zeros(1, 1, dim))
        self.pos_embed = nn.Parameter(torch.randn(1, tokens, dim) * 0.02)
        self.blocks = nn.Sequential(*[Block(dim, heads, ratio) for _ in range(depth)])
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        self.head = nn.Linear(dim, num_classes)
        nn.init.normal_(self.cls_token, std=1e-6)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                nn.init.zeros_(m.bias)

    def forward(self, x):
        x = self.patch_embed(x)
        x = torch.cat([self.cls_token.expand(x.size(0), -1, -1), x], dim=1) + self.pos_embed
        x = self.norm(self.blocks(x))
        return self.head(x[:, 0])


def vit_base_patch16_224(**kw):
    return VisionTransformer(**kw)


def vit_tiny_patch16_224(**kw):
    return VisionTransformer(dim=192, depth=12, heads=3, **kw)
