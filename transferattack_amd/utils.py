"""Host-side utilities of the hot path, mirroring ``transferattack/utils.py`` of the reference:
constants (:12-13), model-name lists (:15-27), ``wrap_model`` / ``PreprocessingModel`` (:37-60, :72-79),
``EnsembleModel`` (:82-105), ``clamp`` (:68-69), ``save_images`` (:63-66) and ``AdvDataset`` (:108-153).

Differences that matter on MI355X: surrogates come from ``transferattack_amd.backbones`` (no
torchvision / timm here), tensors live on a HIP device, and ``save_images`` quantises on the GPU
(``ta_quantize_u8_nhwc``: (x+d)*255 truncated, NCHW -> NHWC) so only uint8 crosses PCIe.
"""
import csv
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd.function import once_differentiable

from . import _hip, backbones

img_height, img_width = 224, 224
img_max, img_min = 1., 0

cnn_model_paper = ['resnet50', 'vgg16', 'mobilenet_v2', 'inception_v3']
vit_model_paper = ['vit_base_patch16_224', 'pit_b_224', 'visformer_small', 'swin_tiny_patch4_window7_224']
cnn_model_pkg = ['vgg19', 'resnet18', 'resnet101', 'resnext50_32x4d', 'densenet121', 'mobilenet_v2']
vit_model_pkg = ['vit_base_patch16_224', 'pit_b_224', 'cait_s24_224', 'visformer_small', 'tnt_s_patch16_224',
                 'levit_256', 'convit_base', 'swin_tiny_patch4_window7_224']
generation_target_classes = [24, 99, 245, 344, 471, 555, 661, 701, 802, 919]


def default_device():
    """The HIP device this process drives (one process per GPU: LOCAL_RANK picks it)."""
    if not torch.cuda.is_available():
        raise _hip.HipExtensionError("no HIP device visible: the attack path runs on MI355X only (no CPU fallback)")
    return torch.device("cuda", int(os.environ.get("LOCAL_RANK", torch.cuda.current_device())))


def load_pretrained_model(cnn_model=(), vit_model=()):
    """(name, backbone) pairs for the eval step (utils.py:29-34); names without a local definition are
    skipped with a note instead of being downloaded."""
    for name in list(cnn_model) + list(vit_model):
        if name in backbones.available():
            yield name, backbones.create(name)
        else:
            print('=> Skipping victim {}: no local definition (reference pulls it from torchvision/timm)'.format(name))


class _Resize(nn.Module):
    """torchvision.transforms.Resize(int) on a square NCHW batch: identity at that side, else bilinear."""

    def __init__(self, size):
        super().__init__()
        self.size = size

    def forward(self, x):
        if x.shape[-1] == self.size and x.shape[-2] == self.size:
            return x
        return F.interpolate(x, size=(self.size, self.size), mode="bilinear", align_corners=False)


class _NormalizeFn(torch.autograd.Function):
    """(x - mean[c]) / std[c] as one HIP kernel each way; the backward is the producer of the input-gradient the
    update stack consumes, so it also leaves the per-tile |g| sums for the fused update (no separate K1 pass)."""

    @staticmethod
    def forward(ctx, x, mean, std):
        x = x.contiguous()
        y = torch.empty_like(x)
        _hip.normalize_fwd(x, y, mean, std)
        ctx.save_for_backward(std)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        (std,) = ctx.saved_tensors
        gy = gy.contiguous()
        gx = torch.empty_like(gy)
        _hip.normalize_bwd(gy, gx, std)
        return gx, None, None


class _Normalize(nn.Module):
    def __init__(self, mean, std):
        super().__init__()
        self.register_buffer("mean", torch.tensor(list(mean), dtype=torch.float32).view(1, -1, 1, 1), persistent=False)
        self.register_buffer("std", torch.tensor(list(std), dtype=torch.float32).view(1, -1, 1, 1), persistent=False)

    def forward(self, x):
        if x.dtype == torch.float32 and x.dim() == 4:
            # the attack path: fp32 NCHW batches.  HIP kernels, whatever device x claims to be on -- a CPU tensor is
            # refused by the binding like everywhere else (no silent torch fallback on the hot path)
            return _NormalizeFn.apply(x, self.mean.reshape(-1).contiguous(), self.std.reshape(-1).contiguous())
        return (x - self.mean) / self.std            # other dtypes / ranks: not the path this package accelerates


class _ResizeNormalizeFn(torch.autograd.Function):
    """Normalize(Resize(x)) of the 299-pixel members (utils.py:50-53, 75-76) as one HIP kernel each way: ATen's bilinear
    arithmetic (``F.interpolate(mode='bilinear', align_corners=False)``) with the Normalize folded in; the backward -- the
    last kernel of that member's input gradient -- leaves the |g| tile sums for the fused update."""

    @staticmethod
    def forward(ctx, x, mean, std, size):
        x = x.contiguous()
        y = torch.empty((x.shape[0], x.shape[1], size, size), dtype=x.dtype, device=x.device)
        _hip.resize_normalize_fwd(x, y, mean, std)
        ctx.save_for_backward(std)
        ctx.in_shape = tuple(x.shape)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        (std,) = ctx.saved_tensors
        gy = gy.contiguous()
        gx = torch.empty(ctx.in_shape, dtype=gy.dtype, device=gy.device)
        _hip.resize_normalize_bwd(gy, gx, std)
        return gx, None, None, None


def _observed(module):
    """does anybody watch this module's calls (forward / pre / backward hooks)?  Then its own forward must run."""
    return bool(module._forward_hooks or module._forward_pre_hooks or module._backward_hooks
                or getattr(module, "_backward_pre_hooks", None))


class PreprocessingModel(nn.Module):
    """normalize(resize(x)) in front of the backbone -- utils.py:72-79."""

    def __init__(self, resize, mean, std):
        super().__init__()
        self.resize = _Resize(resize)
        self.normalize = _Normalize(mean, std)

    def forward(self, x):
        size = self.resize.size
        if (x.dtype == torch.float32 and x.dim() == 4 and x.shape[-1] == x.shape[-2] < size and 2 * size <= 3 * x.shape[-1]
                and max(size, x.shape[-1]) <= 1024 and os.environ.get("TA_RESIZE_KERNEL", "1") != "0"
                and x.shape[1] == self.normalize.mean.numel() == self.normalize.std.numel()      # else: the modules' broadcast error
                and not any(_observed(m) for m in (self.resize, self.normalize))):
            # the attack path of a 299-pixel member: one fused kernel each way instead of F.interpolate + Normalize
            return _ResizeNormalizeFn.apply(x, self.normalize.mean.reshape(-1).contiguous(),
                                            self.normalize.std.reshape(-1).contiguous(), size)
        return self.normalize(self.resize(x))


def wrap_model(model):
    """nn.Sequential(PreprocessingModel, model) with the statistics chosen as utils.py:37-60 does:
    timm-style ``default_cfg`` mean/std, Inception 299 px + 0.5/0.5, otherwise ImageNet statistics.
    Kept a Sequential so ``self.model[1]`` and module names like '1.layer1.1' keep working."""
    resize = 224
    if hasattr(model, 'default_cfg'):
        mean, std = model.default_cfg['mean'], model.default_cfg['std']
    elif 'Inc' in model.__class__.__name__:
        mean, std, resize = [0.5, 0.5, 0.5], [0.5, 0.5, 0.5], 299
    else:
        mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    pre = PreprocessingModel(resize, mean, std)
    try:
        pre = pre.to(next(model.parameters()).device)
    except StopIteration:
        pass
    return nn.Sequential(pre, model)


def clamp(x, x_min, x_max):
    return torch.min(torch.max(x, x_min), x_max)


def quantize_images(images, perturbations):
    """uint8 NHWC array of floor((x + d) * 255) computed on the GPU (utils.py:64 + the add of main.py:53)."""
    dev = perturbations.device if perturbations.is_cuda else default_device()
    x = images.to(dev, torch.float32).contiguous()
    d = perturbations.to(dev, torch.float32).contiguous()
    n, c, h, w = x.shape
    out = torch.empty((n, h, w, c), dtype=torch.uint8, device=dev)
    _hip.quantize_u8_nhwc(x, d, out)
    return out.cpu().numpy()


IO_THREADS = 4            # host threads of the PNG encoder (main.py sets it from --io_threads)


def save_images(output_dir, adversaries, filenames, perturbations=None):
    """Write PNGs.  Reference signature ``save_images(output_dir, images + perturbations.cpu(), filenames)``
    (utils.py:63-66) is kept; passing ``perturbations`` separately lets the add + quantisation run fused on
    the GPU."""
    from PIL import Image
    if perturbations is None:
        perturbations = torch.zeros_like(adversaries)
    arr = quantize_images(adversaries, perturbations)
    from concurrent.futures import ThreadPoolExecutor

    def write(i):
        Image.fromarray(arr[i]).save(os.path.join(output_dir, filenames[i]))

    with ThreadPoolExecutor(max_workers=max(1, IO_THREADS)) as pool:      # zlib releases the GIL: PNG encodes run in parallel
        list(pool.map(write, range(len(filenames))))


class _FanOut(torch.autograd.Function):
    """x handed to every member of an ensemble.  Backward: the members' input gradients are added by ONE HIP kernel in
    the order autograd's input buffer would add them (last member first: ((g[m-1] + g[m-2]) + ...) + g[0]), which
    also leaves the per-tile sums of |g| for the fused update."""

    @staticmethod
    def forward(ctx, x, members):
        return tuple(x.view_as(x) for _ in range(members))

    @staticmethod
    @once_differentiable
    def backward(ctx, *grads):
        if len(grads) == 1:
            return grads[0], None
        grads = [g.contiguous() for g in grads]
        gx = torch.empty_like(grads[0])
        _hip.sum_members(grads, gx)
        return gx, None


class EnsembleModel(nn.Module):
    """Several wrapped surrogates evaluated on the same input; 'mean' averages logits, 'ind' stacks them
    (utils.py:82-105).  ``models`` stays a plain list (attacks index ``self.model.models[k]``)."""

    def __init__(self, models, mode='mean'):
        super().__init__()
        self.device = next(models[0].parameters()).device
        for model in models:
            model.to(self.device)
        self.models = models
        self.softmax = nn.Softmax(dim=1)
        self.type_name = 'ensemble'
        self.num_models = len(models)
        self.mode = mode

    def forward(self, x):
        if x.requires_grad and x.dtype == torch.float32 and x.dim() == 4 and len(self.models) > 1:
            # the attack path (HIP kernels whatever device x claims to be on, like _Normalize: no torch fallback)
            views = _FanOut.apply(x, len(self.models))
            outputs = torch.stack(self._members_on_streams(views) if self._use_streams(x) else
                                  [model(v) for model, v in zip(self.models, views)], dim=0)
        else:
            outputs = torch.stack([model(x) for model in self.models], dim=0)
        if self.mode == 'mean':
            return torch.mean(outputs, dim=0)
        if self.mode == 'ind':
            return outputs
        raise NotImplementedError

    # ---- one HIP stream per member (MI355X: 256 CUs, and most kernels of a 32-image Inception-v3 / MobileNet / ViT
    # evaluation are too small to fill them): the members of an ensemble are independent between the fan-out of x and the
    # stack of their logits, so member k's forward runs on stream k -- and, because autograd runs every backward node on
    # its forward's stream and orders streams itself, so does its backward.  Same kernels, same per-member order, same
    # bits; the members' kernels overlap on the device.  ``TA_ENS_STREAMS=0`` runs them one after the other.
    def _use_streams(self, x):
        return x.is_cuda and os.environ.get("TA_ENS_STREAMS", "1") != "0"

    def _members_on_streams(self, views):
        dev = views[0].device
        main = torch.cuda.current_stream(dev)
        streams = getattr(self, "_member_streams", None)
        if streams is None or len(streams) != len(self.models) or streams[0].device != dev:
            streams = self._member_streams = [torch.cuda.Stream(device=dev) for _ in self.models]
        outs = []
        for model, v, s in zip(self.models, views, streams):
            s.wait_stream(main)                       # x (and whatever produced it) is ready
            with torch.cuda.stream(s):
                outs.append(model(v))
        for o, s in zip(outs, streams):
            main.wait_stream(s)
            o.record_stream(main)                     # allocated on s, consumed (stacked) on main
        return outs

    def eval(self):
        for model in self.models:
            model.eval()
        return super().eval()

    def parameters(self, recurse=True):
        for model in self.models:
            yield from model.parameters(recurse)


class AdvDataset(torch.utils.data.Dataset):
    """<input_dir>/labels.csv + <input_dir>/images/*.png -> (fp32 CHW in [0,1], label, filename)
    (utils.py:108-153).  In eval mode images are read back from ``output_dir``."""

    def __init__(self, input_dir=None, output_dir=None, targeted=False, target_class=None, eval=False):
        self.targeted = targeted
        self.target_class = target_class
        self.data_dir = input_dir
        self.f2l = self.load_labels(os.path.join(self.data_dir, 'labels.csv'))
        self.filenames = list(self.f2l.keys())
        if eval:
            self.data_dir = output_dir
            print('=> Eval mode: evaluating on {}'.format(self.data_dir))
        else:
            self.data_dir = os.path.join(self.data_dir, 'images')
            print('=> Train mode: training on {}'.format(self.data_dir))
            print('Save images to {}'.format(output_dir))

    def __len__(self):
        return len(self.filenames)

    def __getitem__(self, idx):
        from PIL import Image
        filename = self.filenames[idx]
        image = Image.open(os.path.join(self.data_dir, filename))
        image = image.resize((img_height, img_width)).convert('RGB')
        image = torch.from_numpy(np.array(image).astype(np.float32) / 255).permute(2, 0, 1)
        return image, self.f2l[filename], filename

    def load_labels(self, file_name):
        f2l = {}
        with open(file_name, newline='') as fh:
            for row in csv.DictReader(fh):
                label = int(row['label'])
                if self.targeted:
                    tgt = self.target_class if self.target_class else int(row['targeted_label'])
                    f2l[row['filename']] = [label, tgt]
                else:
                    f2l[row['filename']] = label
        return f2l
