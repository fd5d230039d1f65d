"""Everywhere Attack (Zeng et al., AAAI 2025) -- targeted: besides the whole image, every iteration attacks four of the
nine cells of a 3 x 3 partition (the rest of the image replaced by the ImageNet mean) towards the same target, on top of
clean-feature mixup (CFM: with probability 0.1 per layer, the output of a convolution / linear layer is blended,
channel-wise random ratio below 0.75, with the CLEAN features of a shuffled batch member recorded before the attack),
resolution-keeping DI, 5 x 5 TI and momentum.  Mirror of transferattack/input_transformation/everywhere.py:14-412
(its 'CDTM' configuration, the only one the class runs).

Draws as in the reference: one ``torch.rand(1)`` per eligible layer and forward (host), a host permutation and a host
matrix of ratios when a layer mixes, ``torch.randperm(9)`` for the cells, numpy for DI.  The method brings its own
update arithmetic -- momentum on g / SUM|g|, and the image box as ``clamp(x + delta, 0, 1) - x`` -- which rounds
differently from the base hooks, so it is kept as written (elementwise device ops); the TI smoothing is
``ta_depthwise_conv2d_same``."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _hip
from ..attack import Attack
from .tim import _gaussian_profile

config = {
    'p': 1.,
    'mixed_image_type_feature': 'C',        # 'C': clean features / 'A': the current batch's
    'shuffle_image_feature': 'SelfShuffle',
    'blending_mode_feature': 'M',           # 'M': convex interpolation / 'A': addition
    'mix_lower_bound_feature': 0.,
    'mix_upper_bound_feature': 0.75,
    'mix_prob': 0.1,
    'divisor': 4,
    'channelwise': True,
    'mixup_layer': 'conv_linear_include_last',
    'comment': 'CFM-RDI Main Result',
}


class EverywhereAttack(Attack):
    """Official arguments: epsilon=16/255, lr=1.6/255, epoch=300, num_blocks=16, N=9, img_size=224 (targeted)."""

    def __init__(self, model_name, targeted, attack="everywhere", img_size=224, epsilon=16/255, lr=1.6/255, epoch=300,
                 num_blocks=16, N=9):
        super().__init__(attack, model_name, epsilon, targeted, random_start=True, norm='linfty', loss='crossentropy',
                         device=None)
        self.img_size, self.lr, self.max_iterations, self.num_blocks, self.N = img_size, lr, epoch, num_blocks, N

    def forward(self, image_tensor, labels):
        X_ori = image_tensor.to(self.device)
        X_adv = advanced_fgsm_every_memory(attack_type='CDTM', source_model=self.model, x=X_ori, y=labels[0].to(self.device),
                                           lr=self.lr, target_label=labels[1].to(self.device),
                                           num_iter=self.max_iterations, max_epsilon=self.epsilon, device=self.device)
        return (X_adv - X_ori).detach()


class FeatureMixupEverywhere(nn.Module):
    """forward hooks on every convolution / linear layer of ``model`` whose output is at most input_size / divisor wide:
    recording mode stores the clean outputs, mixup mode blends them in"""

    def __init__(self, model, input_size):
        super().__init__()
        self.mixup_layer, self.prob, self.channelwise = config['mixup_layer'], config['mix_prob'], config['channelwise']
        self.model, self.input_size = model, input_size
        self.record, self.outputs, self.forward_hooks = False, {}, []
        self.batchsize, self.masknum, self.selected_region = 1, 0, []
        if self.mixup_layer in ('conv_linear_no_last', 'conv_linear_include_last'):
            kinds = (nn.Conv2d, nn.Linear)
        elif self.mixup_layer in ('bn', 'relu'):
            kinds = (nn.BatchNorm2d,)
        else:
            kinds = (nn.Conv2d,)
        leaves = [m for m in model.modules() if not list(m.children()) and type(m) in kinds]      # depth-first, as the reference
        self.layer_num = len(leaves)
        for index, module in enumerate(leaves):
            self.forward_hooks.append(module.register_forward_hook(self.save_outputs_hook(index)))

    def save_outputs_hook(self, layer_idx):
        low, high = config['mix_lower_bound_feature'], config['mix_upper_bound_feature']

        def hook_fn(module, inputs, output):
            if not (type(module) == nn.Linear or output.size()[-1] <= self.input_size // config['divisor']):
                return None
            if self.mixup_layer == 'conv_linear_no_last' and layer_idx + 1 == self.layer_num and type(module) == nn.Linear:
                return None
            if self.record:
                self.outputs[layer_idx] = output.clone().detach()
                return None
            if layer_idx not in self.outputs:
                return None
            if torch.rand(1).item() > self.prob:
                return output
            clean = output.clone().detach() if config['mixed_image_type_feature'] == 'A' else self.outputs[layer_idx].clone().detach()
            if config['shuffle_image_feature'] == 'SelfShuffle':        # the same shuffle of the batch in every selected cell
                order = torch.randperm(self.batchsize)
                rows = order.clone()
                for region in self.selected_region:
                    rows = torch.cat([rows, order + (region + 1) * self.batchsize], dim=0)
                total = (len(self.selected_region) + 1) * self.batchsize
                clean = clean[rows].view(clean[:total].size())
            samples, channels = clean.shape[0], clean.shape[1]
            ratio = torch.rand(samples, channels) if self.channelwise else torch.rand(samples)
            ratio = ratio * (high - low) + low
            ratio = ratio.view(ratio.shape + (1,) * (output.dim() - ratio.dim())).to(output.device)
            if self.mixup_layer == 'relu':
                output = F.relu(output, inplace=True)
            if config['blending_mode_feature'] == 'M':
                return (1 - ratio) * output + ratio * clean
            return output + ratio * clean

        return hook_fn

    def start_feature_record(self):
        self.record = True

    def end_feature_record(self):
        self.record = False

    def set_paras(self, batchsize, masknum, selected_region):
        self.batchsize, self.masknum, self.selected_region = batchsize, masknum, selected_region

    def remove_hooks(self):
        for hook in self.forward_hooks:
            hook.remove()
        del self.outputs

    def forward(self, x):
        return self.model(x)


class CELoss(nn.Module):
    def __init__(self, labels):
        super().__init__()
        self.labels = labels
        self.ce = nn.CrossEntropyLoss(reduction='mean')
        self.labels.requires_grad = False

    def forward(self, logits):
        return self.ce(logits, self.labels)


class LogitLoss(nn.Module):
    def __init__(self, labels, targeted=True):
        super().__init__()
        self.labels, self.targeted = labels, targeted
        self.labels.requires_grad = False

    def forward(self, logits):
        loss = (1 * logits.gather(1, self.labels.unsqueeze(1)).squeeze(1)).sum()
        return loss if self.targeted else -loss


def DI_keepresolution(X_in):
    """nearest-neighbour shrink by up to 29 pixels and zero padding back to the input size, probability 0.7 (numpy draws,
    all taken whether or not the branch is)"""
    img_size = X_in.shape[3]
    rnd = np.random.randint(img_size - 29, img_size, size=1)[0]
    rem = img_size - rnd
    pad_top = np.random.randint(0, rem, size=1)[0]
    pad_left = np.random.randint(0, rem, size=1)[0]
    if np.random.rand(1) <= 0.7:
        return F.pad(F.interpolate(X_in, size=(rnd, rnd)), (pad_left, rem - pad_left, pad_top, rem - pad_top),
                     mode='constant', value=0)
    return X_in


def gkern(kernlen=15, nsig=3):
    profile = _gaussian_profile(kernlen, nsig)
    plane = np.outer(profile, profile)
    return plane / plane.sum()


def _cell_masks(count, batch, H, W, device):
    masks = torch.zeros(count, batch, 3, H, W).to(device)
    h_block, w_block = H // 3, W // 3
    for cell in range(count):
        up, left = int(np.floor(cell / 3) * h_block), int((cell % 3) * w_block)
        masks[cell, :, :, up:min(up + h_block, H), left:min(left + w_block, W)] = 1
    return masks


def advanced_fgsm_every_memory(attack_type, source_model, x, y, device, target_label, num_iter, max_epsilon, lr, mu=1.0):
    """CFM + Everywhere: ``attack_type`` letters -- C clean-feature mixup, D resolution-keeping DI, T 5 x 5 TI, M momentum"""
    sample_num = 4
    batch, _, H, W = x.size()
    mask = _cell_masks(9, batch, H, W, device)
    labels_combine = torch.cat([target_label] * (sample_num + 1), dim=0)
    delta = torch.zeros_like(x, requires_grad=True).to(device)
    if 'targeted' not in config:
        config['targeted'] = True
    if "M" not in attack_type and "N" not in attack_type:
        mu = 0
    ti_kernel_size = 5
    ti_kernel = None
    if 'T' in attack_type:
        ti_kernel = torch.from_numpy(gkern(ti_kernel_size, 3).astype(np.float32)).to(device).contiguous()
    source_model.eval()
    eps, alpha = max_epsilon, lr
    g = 0
    loss_fn = LogitLoss(labels_combine, config['targeted']) if config['targeted'] else None
    mean_tensor = torch.Tensor([0.485, 0.456, 0.406]).type_as(x)[None, :, None, None] * torch.ones_like(x)

    def with_cells(whole, cells):
        """the whole image followed by, per chosen cell, that cell of it on the mean-colour canvas"""
        stacked = torch.zeros((len(cells) + 1) * batch, 3, H, W).to(device)
        stacked[:batch] = whole
        for slot, cell in enumerate(cells):
            stacked[(slot + 1) * batch:(slot + 2) * batch] = (mask[cell] * whole) + ((1 - mask[cell]) * mean_tensor)
        return stacked

    consumed_iteration = 0
    if 'C' in attack_type:                                   # record the clean features of the image and all nine cells
        with torch.no_grad():
            model = FeatureMixupEverywhere(source_model, x.size()[-1])
            model.start_feature_record()
            model(with_cells(x, range(9)))
            model.end_feature_record()
            consumed_iteration = 1                           # that pass counts as one iteration (a fair budget)
    else:
        model = source_model

    for t in range(num_iter):
        if t < consumed_iteration:
            continue
        idx = torch.randperm(9)
        model.set_paras(batchsize=batch, masknum=9, selected_region=idx[:4])
        inputs = with_cells(x + delta, [idx[i] for i in range(sample_num)])
        if 'D' in attack_type:
            inputs = DI_keepresolution(inputs)
        ghat = torch.autograd.grad(loss_fn(model(inputs)), delta, retain_graph=False, create_graph=False)[0]
        if ti_kernel is not None:
            smoothed = torch.empty_like(ghat)
            _hip.depthwise_conv2d_same(ghat.contiguous(), smoothed, ti_kernel)
            ghat = smoothed
        if 'M' in attack_type or 'N' in attack_type:
            g = mu * g + ghat / torch.sum(torch.abs(ghat), dim=[1, 2, 3], keepdim=True)
        else:
            g = ghat
        delta.data = delta.data + alpha * g.sign()
        delta.data = delta.data.clamp(-eps, eps)
        delta.data = ((x + delta.data).clamp(0, 1)) - x
    x_adv = (x + delta).detach()
    if 'C' in attack_type:
        model.remove_hooks()
    return x_adv
