"""TEST INFRASTRUCTURE -- CPU oracle for the iterative FGSM-family hot path.

This file is a *restatement* (not a copy) of the arithmetic the reference performs on its PyTorch
CPU path; every function cites the reference ``file:line`` (relative to /root/reference) it follows.
It exists only as the checker: ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it; the product package ``transferattack_amd`` never does (its HIP path fails
loudly instead of falling back here).

Parity pin: the reference has no tests / golden vectors of its own, so this oracle
is pinned against outputs of the reference's *own classes* imported in the build container through
``oracle/ref_shim.py``; the vectors live in ``tests/golden/*.npz`` and were produced by
``oracle/gen_golden.py`` (committed).  ``tests/test_oracle_golden.py`` checks every function here against
them, bit for bit where the ATen op is thread-count invariant, <=2 ulp for ``F.interpolate``
(SURVEY.md section 8c').

All tensors are fp32 NCHW on the CPU.  Because the reference *is* PyTorch, the oracle uses the same
ATen CPU ops in the same order -- that is what "the reference CPU path" means bit-wise.  The plain-C
restatement of the same arithmetic (no torch) is ``oracle/ta_oracle.c``.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

IMG_MAX, IMG_MIN = 1.0, 0  # transferattack/utils.py:13


# ------------------------------------------------------------------------------------------------
# update stack
# ------------------------------------------------------------------------------------------------
def box_clamp(x, lo, hi):
    """min(max(x, lo), hi) with tensor bounds -- transferattack/utils.py:68-69."""
    return torch.min(torch.max(x, lo), hi)


def momentum_step(grad, momentum, decay):
    """m <- m*decay + g / mean_{CHW}|g|   (no epsilon; ``momentum`` may be the Python int 0).

    transferattack/attack.py:124-128.  An all-zero gradient image gives 0/0 = NaN momentum that
    persists; ``torch.sign(NaN) == 0`` so delta_step then takes a zero step for that image.
    """
    l1 = grad.abs().mean(dim=(1, 2, 3), keepdim=True)
    return momentum * decay + grad / l1


def delta_step(delta, data, grad, alpha, epsilon, norm="linfty"):
    """One perturbation update + projection -- transferattack/attack.py:145-153.

    linfty: d <- clamp(d + alpha*sign(g), -eps, eps); l2: d <- renorm(d + alpha*g/(|g|_2+1e-20), eps);
    then the image box d <- min(max(d, 0-x), 1-x).  ``alpha`` may be a float (also negative,
    ensemble/cwa.py:69) or a tensor shaped like delta (gradient/gra.py:149).
    """
    if norm == "linfty":
        delta = torch.clamp(delta + alpha * grad.sign(), -epsilon, epsilon)
    else:
        gn = torch.norm(grad.view(grad.size(0), -1), dim=1).view(-1, 1, 1, 1)
        delta = (delta + grad / (gn + 1e-20) * alpha).view(delta.size(0), -1) \
            .renorm(p=2, dim=0, maxnorm=epsilon).view_as(delta)
    return box_clamp(delta, IMG_MIN - data, IMG_MAX - data)


def delta_init(data, epsilon, random_start=False, norm="linfty", noise=None):
    """transferattack/attack.py:130-143.  ``noise`` (optional, same shape) replaces the
    ``uniform_(-eps, eps)`` draw so device-RNG paths can be compared on identical noise."""
    delta = torch.zeros_like(data)
    if random_start:
        if norm != "linfty":
            raise NotImplementedError("oracle restates the linfty random start only")
        if noise is None:
            delta.uniform_(-epsilon, epsilon)
        else:
            delta.copy_(noise)
        delta = box_clamp(delta, IMG_MIN - data, IMG_MAX - data)
    return delta


def quantize_u8(adv):
    """NCHW fp32 in [0,1] -> NHWC uint8 by *truncation* -- transferattack/utils.py:63-66."""
    return (adv.detach().permute(0, 2, 3, 1).cpu().numpy() * 255).astype(np.uint8)


# ------------------------------------------------------------------------------------------------
# TIM
# ------------------------------------------------------------------------------------------------
def tim_kernel(kernel_type="gaussian", kernel_size=15, nsig=3):
    """[3,1,k,k] fp32 depthwise kernel, built in fp64 then cast --
    transferattack/input_transformation/tim.py:42-66.  The normal pdf is written out
    (exp(-x^2/2)/sqrt(2pi)) instead of importing scipy; identical to scipy.stats.norm.pdf in fp64
    up to the final normalisation (checked against the reference's kernel in the golden test)."""
    kind = kernel_type.lower()
    if kind == "gaussian":
        x = np.linspace(-nsig, nsig, kernel_size)
        k1 = np.exp(-x ** 2 / 2.0) / np.sqrt(2 * np.pi)
        k2 = np.outer(k1, k1)
        k2 = k2 / k2.sum()
    elif kind == "uniform":
        k2 = np.ones((kernel_size, kernel_size)) / (kernel_size ** 2)
    elif kind == "linear":
        k1 = 1 - np.abs(np.linspace((-kernel_size + 1) // 2, (kernel_size - 1) // 2, kernel_size)
                        / (kernel_size ** 2))
        k2 = np.outer(k1, k1)
        k2 = k2 / k2.sum()
    else:
        raise Exception("Unspported kernel type {}".format(kernel_type))
    return torch.from_numpy(np.stack([k2, k2, k2])[:, None].astype(np.float32))


def tim_smooth(grad, kernel):
    """Depthwise k x k 'same' (zero-padded) correlation of the gradient -- tim.py:72-74."""
    return F.conv2d(grad, kernel, stride=1, padding="same", groups=3)


# ------------------------------------------------------------------------------------------------
# DIM
# ------------------------------------------------------------------------------------------------
def dim_draw(img_size, resize_rate=1.1, diversity_prob=0.5):
    """Draw one DIM geometry from the *CPU default generator* in the reference's order
    (rand -> randint(rnd) -> randint(top) -> randint(left)) -- dim.py:47-63.
    Returns (apply, rnd, pad_top, pad_left); when ``apply`` is False nothing else is drawn."""
    if torch.rand(1) > diversity_prob:
        return (False, img_size, 0, 0)
    img_resize = int(img_size * resize_rate)
    rnd = int(torch.randint(low=min(img_size, img_resize), high=max(img_size, img_resize),
                            size=(1,), dtype=torch.int32))
    rem = img_resize - rnd
    top = int(torch.randint(low=0, high=rem, size=(1,), dtype=torch.int32))
    left = int(torch.randint(low=0, high=rem, size=(1,), dtype=torch.int32))
    return (True, rnd, top, left)


def dim_apply(x, geom, resize_rate=1.1):
    """bilinear H->rnd, zero-pad to int(H*rate) at (top,left), bilinear back to H -- dim.py:55-68."""
    apply, rnd, top, left = geom
    if not apply:
        return x
    img_size = x.shape[-1]
    img_resize = int(img_size * resize_rate)
    rem = img_resize - rnd
    y = F.interpolate(x, size=[rnd, rnd], mode="bilinear", align_corners=False)
    y = F.pad(y, [left, rem - left, top, rem - top], value=0)
    return F.interpolate(y, size=[img_size, img_size], mode="bilinear", align_corners=False)


# ------------------------------------------------------------------------------------------------
# SIM / Admix
# ------------------------------------------------------------------------------------------------
def sim_copies(x, num_scale=5):
    """cat_i x / 2^i along the batch axis -- sim.py:36-40."""
    return torch.cat([x / (2 ** i) for i in range(num_scale)])


def admix_draw(n, num_admix=3):
    """``num_admix`` CPU-generator permutations of the batch, in call order -- admix.py:44."""
    return [torch.randperm(n) for _ in range(num_admix)]


def admix_copies(x, perms, admix_strength=0.2, num_scale=5):
    """cat_j (x + s*x[perm_j].detach()) then cat_i (./2^i) -- admix.py:40-45."""
    mixed = torch.cat([x + admix_strength * x[p].detach() for p in perms], dim=0)
    return torch.cat([mixed / (2 ** i) for i in range(num_scale)])


SIA_OPS = ("vertical_shift", "horizontal_shift", "vertical_flip", "horizontal_flip", "rotate180", "scale", "add_noise")
SIA_NOISE = 16 / 255            # sia.py:67: the noise radius is a constant of the method, not the attack's epsilon


def sia_draw(shape, num_block=3, num_copies=20, noise_source=None):
    """The random choices of ``num_copies`` calls of SIA.blocktransform on a tensor of ``shape`` (N,C,H,W), drawn in
    the reference's order from the reference's generators (sia.py:86-100): numpy -- column cuts, row cuts, then per
    block (rows outer, columns inner) the operation index and, for the two shifts, the roll step; torch -- the scale
    factor ``torch.rand(1)[0]`` and the block-shaped uniform noise.  Returns one dict per copy:
    rows / cols (cut positions incl. 0 and the size) and blocks = [(op, step, scale, noise)] in visiting order."""
    import numpy as np
    n, c, height, width = shape
    plans = []
    for _ in range(num_copies):
        cols = [0] + np.random.choice(list(range(1, width)), num_block - 1, replace=False).tolist() + [width]
        rows = [0] + np.random.choice(list(range(1, height)), num_block - 1, replace=False).tolist() + [height]
        cols.sort()
        rows.sort()
        blocks = []
        for i in range(num_block):
            for j in range(num_block):
                bh, bw = rows[i + 1] - rows[i], cols[j + 1] - cols[j]
                op = int(np.random.randint(0, high=len(SIA_OPS), dtype=np.int32))
                step, scale, noise = 0, None, None
                if op == 0:
                    step = int(np.random.randint(low=0, high=bh, dtype=np.int32))
                elif op == 1:
                    step = int(np.random.randint(low=0, high=bw, dtype=np.int32))
                elif op == 5:
                    scale = torch.rand(1)[0]
                elif op == 6:
                    noise = (noise_source((n, c, bh, bw), -SIA_NOISE, SIA_NOISE) if noise_source is not None
                             else torch.zeros(n, c, bh, bw).uniform_(-SIA_NOISE, SIA_NOISE))
                blocks.append((op, step, scale, noise))
        plans.append(dict(rows=rows, cols=cols, blocks=blocks))
    return plans


def sia_apply(x, plans):
    """cat over the copies of the block-wise transformed clone of x (sia.py:86-100) -- the ATen ops the reference's
    seven operations run (roll / flip / rot90 / mul / add+clip), applied to the same slices in the same order, so
    both the values and the autograd accumulation order are the reference's."""
    outs = []
    for plan in plans:
        rows, cols = plan["rows"], plan["cols"]
        nb = len(rows) - 1
        y = x.clone()
        for i in range(nb):
            for j in range(nb):
                op, step, scale, noise = plan["blocks"][i * nb + j]
                region = (slice(None), slice(None), slice(rows[i], rows[i + 1]), slice(cols[j], cols[j + 1]))
                block = y[region]
                if op == 0:
                    block = block.roll(step, dims=2)
                elif op == 1:
                    block = block.roll(step, dims=3)
                elif op == 2:
                    block = block.flip(dims=(2,))
                elif op == 3:
                    block = block.flip(dims=(3,))
                elif op == 4:
                    block = block.rot90(k=2, dims=(2, 3))
                elif op == 5:
                    block = scale * block
                else:
                    block = torch.clip(block + noise, 0, 1)
                y[region] = block
        outs.append(y)
    return torch.cat(outs)


# ---- BSR (input_transformation/bsr.py:41-71) ------------------------------------------------------
BSR_DEGREES = 24.0


def rotate_tensor(img, angle, mode="bilinear"):
    """torchvision.transforms.functional.rotate(img, angle, BILINEAR) for a float NCHW tensor, expand=False, centre of
    the image, no fill -- what ``T.RandomRotation(...)(x_strip)`` (bsr.py:53-55) runs.  torchvision is a dependency that
    is neither vendored in the reference nor installed here; this restates the tensor path of torchvision 0.13 (the
    reference's pin, requirements.txt:3):  functional.rotate builds the INVERSE affine matrix of a rotation by -angle
    (_get_inverse_affine_matrix(center=[0, 0], -angle, [0, 0], 1, [0, 0]) = [cos, sin, 0, -sin, cos, 0] of
    radians(-angle)); functional_tensor.rotate turns it into a sampling grid (_gen_affine_grid: pixel centres relative to
    the image centre, times theta^T / (w/2, h/2), one bmm) and samples it with
    ``grid_sample(mode, padding_mode='zeros', align_corners=False)`` -- the arithmetic itself is torch's."""
    import math
    rot = math.radians(-angle)
    matrix = [math.cos(rot), math.sin(rot), 0.0, -math.sin(rot), math.cos(rot), 0.0]
    h, w = img.shape[-2], img.shape[-1]
    theta = torch.tensor(matrix, dtype=img.dtype).reshape(1, 2, 3)
    base = torch.empty(1, h, w, 3, dtype=img.dtype)
    base[..., 0].copy_(torch.linspace(-w * 0.5 + 0.5, w * 0.5 + 0.5 - 1, steps=w))
    base[..., 1].copy_(torch.linspace(-h * 0.5 + 0.5, h * 0.5 + 0.5 - 1, steps=h).unsqueeze_(-1))
    base[..., 2].fill_(1)
    rescaled = theta.transpose(1, 2) / torch.tensor([0.5 * w, 0.5 * h], dtype=img.dtype)
    grid = base.view(1, h * w, 3).bmm(rescaled).view(1, h, w, 2)
    return torch.nn.functional.grid_sample(img, grid.expand(img.shape[0], h, w, 2), mode=mode, padding_mode="zeros",
                                           align_corners=False)


def _bsr_lengths(length, num_block):
    """BSR.get_length (bsr.py:41-45): strip lengths summing to ``length`` (numpy generator)"""
    import numpy as np
    rand = np.random.uniform(2, size=num_block)
    rand_norm = np.round(rand / rand.sum() * length).astype(np.int32)
    rand_norm[rand_norm.argmax()] += length - rand_norm.sum()
    return [int(v) for v in rand_norm]


def bsr_draw(shape, num_block=3, num_copies=20):
    """The random choices of ``num_copies`` calls of BSR.shuffle on a tensor of ``shape`` (bsr.py:57-61), in the
    reference's order from the reference's three generators: python ``random`` -- the order of the two axes, the strip
    permutations; numpy -- the strip lengths; torch -- the rotation angle of every strip
    (``torch.empty(1).uniform_(-24, 24)``, torchvision RandomRotation.get_params).  One dict per copy:
    dims (first axis, second axis), lengths0 / order0 (strips along the first axis: lengths in source order, and the
    source strip shown at each output position), and per OUTPUT strip: angle, lengths1, order1."""
    import random
    n, c, height, width = shape
    size = {2: height, 3: width}
    plans = []
    for _ in range(num_copies):
        dims = [2, 3]
        random.shuffle(dims)
        lengths0 = _bsr_lengths(size[dims[0]], num_block)
        order0 = list(range(num_block))
        random.shuffle(order0)                       # list.shuffle consumes the same draws whatever the elements are
        strips = []
        for _s in order0:
            angle = float(torch.empty(1).uniform_(-BSR_DEGREES, BSR_DEGREES).item())
            lengths1 = _bsr_lengths(size[dims[1]], num_block)
            order1 = list(range(num_block))
            random.shuffle(order1)
            strips.append(dict(angle=angle, lengths1=lengths1, order1=order1))
        plans.append(dict(dims=dims, lengths0=lengths0, order0=order0, strips=strips))
    return plans


def bsr_apply_table(x, table, num_block):
    """``bsr_apply`` driven by the product's int32 plan table (transferattack_amd.transforms.bsr_draw: per copy the first
    axis, then per OUTPUT strip src_start, length, out_start and the four fp32 entries of theta^T / (w/2, h/2), then per
    block src_start, length, out_start) -- the same ATen ops as ``bsr_apply``; the sampling grid is built from the
    table's rotation entries with the same bmm."""
    import numpy as np
    table = np.asarray(table)
    nb = num_block
    outs = []
    for row in table:
        d0 = 2 + int(row[0])
        d1 = 5 - d0
        strips = []
        for pos in range(nb):
            s0, length, _o0 = (int(v) for v in row[1 + 7 * pos:4 + 7 * pos])
            rt = row[4 + 7 * pos:8 + 7 * pos].astype(np.int32).view(np.float32)
            strip = x.narrow(d0, s0, length)
            h, w = strip.shape[-2], strip.shape[-1]
            base = torch.empty(1, h, w, 3, dtype=x.dtype)
            base[..., 0].copy_(torch.linspace(-w * 0.5 + 0.5, w * 0.5 + 0.5 - 1, steps=w))
            base[..., 1].copy_(torch.linspace(-h * 0.5 + 0.5, h * 0.5 + 0.5 - 1, steps=h).unsqueeze_(-1))
            base[..., 2].fill_(1)
            rescaled = torch.tensor([[rt[0], rt[2]], [rt[1], rt[3]], [0.0, 0.0]], dtype=x.dtype).reshape(1, 3, 2)
            grid = base.view(1, h * w, 3).bmm(rescaled).view(1, h, w, 2)
            rotated = torch.nn.functional.grid_sample(strip, grid.expand(strip.shape[0], h, w, 2), mode="bilinear",
                                                      padding_mode="zeros", align_corners=False)
            blocks = []
            for j in range(nb):
                cell = 1 + 7 * nb + 3 * (nb * pos + j)
                b0, blen, _ob = (int(v) for v in row[cell:cell + 3])
                blocks.append(rotated.narrow(d1, b0, blen))
            strips.append(torch.cat(blocks, dim=d1))
        outs.append(torch.cat(strips, dim=d0))
    return torch.cat(outs)


def bsr_apply(x, plans):
    """cat over the copies of BSR.shuffle(x) (bsr.py:57-67) with the draws of ``bsr_draw``: split along the first
    axis, shuffle, rotate every strip about its own centre, split along the second axis, shuffle, cat -- the same ATen
    ops on the same slices, so values and autograd accumulation order are the reference's."""
    outs = []
    for plan in plans:
        d0, d1 = plan["dims"]
        parts = list(x.split(plan["lengths0"], dim=d0))
        rows = []
        for pos, src in enumerate(plan["order0"]):
            st = plan["strips"][pos]
            rotated = rotate_tensor(parts[src], st["angle"])
            sub = list(rotated.split(st["lengths1"], dim=d1))
            rows.append(torch.cat([sub[k] for k in st["order1"]], dim=d1))
        outs.append(torch.cat(rows, dim=d0))
    return torch.cat(outs)


# ------------------------------------------------------------------------------------------------
# surrogate wrapper (utils.py:37-60, 72-79) and ensemble (utils.py:82-105)
# ------------------------------------------------------------------------------------------------
def preprocess(x, resize, mean, std):
    """Normalize(mean,std)(Resize(resize)(x)) -- utils.py:72-79.  torchvision's Resize on a square
    tensor that already has that side is the identity, else bilinear align_corners=False."""
    if x.shape[-1] != resize or x.shape[-2] != resize:
        x = F.interpolate(x, size=(resize, resize), mode="bilinear", align_corners=False)
    m = torch.as_tensor(mean, dtype=x.dtype).view(-1, 1, 1)
    s = torch.as_tensor(std, dtype=x.dtype).view(-1, 1, 1)
    return (x - m) / s


def preprocess_cfg(backbone):
    """(resize, mean, std) chosen as wrap_model does -- utils.py:41-56."""
    if hasattr(backbone, "default_cfg"):
        return 224, backbone.default_cfg["mean"], backbone.default_cfg["std"]
    if "Inc" in backbone.__class__.__name__:
        return 299, [0.5, 0.5, 0.5], [0.5, 0.5, 0.5]
    return 224, [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]


def logits_of(backbones, x, tap=None):
    """Surrogate forward: single wrapped model, or mean of the M members' logits (EnsembleModel
    mode='mean', utils.py:94-101).  ``tap`` (optional, single model only): called with the gradient
    that reaches the backbone's input -- PreprocessingModel's output, utils.py:78-79 -- during backward."""
    if not isinstance(backbones, (list, tuple)):
        y = preprocess(x, *preprocess_cfg(backbones))
        if tap is not None and y.requires_grad:
            y.register_hook(tap)
        return backbones(y)
    outs = [b(preprocess(x, *preprocess_cfg(b))) for b in backbones]
    return torch.mean(torch.stack(outs, dim=0), dim=0)


# ------------------------------------------------------------------------------------------------
# loop-level oracle
# ------------------------------------------------------------------------------------------------
RECIPES = {
    # name: overrides of the defaults below.  Presets: gradient/fgsm.py:28-33, ifgsm.py:30-35,
    # mifgsm.py:31-36, nifgsm.py:35-39, vmifgsm.py:28-35, vnifgsm.py:37-41, dim.py:31-40, tim.py:35-40,
    # sim.py:29-33, admix.py:32-38, ensemble/ens.py:31-36; DTS composition per SURVEY.md a17.
    "fgsm": dict(alpha=16 / 255, epoch=1, decay=0.0),
    "ifgsm": dict(decay=0.0),
    "mifgsm": dict(),
    "nifgsm": dict(lookahead=True),
    "vmifgsm": dict(variance=True),
    "vnifgsm": dict(variance=True, lookahead=True),
    "dim": dict(dim=True),
    "tim": dict(tim=True),
    "sim": dict(sim=True),
    "admix": dict(admix=True),
    "ens": dict(),
    "dts": dict(dim=True, tim=True, sim=True),
    "bsr": dict(bsr=True, num_scale=20),
}

DEFAULTS = dict(epsilon=16 / 255, alpha=1.6 / 255, epoch=10, decay=1.0, targeted=False,
                random_start=False, lookahead=False, variance=False, beta=1.5, num_neighbor=20,
                dim=False, resize_rate=1.1, diversity_prob=0.5, tim=False, kernel_type="gaussian",
                kernel_size=15, sim=False, admix=False, num_scale=5, num_admix=3, admix_strength=0.2,
                bsr=False, num_block=3)


def run_attack(name, backbones, data, label, trace=None, **overrides):
    """Whole K-iteration attack on the CPU; returns delta (detached) like Attack.forward
    (attack.py:67-102; VMI variant vmifgsm.py:60-97).

    RNG: DIM / Admix draw from the CPU default generator per call exactly as the reference does;
    VMI neighbours draw ``zeros_like(delta).uniform_(-r, r)`` (vmifgsm.py:50) -- on the CPU that is
    also the default generator, so seeding with ``torch.manual_seed`` reproduces the reference.
    ``trace`` (optional list) receives per-iteration dicts (grad, momentum, delta, noise...) used
    by the GPU parity tests to feed the HIP kernels the oracle's own intermediate tensors.
    """
    cfg = dict(DEFAULTS)
    cfg.update(RECIPES[name])
    cfg.update(overrides)
    eps, alpha, decay = cfg["epsilon"], cfg["alpha"], cfg["decay"]
    if cfg["targeted"]:
        assert len(label) == 2  # attack.py:76-78
        label = label[1]
    data = data.clone().detach()
    label = label.clone().detach()
    kernel = tim_kernel(cfg["kernel_type"], cfg["kernel_size"]) if cfg["tim"] else None
    copies = 1
    if cfg["sim"]:
        copies = cfg["num_scale"]
    if cfg["admix"]:
        copies = cfg["num_scale"] * cfg["num_admix"]
    if cfg["bsr"]:
        copies = cfg["num_scale"]
    ce = torch.nn.CrossEntropyLoss()

    def transform(x, momentum, rec):
        if cfg["lookahead"]:                       # nifgsm.py:35-39
            x = x + alpha * decay * momentum
        if cfg["bsr"]:                             # bsr.py:63-67
            return bsr_apply(x, bsr_draw(tuple(x.shape), cfg["num_block"], cfg["num_scale"]))
        if cfg["admix"]:                           # admix.py:40-45
            perms = admix_draw(x.size(0), cfg["num_admix"])
            rec.setdefault("perms", []).append([p.clone() for p in perms])
            x = admix_copies(x, perms, cfg["admix_strength"], cfg["num_scale"])
        elif cfg["sim"]:                           # sim.py:36-40
            x = sim_copies(x, cfg["num_scale"])
        if cfg["dim"]:                             # dim.py:42-68 (after SIM for DTS)
            geom = dim_draw(x.shape[-1], cfg["resize_rate"], cfg["diversity_prob"])
            rec.setdefault("geoms", []).append(geom)
            x = dim_apply(x, geom, cfg["resize_rate"])
        return x

    def grad_at(x_in, delta, momentum, rec):
        tap = None
        if trace is not None and not isinstance(backbones, (list, tuple)):
            # d(loss)/d(backbone input): what Normalize's backward (gy / std, utils.py:76) turns into the gradient below
            tap = lambda gy: rec.setdefault("grads_y", []).append(gy.detach().clone())     # noqa: E731
        logits = logits_of(backbones, transform(x_in, momentum, rec), tap)
        lab = label.repeat(copies) if copies > 1 else label
        loss = -ce(logits, lab) if cfg["targeted"] else ce(logits, lab)     # attack.py:110-115
        g = torch.autograd.grad(loss, delta, retain_graph=False, create_graph=False)[0]
        if kernel is not None:
            g = tim_smooth(g, kernel)              # tim.py:72-74
        if trace is not None:
            rec.setdefault("grads", []).append(g)  # every get_grad result of the iteration, in call order
        return g

    delta = delta_init(data, eps, cfg["random_start"]).requires_grad_(True)
    momentum, variance = 0, 0
    for it in range(cfg["epoch"]):
        rec = {}
        grad = grad_at(data + delta, delta, momentum, rec)
        if cfg["variance"]:                        # vmifgsm.py:86-95
            momentum_new = momentum_step(grad + variance, momentum, decay)
            acc = 0
            radius = cfg["beta"] * eps
            noises = []
            for _ in range(cfg["num_neighbor"]):
                noise = torch.zeros_like(delta).uniform_(-radius, radius)
                noises.append(noise)
                acc = acc + grad_at(data + delta + noise, delta, momentum_new, rec)
            variance_new = acc / cfg["num_neighbor"] - grad
            rec["noises"] = noises
            rec["variance_in"] = variance
            variance = variance_new
        else:
            momentum_new = momentum_step(grad, momentum, decay)
        delta_new = delta_step(delta.detach(), data, momentum_new, alpha, eps)
        if trace is not None:
            rec.update(grad=grad, momentum_in=momentum, momentum=momentum_new,
                       delta_in=delta.detach().clone(), delta=delta_new.clone())
            trace.append(rec)
        momentum = momentum_new
        delta = delta_new.detach().requires_grad_(True)
    return delta.detach()
