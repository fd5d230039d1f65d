"""``Attack``: the reference's base class / plug-in API (transferattack/attack.py:8-169) with the
per-iteration update stack on hand-written gfx950 kernels.

Hook names, signatures, defaults and error behaviour are the reference's, so its subclasses keep working:

    load_model(model_name)            attack.py:40-65      get_momentum(grad, momentum, **kw)   :124-128
    forward(data, label, **kw)        :67-102              init_delta(data, **kw)               :130-143
    get_logits(x, **kw)               :104-108             update_delta(delta, data, grad, alpha, **kw) :145-153
    get_loss(logits, label)           :110-115             loss_function(loss)                  :155-162
    get_grad(loss, delta, **kw)       :118-122             transform(data, **kw)                :164-165

What runs where: the surrogate forward/backward is PyTorch-ROCm (MIOpen / rocBLAS); ``get_momentum`` and
``update_delta`` are HIP kernels (``ta_momentum``, ``ta_update_delta_linf|l2``).  When a subclass overrides
neither hook, ``forward`` replaces the pair by ONE fused launch (``ta_mi_update``): momentum accumulate + sign +
alpha-step + eps-ball + image-box in a single pass that also writes ``data + delta`` for the next iteration
(attack.py:88) -- 24 (+4) B/element instead of the reference's 13 ATen kernels / ~116 B/element.  The per-image
sum|g| it needs comes from the kernel that wrote the gradient (``_hip`` partials registry), so g is read once.  With
``decay == 0`` (FGSM, I-FGSM) the momentum is neither read nor stored (16 B/element).  There is no CPU fallback:
tensors must be on a HIP device.
"""
import os

import torch
from torch.autograd.function import once_differentiable
import torch.nn as nn

from . import _hip, backbones
from .utils import EnsembleModel, default_device, wrap_model, img_max, img_min, clamp  # noqa: F401


class _AdvInput(torch.autograd.Function):
    """``data + delta`` when the sum is already in memory (written by the fused update of the previous iteration):
    hands out that buffer, the gradient flows to ``delta`` unchanged -- exactly AddBackward's behaviour."""

    @staticmethod
    def forward(ctx, delta, x_adv):
        return x_adv.detach()

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        return g, None


def takes_channels_last(backbone):
    """Which surrogates the opt-in NHWC arrangement (``TA_CHANNELS_LAST=1``) applies to: not Inception-v3 (ROCm 7.2's NHWC
    fp32 backward-data kernels fault on its 1x7 / 7x1 convolutions) and, unless ``TA_VGG_CHANNELS_LAST=1``, not the VGGs
    (input gradient 2-4x further from the fp64 truth for +8 % throughput)."""
    name = backbone.__class__.__name__
    if "Inc" in name:
        return False
    return not ("VGG" in name.upper() and os.environ.get("TA_VGG_CHANNELS_LAST", "0") != "1")


def deterministic_mode():
    """``TA_DETERMINISTIC=1``: the same command on the same GPU writes the same bytes.  The reference's contract is "final uint8
    bit-exact" (utils.py:63-66); on the CPU it reproduces itself, on a GPU the libraries' default algorithms do not -- MIOpen's
    backward-data solvers and ATen's max-pool backward accumulate with atomics, rocBLAS may split K with them.  This switch
    (read when an attack is constructed) asks every library for its deterministic algorithms:
      * ``torch.backends.cudnn.deterministic``: PyTorch-ROCm then sets MIOPEN_CONVOLUTION_ATTRIB_DETERMINISTIC on every
        convolution descriptor -- MIOpen's find / immediate mode skip the solvers that use atomics;
      * ``torch.use_deterministic_algorithms(True, warn_only=True)``: rocBLAS / hipBLASLt atomics off, ATen's deterministic
        variants where they exist (an op without one warns instead of raising);
    this package's own kernels never use floating-point atomics (include/ta_hip.h).  What remains outside: ATen's max-pool
    backward on the plain module path of a CNN surrogate -- the fused ResNet path (folded BatchNorm + NHWC, bench.py's
    arrangement) replaces it by the gather kernel ``ta_maxpool_bwd_relu``; that arrangement is the one
    ``tests/test_hip_attacks.py::test_deterministic_mode_writes_identical_pngs`` pins.  The torch flags are process-wide: they
    stay as this function set them until an attack is constructed (or this function is called) with the variable unset or 0,
    which puts back what it found.  ``torch.utils.deterministic.fill_uninitialized_memory`` is switched off with it: that
    companion of ``use_deterministic_algorithms`` NaN-fills every ``torch.empty`` -- one more pass over each buffer of the hot
    loops, all of which the HIP kernels overwrite completely.  Returns whether the mode is on."""
    global _flags_before_deterministic
    if os.environ.get("TA_DETERMINISTIC", "0") != "1":
        if _flags_before_deterministic is not None:
            cudnn_det, algos, warn_only, fill = _flags_before_deterministic
            torch.backends.cudnn.deterministic = cudnn_det
            torch.use_deterministic_algorithms(algos, warn_only=warn_only)
            torch.utils.deterministic.fill_uninitialized_memory = fill
            _flags_before_deterministic = None
        return False
    if _flags_before_deterministic is None:
        _flags_before_deterministic = (torch.backends.cudnn.deterministic, torch.are_deterministic_algorithms_enabled(),
                                       torch.is_deterministic_algorithms_warn_only_enabled(),
                                       torch.utils.deterministic.fill_uninitialized_memory)
    torch.backends.cudnn.deterministic = True
    torch.use_deterministic_algorithms(True, warn_only=True)
    torch.utils.deterministic.fill_uninitialized_memory = False
    return True


_flags_before_deterministic = None          # the process-wide torch flags deterministic_mode() found when it first changed them


class Attack(object):
    """Base class for all attacks (same constructor as transferattack/attack.py:12-38)."""

    def __init__(self, attack, model_name, epsilon, targeted, random_start, norm, loss, device=None):
        if norm not in ['l2', 'linfty']:
            raise Exception("Unsupported norm {}".format(norm))
        deterministic_mode()
        self.attack = attack
        self.model = self.load_model(model_name)
        self.epsilon = epsilon
        self.targeted = targeted
        self.random_start = random_start
        self.norm = norm
        if isinstance(self.model, EnsembleModel):
            self.device = self.model.device
        else:
            self.device = next(self.model.parameters()).device if device is None else device
        self.loss = self.loss_function(loss)
        # RNG state of the device-side draws (random start, VMI neighbours): Philox (seed, running offset)
        self.rng_seed = int(os.environ.get("TA_SEED", "0"))
        self.rng_offset = 0
        # test hook: callable(shape, low, high) -> device tensor replacing an in-kernel uniform draw
        self.noise_source = None
        # test hook: callable(shape, mean, std) -> tensor replacing the normal draw of the L2 random start
        self.normal_source = None
        # test hook of loops that never call get_grad (the folded VMI chain): callable(iteration, gradient)
        self.grad_probe = None
        # test hook of the folded plain loop: callable(iteration, gy) -> the tensor the update consumes INSTEAD of gy, the
        # backbone's own input gradient (tests replay the reference's recorded gy through the default loop form)
        self.grad_inject = None

    # ------------------------------------------------------------------------------------------ model
    def load_model(self, model_name):
        """Build the surrogate(s) from ``transferattack_amd.backbones`` (torchvision names first, then timm
        names, as attack.py:52-57), put them in eval mode on this process's HIP device and wrap them with
        the preprocessing layer.  Override for customised surrogates, as in the reference."""
        def load_single_model(name):
            model = backbones.create(name)          # raises ValueError('Model {} not supported')
            for p in model.parameters():
                p.requires_grad_(False)             # only d(loss)/d(delta) is ever needed
            if os.environ.get("TA_FOLD_BN", "0") == "1":
                backbones.fold_batchnorm(model)     # opt-in: eval-mode BN folded into the convolutions
            wrapped = wrap_model(model.eval().to(default_device()))
            # NHWC is opt-in and skipped for Inception-v3 -- on ROCm 7.2 the NHWC fp32 backward-data kernels fault on its
            # asymmetric 1x7 / 7x1 convolutions (profiles/r01/ens_diag/inc_nhwc.txt) -- and for the VGGs: +8 % throughput
            # (profiles/r01/ens_diag/vgg_nhwc.txt) for an input gradient 2-4x further from the fp64 truth (4.4e-3 -> up to
            # 1.75e-2 on the seeded VGG-16; GPUTEST_r03: 13 un-normalised conv + ReLU stages amplify the other
            # accumulation order of MIOpen's NHWC kernels) is not a trade a parity-first engine makes by default
            if os.environ.get("TA_CHANNELS_LAST", "0") == "1" and takes_channels_last(model):
                wrapped = wrapped.to(memory_format=torch.channels_last)
            return wrapped

        if isinstance(model_name, list):
            return EnsembleModel([load_single_model(name) for name in model_name])
        return load_single_model(model_name)

    # ------------------------------------------------------------------------------------------- loop
    def forward(self, data, label, **kwargs):
        """The general attack procedure (attack.py:67-102).

        data (N, C, H, W) images in [0, 1]; label (N,) or [ground-truth, target] when targeted.
        Returns the perturbation ``delta`` (detached, on ``self.device``)."""
        if self.targeted:
            assert len(label) == 2
            label = label[1]
        data = data.clone().detach().to(self.device)
        label = label.clone().detach().to(self.device)
        self._attach_byte_source(data)

        delta = self.init_delta(data)
        momentum = 0
        fused = self._can_fuse_update()
        chain = self._normalize_chain(data) if fused else None
        if chain is not None:
            return self._forward_normalize_folded(data, label, delta, chain)
        x_adv = None                      # data + delta as left behind by the fused update (bit-identical to the add)
        for it in range(self.epoch):
            x_in = data + delta if x_adv is None else _AdvInput.apply(delta, x_adv)
            logits = self.get_logits(self.transform(x_in, momentum=momentum))
            loss = self.get_loss(logits, label)
            grad = self.get_grad(loss, delta)
            if fused:
                if x_adv is None and it + 1 < self.epoch:
                    x_adv = torch.empty_like(data)
                momentum = self._fused_update(grad, momentum, delta, data,
                                              x_adv=x_adv if it + 1 < self.epoch else None)
            else:
                momentum = self.get_momentum(grad, momentum)
                delta = self.update_delta(delta, data, momentum, self.alpha)
        return delta.detach()

    def _overrides(self, *hooks):
        """does this attack -- its class or the instance itself -- replace any of the named base hooks?"""
        return any(getattr(type(self), h) is not getattr(Attack, h) or h in self.__dict__ for h in hooks)

    def _can_fuse_update(self):
        return (self.norm == 'linfty' and not self._overrides("get_momentum", "update_delta")
                and not isinstance(self.alpha, torch.Tensor))

    def _normalize_chain(self, data):
        """(mean, std) of the surrogate's Normalize if NOTHING else sits between ``data + delta`` (attack.py:88) and the
        backbone, and nobody can observe the tensors in between: the base ``transform`` (identity), ``get_logits`` and
        ``get_grad``; ``self.model`` the plain ``nn.Sequential(PreprocessingModel, backbone)`` of ``wrap_model`` whose Resize is
        the identity at this image size; no hook on the wrapper, the preprocessing layer or its two sub-modules, and no
        process-wide module hook (``torch.nn.modules.module.register_module_*_hook``: it would be called for the modules the
        folded loop skips).  Then the Normalize is folded into both ends of the iteration (``_forward_normalize_folded``;
        ``grad_probe`` / ``grad_inject`` are served inside it).  ``TA_FOLD_NORMALIZE=0`` turns it off."""
        from torch.nn.modules import module as _nn_module
        from .utils import PreprocessingModel, _Normalize, _Resize
        if os.environ.get("TA_FOLD_NORMALIZE", "1") == "0" or self._overrides("transform", "get_logits", "get_grad"):
            return None
        for registry in ("_global_forward_hooks", "_global_forward_pre_hooks", "_global_backward_hooks",
                         "_global_backward_pre_hooks", "_global_forward_hooks_always_called"):
            if getattr(_nn_module, registry, None):
                return None
        model = self.model
        if type(model) is not nn.Sequential or len(model) != 2 or type(model[0]) is not PreprocessingModel:
            return None
        pre = model[0]
        if type(pre.resize) is not _Resize or type(pre.normalize) is not _Normalize:
            return None
        for m in (model, pre, pre.resize, pre.normalize):      # (these three modules behave alike in train and eval mode)
            if m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or getattr(m, "_backward_pre_hooks", None):
                return None
        if (data.dim() != 4 or data.dtype != torch.float32 or not data.is_contiguous()
                or data.shape[-1] != pre.resize.size or data.shape[-2] != pre.resize.size
                or pre.normalize.mean.numel() != data.shape[1] or pre.normalize.std.numel() != data.shape[1]
                or pre.normalize.mean.device != data.device):
            return None
        return pre.normalize.mean.reshape(-1).contiguous(), pre.normalize.std.reshape(-1).contiguous()

    def _forward_normalize_folded(self, data, label, delta, chain):
        """The loop of ``forward`` (attack.py:86-100) for the plain chain of ``_normalize_chain``, with the surrogate's Normalize
        (utils.py:72-79) folded into the two HIP kernels on either side of the backbone:

            y     = ((data + delta) - mean) / std       ta_normalize_adv_fwd: attack.py:88's add + Normalize, one pass; the image
                                                        comes from the byte source (5 B in, 4 B out per element)
            gy    = d loss / d y                        the backbone's own input gradient -- its last kernel (the stem kernel
                                                        of the fused ResNet path) leaves the sums of |gy / std|
            m, delta <- update(gy / std, ...)           ta_mi_update_std: Normalize's backward formed inline

        Per element and iteration the memory-bound passes move 9 + 21 bytes instead of 8 + 8 + 25 (no ``x + delta`` store, no
        ``gx = gy / std`` store / reload); every rounding point of the module path is kept, so momentum and delta carry its
        bits (``tests/test_hip_kernels.py::test_normalize_folded_update``)."""
        mean, std = chain
        backbone = self.model[1]
        momentum = 0
        src = self._byte_source_of(data)
        # a surrogate kept in NHWC memory (TA_CHANNELS_LAST=1) gets its input written in NHWC: no layout copy before its first layer
        first = next((p for p in backbone.parameters() if p.dim() == 4), None)
        nhwc = (first is not None and not first.is_contiguous() and first.is_contiguous(memory_format=torch.channels_last)
                and data.shape[1] == 3 and data[0, 0].numel() % 4 == 0)
        for it in range(self.epoch):
            y = torch.empty_like(data, memory_format=torch.channels_last) if nhwc else torch.empty_like(data)
            _hip.normalize_adv_fwd(data, delta.detach(), y, mean, std, data_u8=src)
            y.requires_grad_(True)
            setattr(y, _hip._SCALE_ATTR, std)          # a fused backbone hands it to its last backward kernel
            loss = self.get_loss(backbone(y), label)
            gy = torch.autograd.grad(loss, y, retain_graph=False, create_graph=False)[0]
            if self.grad_probe is not None:             # test hook: the gradient of attack.py:118-122, materialised for it
                self.grad_probe(it, gy / std.view(1, -1, 1, 1))
            if self.grad_inject is not None:            # test hook: a recorded gy replaces the device's (a fresh tensor: it
                gy = self.grad_inject(it, gy).contiguous()      # carries no sums, so the sum-only pass precedes the update)
            momentum = self._fused_update(gy, momentum, delta, data, grad_std=std)
        return delta.detach()

    def _fused_update(self, grad, momentum, delta, data, variance=None, alpha=None, x_adv=None, grad_std=None):
        """get_momentum + update_delta in one pass; ``delta`` (a leaf) is updated in place -- the graph of
        this iteration has already been consumed by ``get_grad``.  Returns the new momentum (a tensor, or the
        Python 0 it started as when ``decay == 0``: ``m*0 + g/mean|g|`` never looks at the old momentum, so it is
        neither stored nor read back -- FGSM / I-FGSM move 16 B/element).  ``x_adv`` (optional buffer) receives
        ``data + delta`` for the next iteration."""
        grad = grad.contiguous()
        m_in = momentum if isinstance(momentum, torch.Tensor) else None
        if self.decay == 0 and m_in is None:
            m_out = None
        else:
            m_out = m_in if m_in is not None else torch.empty_like(grad)
        if variance is not None and not isinstance(variance, torch.Tensor):
            variance = None                                   # the Python 0 of the first VMI iteration
        _hip.mi_update(grad, m_in, m_out, delta.detach(), data, self.decay, self.alpha if alpha is None else alpha,
                       self.epsilon, variance=variance, x_adv=x_adv, data_u8=self._byte_source_of(data), std=grad_std)
        return momentum if m_out is None else m_out

    @staticmethod
    def _byte_source_of(data):
        """(bytes, mismatch flag) attached to ``data`` by ``_attach_byte_source``, if ``data`` still has the probed version"""
        src = getattr(data, "_ta_u8", None)                 # (version of data when probed, bytes, mismatch flag)
        if src is None or src[0] != data._version:
            return None
        if os.environ.get("TA_DEBUG_PARTIALS") == "verify":
            # the cached bytes rest on torch's version counter, which a write through .data / dlpack / a foreign kernel does
            # not move (every writer of THIS package drops the attribute): this debug mode probes again and refuses stale bytes
            again = _hip.u8_source_probe(data)
            if not (torch.equal(again[0], src[1]) and int(again[1].item()) == int(src[2].item())):
                raise _hip.HipExtensionError("TA_DEBUG_PARTIALS=verify: the byte source attached to this image batch is stale "
                                             "(the tensor was modified without torch noticing)")
        return src[1:]

    @staticmethod
    def _attach_byte_source(data):
        """The images of this path are PNG-decoded: ``float(byte) / 255`` (utils.py:136).  One asynchronous pass per batch
        writes the bytes next to ``data`` and a device-side flag saying whether they reproduce it bit for bit; the fused
        update then reads 1 B instead of 4 B per element of ``data`` in each of the K iterations -- or, if the flag says
        otherwise (any other caller of the plug-in API), the fp32 operand: decided inside the kernel, identical results.
        Travels as an attribute of the tensor (as the |g| sums do), valid while ``data`` keeps its version.
        ``TA_U8_SOURCE=0`` turns it off."""
        if (os.environ.get("TA_U8_SOURCE", "1") == "0" or data.dtype != torch.float32 or not data.is_contiguous()
                or data.dim() < 2 or data[0].numel() % 4 or data.data_ptr() % 16):
            return
        data._ta_u8 = (data._version,) + tuple(_hip.u8_source_probe(data))

    # ------------------------------------------------------------------------------------------ hooks
    def get_logits(self, x, **kwargs):
        return self.model(x)

    def get_loss(self, logits, label):
        return -self.loss(logits, label) if self.targeted else self.loss(logits, label)

    def get_grad(self, loss, delta, **kwargs):
        """d(loss)/d(delta) (attack.py:118-122).  Contract for overrides: the kernel that wrote the returned tensor may have
        attached per-tile sums of |g| to it (``tensor._ta_partials``, see ``_hip``), which the fused update uses instead of
        reading g again.  torch in-place operations invalidate them by themselves (version counter); code that changes the
        gradient's memory behind torch's back (a foreign kernel, a collective, a numpy / dlpack alias, ``.data``) must call
        ``_hip.invalidate_partials(grad)`` -- or simply return a new tensor, which carries no sums."""
        return torch.autograd.grad(loss, delta, retain_graph=False, create_graph=False)[0]

    def get_momentum(self, grad, momentum, **kwargs):
        """momentum*decay + grad / mean|grad| per image (attack.py:124-128); ``momentum`` may be 0."""
        grad = grad.contiguous()
        if not isinstance(momentum, torch.Tensor):
            momentum = None if momentum == 0 else torch.full_like(grad, float(momentum))
        out = torch.empty_like(grad)
        _hip.momentum(grad, None if momentum is None else momentum.contiguous(), out, self.decay)
        return out

    def init_delta(self, data, **kwargs):
        """Zeros, or a uniform / scaled-normal random start clamped to the image box (attack.py:130-143)."""
        delta = torch.zeros_like(data).to(self.device)
        if self.random_start:
            if self.norm == 'linfty':
                noise = None
                if self.noise_source is not None:
                    noise = self.noise_source(data.shape, -self.epsilon, self.epsilon).to(self.device).contiguous()
                _hip.init_delta_uniform(delta, data.contiguous(), self.epsilon, self.rng_seed, self._next_offset(),
                                        noise=noise)
            else:
                if self.normal_source is not None:          # test hook: the reference's CPU draws, in its order
                    delta.copy_(self.normal_source(data.shape, -self.epsilon, self.epsilon))
                else:
                    delta.normal_(-self.epsilon, self.epsilon)
                d_flat = delta.view(delta.size(0), -1)
                n = d_flat.norm(p=2, dim=-1).view(delta.size(0), 1, 1, 1)
                if self.noise_source is not None:
                    r = self.noise_source(data.shape, 0, 1).to(self.device)
                else:
                    r = torch.zeros_like(data).uniform_(0, 1).to(self.device)
                delta *= r / n * self.epsilon
                delta = clamp(delta, img_min - data, img_max - data)
        delta.requires_grad = True
        return delta

    def update_delta(self, delta, data, grad, alpha, **kwargs):
        """delta + alpha*sign(grad) projected on the eps-ball and the image box (attack.py:145-153);
        returns a fresh leaf with requires_grad=True.  ``alpha``: float (may be negative) or tensor."""
        src = delta.detach().contiguous()
        out = torch.empty_like(src)
        if self.norm == 'linfty':
            _hip.update_delta_linf(src, data.contiguous(), grad.detach().contiguous(), alpha, self.epsilon, out)
        else:
            if isinstance(alpha, torch.Tensor):
                raise Exception("Unsupported tensor step size for norm l2")
            _hip.update_delta_l2(src, data.contiguous(), grad.detach().contiguous(), alpha, self.epsilon, out)
        return out.requires_grad_(True)

    def loss_function(self, loss):
        if loss == 'crossentropy':
            return nn.CrossEntropyLoss()
        raise Exception("Unsupported loss {}".format(loss))

    def transform(self, data, **kwargs):
        return data

    # ------------------------------------------------------------------------------- helpers for subclasses
    def _schedule(self, alpha, epoch, decay):
        """step size, iteration count and momentum decay -- the three numbers that tell FGSM / I-FGSM / MI-FGSM
        apart (gradient/fgsm.py:31-33, ifgsm.py:33-35, mifgsm.py:34-36)"""
        self.alpha, self.epoch, self.decay = alpha, epoch, decay

    def _to_device(self, data, label):
        """The prologue every ``forward`` of the reference repeats (attack.py:76-80)."""
        if self.targeted:
            assert len(label) == 2
            label = label[1]
        data = data.clone().detach().to(self.device)
        self._attach_byte_source(data)
        return data, label.clone().detach().to(self.device)

    def l1_normalize(self, grad):
        """grad / mean_{CHW}|grad| -- the normalisation inside get_momentum, used on its own by several
        gradient attacks (e.g. gnp.py:74, iefgsm.py:70, emifgsm.py:98); same HIP kernels as get_momentum."""
        grad = grad.contiguous()
        out = torch.empty_like(grad)
        _hip.momentum(grad, None, out, 1.0)
        return out

    def _uniform_like(self, data, radius):
        """Noise tensor for neighbour sampling when a test injects the reference's CPU draws, else None (the HIP
        kernel then draws from its Philox stream)."""
        if self.noise_source is None:
            return None
        return self.noise_source(data.shape, -radius, radius).to(self.device).contiguous()

    def _next_offset(self):
        self.rng_offset += 1
        return self.rng_offset

    def __call__(self, *input, **kwargs):
        self.model.eval()
        return self.forward(*input, **kwargs)
