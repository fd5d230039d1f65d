"""ctypes binding of libta_hip.so (include/ta_hip.h) -- the only door from the Python host code to the
gfx950 kernels.

There is no fallback: if the library is missing, does not export a symbol of the header, or a tensor is
not a contiguous fp32 tensor on a HIP device, this module raises.  (The CPU restatement of the arithmetic
lives under oracle/ and is test infrastructure only; nothing here imports it.)

Every wrapper takes torch tensors, passes raw device pointers plus the *current* torch HIP stream, and
returns nothing (outputs are preallocated by the caller) -- no allocation, no synchronisation, so the
calls are legal inside ``torch.cuda.graph`` capture.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TA_HIP_LIB", os.path.join(_HERE, "lib", "libta_hip.so"))

_i64 = ctypes.c_int64
_int = ctypes.c_int
_f32 = ctypes.c_float
_u64 = ctypes.c_uint64
_vp = ctypes.c_void_p

# name -> (restype, argtypes): mirrors include/ta_hip.h one to one
SIGNATURES = {
    "ta_abi_version": (_int, []),
    "ta_last_error": (ctypes.c_char_p, []),
    "ta_set_sum_order": (_int, [_int]),
    "ta_get_sum_order": (_int, []),
    "ta_timing_begin": (_int, [_int]),
    "ta_timing_end": (_int, [_vp, _int, _vp]),
    "ta_l1_workspace_floats": (_i64, [_i64, _i64]),
    "ta_update_tiles": (_i64, [_i64]),
    "ta_conv_tiles": (_i64, [_int, _int, _int]),
    "ta_dim_bwd_tiles": (_i64, [_int, _int]),
    "ta_abs_sum_partials": (_int, [_vp, _vp, _vp, _i64, _i64, _vp]),
    "ta_momentum": (_int, [_vp, _vp, _vp, _vp, _vp, _f32, _i64, _i64, _vp]),
    "ta_update_delta_linf": (_int, [_vp, _vp, _vp, _f32, _vp, _f32, _vp, _vp, _i64, _vp]),
    "ta_update_delta_l2": (_int, [_vp, _vp, _vp, _f32, _f32, _vp, _vp, _i64, _i64, _vp]),
    "ta_mi_update": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _f32, _f32, _f32, _i64, _i64, _vp]),
    "ta_mi_update_u8": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _f32, _f32, _f32, _i64, _i64, _vp]),
    "ta_u8_source_probe": (_int, [_vp, _vp, _vp, _i64, _vp]),
    "ta_normalize_adv_fwd": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _int, _i64, _vp]),
    "ta_normalize_adv_fwd_nhwc": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _int, _i64, _vp]),
    "ta_mi_update_std": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _f32, _f32, _f32, _i64, _int, _i64, _vp]),
    "ta_abs_sum_partials_std": (_int, [_vp, _vp, _vp, _i64, _int, _i64, _vp]),
    "ta_stem_tiles": (_i64, [_int, _int]),
    "ta_resize_tiles": (_i64, [_int]),
    "ta_resize_normalize_fwd": (_int, [_vp, _vp, _vp, _vp, _i64, _int, _int, _int, _vp]),
    "ta_resize_normalize_bwd": (_int, [_vp, _vp, _vp, _vp, _i64, _int, _int, _int, _vp]),
    "ta_normalize_fwd": (_int, [_vp, _vp, _vp, _vp, _i64, _int, _i64, _vp]),
    "ta_normalize_bwd": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _int, _i64, _vp]),
    "ta_vmi_neighbor_normalized": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _f32, _u64, _u64, _i64, _int, _i64, _vp]),
    "ta_normalize_bwd_accumulate": (_int, [_vp, _vp, _vp, _int, _i64, _int, _i64, _vp]),
    "ta_init_delta_uniform": (_int, [_vp, _vp, _vp, _f32, _u64, _u64, _i64, _vp]),
    "ta_depthwise_conv2d_same": (_int, [_vp, _vp, _vp, _vp, _int, _i64, _int, _int, _vp]),
    "ta_dim_fwd": (_int, [_vp, _vp, _i64, _int, _int, _int, _int, _int, _vp]),
    "ta_dim_bwd": (_int, [_vp, _vp, _vp, _i64, _int, _int, _int, _int, _int, _vp]),
    "ta_stem7s2_prepare": (_int, [_vp, _vp, _vp]),
    "ta_stem7s2_input_grad": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _int, _int, _vp]),
    "ta_stem7s2_input_grad_nchw": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _int, _int, _vp]),
    "ta_bias_act": (_int, [_vp, _vp, _int, _vp, _i64, _int, _i64, _vp]),
    "ta_bias_add_relu": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _int, _i64, _vp]),
    "ta_relu_mask": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "ta_maxpool_bwd_relu": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _int, _int, _int, _int, _int, _int, _int, _int, _vp]),
    "ta_maxpool3s2_fwd": (_int, [_vp, _vp, _vp, _vp, _i64, _int, _int, _int, _vp]),
    "ta_maxpool3s2_bwd_relu": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _int, _int, _int, _vp]),
    "ta_scale_copies_fwd": (_int, [_vp, _vp, _i64, _i64, _int, _vp]),
    "ta_scale_copies_bwd": (_int, [_vp, _vp, _vp, _i64, _i64, _int, _vp]),
    "ta_sum_copies_bwd": (_int, [_vp, _vp, _vp, _i64, _i64, _int, _vp]),
    "ta_admix_fwd": (_int, [_vp, _vp, _vp, _i64, _i64, _int, _int, _f32, _vp]),
    "ta_admix_bwd": (_int, [_vp, _vp, _vp, _i64, _i64, _int, _int, _vp]),
    "ta_sum_members": (_int, [_vp, _int, _vp, _vp, _i64, _i64, _vp]),
    "ta_sia_fwd": (_int, [_vp, _vp, _vp, _vp, _i64, _int, _int, _int, _int, _f32, _u64, _u64, _vp]),
    "ta_sia_bwd": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _int, _int, _int, _int, _f32, _u64, _u64, _vp]),
    "ta_bsr_tiles": (_i64, [_int]),
    "ta_bsr_fwd": (_int, [_vp, _vp, _vp, _i64, _int, _int, _int, _int, _vp]),
    "ta_bsr_bwd": (_int, [_vp, _vp, _vp, _vp, _i64, _int, _int, _int, _int, _vp]),
    "ta_dct_pair": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _int, _vp]),
    "ta_vmi_neighbor": (_int, [_vp, _vp, _vp, _vp, _f32, _u64, _u64, _i64, _vp]),
    "ta_grad_accumulate": (_int, [_vp, _vp, _int, _i64, _vp]),
    "ta_variance_finalize": (_int, [_vp, _vp, _vp, _f32, _i64, _vp]),
    "ta_axpy": (_int, [_vp, _vp, _f32, _vp, _i64, _vp]),
    "ta_quantize_u8_nhwc": (_int, [_vp, _vp, _vp, _i64, _int, _int, _int, _vp]),
}

ABI_VERSION = 13


class HipExtensionError(RuntimeError):
    pass


_lib = None


def load():
    """Load libta_hip.so and bind every symbol of the header; raises HipExtensionError otherwise."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise HipExtensionError(
            "HIP extension %s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C transferattack_amd/csrc).  There is no CPU fallback." % LIB_PATH)
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as exc:
        raise HipExtensionError("cannot load %s: %s" % (LIB_PATH, exc))
    for name, (restype, argtypes) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise HipExtensionError("%s does not export %s (stale build?)" % (LIB_PATH, name))
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.ta_abi_version() != ABI_VERSION:
        raise HipExtensionError("ABI version mismatch: library %d, binding %d" % (lib.ta_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def _ptr(t, dtype=torch.float32, name="tensor"):
    if t is None:
        return None
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise HipExtensionError("%s is on %s: the HIP path needs tensors on a HIP device (no CPU fallback)"
                                % (name, t.device))
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    return t.data_ptr()


def _stream(like=None):
    """The torch HIP stream of the tensor's device (not of whatever device is current: ``torch.cuda.set_device`` is
    thread-local, so an io thread of rank k > 0 would otherwise launch on GPU 0)."""
    return torch.cuda.current_stream(None if like is None else like.device).cuda_stream


def _check(rc, what):
    if rc != 0:
        msg = load().ta_last_error().decode("utf-8", "replace")
        raise HipExtensionError("%s failed (rc=%d): %s" % (what, rc, msg))


def sum_order():
    """0, or the SIMD width (8 / 16) of the CPU whose ATen sum order ``TA_ATEN_SUM_LANES`` asks the kernels to reproduce"""
    value = os.environ.get("TA_ATEN_SUM_LANES", "0")
    return int(value) if value in ("8", "16") else 0


def _sync_options(lib):
    """The library reads no environment: the one process-wide setting it has -- the order of the |g| sums -- is pushed through
    the ABI whenever ``TA_ATEN_SUM_LANES`` differs from what this library object was last told (tests flip it inside one
    process).  One dict lookup per call when nothing changed."""
    want = sum_order()
    if getattr(lib, "_ta_sum_order", 0) != want:
        if lib.ta_set_sum_order(want) != 0:
            raise HipExtensionError("ta_set_sum_order(%d) refused" % want)
        lib._ta_sum_order = want
    return lib


def _call(name, like, *args):
    """lib.<name>(*args, stream) on ``like``'s device and its current torch stream; raises on a non-zero return."""
    fn = getattr(_sync_options(load()), name)
    dev = like.device
    if dev.index is not None and dev.index != torch.cuda.current_device():
        with torch.cuda.device(dev):                    # the launch needs the tensor's device current in THIS thread
            rc = fn(*args, torch.cuda.current_stream(dev).cuda_stream)
    else:
        rc = fn(*args, torch.cuda.current_stream(dev).cuda_stream)
    _check(rc, name)


def _batch(t):
    return t.shape[0], t[0].numel()


# ------------------------------------------------------------------------------------------ workspaces
class Workspace:
    """Per-(device, stream) scratch for the reductions; grown on demand."""

    def __init__(self):
        self._l1 = {}

    def l1(self, like, n, e):
        _ptr(like)
        key = (like.device, _stream(like))
        need = load().ta_l1_workspace_floats(n, e)
        buf = self._l1.get(key)
        if buf is None or buf.numel() < need:
            buf = torch.empty(max(need, 1024), dtype=torch.float32, device=like.device)
            self._l1[key] = buf
        return buf


workspace = Workspace()


# ---- producer-side |g| tile sums ("partials") -----------------------------------------------------------
# The kernel that writes the input gradient LAST (ta_normalize_bwd for the plain attacks; TIM's convolution, the DIM /
# SIM / Admix / EMI backward kernels, the ensemble's member sum otherwise) also leaves per-tile sums of |g|; mi_update
# consumes them -- and skips its own pass over g -- if and only if it is handed that very tensor, unmodified.  The sums
# travel WITH the gradient: the producer's wrapper attaches them to the tensor object it wrote (``tensor._ta_partials``;
# autograd hands the same Python object on -- through ``autograd.grad``, identity Functions and AddBackward alike), and
# there is no module-level state:
#   * a copy (``.contiguous()`` of a strided gradient, ``.clone()``, a collective's fresh result) is a new object without
#     the attribute: its consumer runs its own pass;
#   * torch in-place operations bump ``_version``, which must still be the producer's;
#   * every wrapper of this module that WRITES a tensor drops the attribute of that tensor and of its base, and code that
#     modifies a gradient behind torch's back (c10d collectives, dist.py) calls ``invalidate_partials(tensor)``;
#   * producer and consumer must be on the same stream (the sums are ordered after the producer only there);
#   * the sums are of |g|, or of |g + variance| for exactly the variance tensor the producer was given (VMI-FGSM).
#   * the sums are of |g / std[c]| for exactly the std vector the producer was given when the tensor is the gradient with
#     respect to the NORMALISED input (``mi_update(..., std=...)``: the update divides inline) -- never mixed with plain sums.
_ATTR = "_ta_partials"     # (gradient's _version, ws tensor, sums per image, producer's stream, variance tensor | None, its _version, std key | None)
_SCALE_ATTR = "_ta_grad_scale"   # on a surrogate's INPUT tensor: the std vector its consumer will divide the input gradient by
stats = {"partials_reused": 0, "k1_passes": 0, "u8_source_launches": 0, "std_form_launches": 0}


def _std_key(std):
    return None if std is None else (std.data_ptr(), std.numel(), std._version)


def _register_partials(grad, ws, slots, variance=None, std=None):
    setattr(grad, _ATTR, (grad._version, ws, int(slots), _stream(grad), variance, None if variance is None else variance._version,
                          _std_key(std)))


def partials_of(grad):
    """(ws, sums per image) the producer of ``grad`` attached to it, or None -- read-only (tests, diagnostics)"""
    entry = getattr(grad, _ATTR, None)
    return None if entry is None else (entry[1], entry[2])


def invalidate_partials(*tensors):
    """``tensors`` were modified behind torch's back (a collective, a foreign kernel): their sums -- and, for an image batch,
    the byte source probed from its old contents (``_ta_u8``, attack.py) -- are stale"""
    for t in tensors:
        if isinstance(t, torch.Tensor):
            t.__dict__.pop(_ATTR, None)
            t.__dict__.pop("_ta_u8", None)
            base = t._base
            if base is not None:
                base.__dict__.pop(_ATTR, None)
                base.__dict__.pop("_ta_u8", None)


def _wrote(*tensors):
    """A kernel of this module wrote ``tensors``: sums attached to that memory are stale."""
    invalidate_partials(*tensors)


def _take_partials(grad, variance=None, std=None):
    """the sums the producer attached to exactly this gradient (and, if ``variance`` is given, taken of |grad + variance|
    for exactly that variance tensor; if ``std`` is given, of |grad / std[c]| for exactly that std vector); the attribute
    is consumed either way"""
    entry = grad.__dict__.pop(_ATTR, None)
    if entry is None or sum_order() != 0:
        return None                                   # the reference-order sum is never taken from a producer
    version, ws, slots, stream, var, var_version, std_key = entry
    same_variance = (var is None and variance is None) or (
        var is not None and variance is not None and var.data_ptr() == variance.data_ptr() and var.shape == variance.shape
        and variance._version == var_version and var._version == var_version)
    same_variance = same_variance and std_key == _std_key(std)
    if same_variance and grad._version == version and ws.device == grad.device and stream == _stream(grad):
        if os.environ.get("TA_DEBUG_PARTIALS") == "verify":
            _verify_partials(grad, variance, ws, slots, std)
        return ws, slots
    if os.environ.get("TA_DEBUG_PARTIALS"):
        print("partials not reused: version %d vs %d, stream %s vs %s, variance / std match %s" % (
            grad._version, version, stream, _stream(grad), same_variance), flush=True)
    return None


def _verify_partials(grad, variance, ws, slots, std=None):
    """``TA_DEBUG_PARTIALS=verify``: the validity of the attached sums rests on object identity + ``_version``, which a write
    through ``.data``, a dlpack / numpy alias or a foreign kernel does not move.  This debug mode recomputes sum|g (+ v)| per
    image (fp64, one synchronising pass) and refuses sums that are not the gradient's -- run a new attack class under it once."""
    n = grad.shape[0]
    g = grad.detach().double()
    if std is not None:
        g = g / std.detach().double().reshape((1, -1) + (1,) * (g.dim() - 2))
    if variance is not None:
        g = g + variance.detach().double()
    want = g.abs().reshape(n, -1).sum(1)
    got = ws[:n * slots].double().reshape(n, slots).sum(1)
    bad = ~((got - want).abs() <= 1e-4 * want.abs() + 1e-30)          # NaN-safe: a NaN sum must meet a NaN sum
    bad &= ~(torch.isnan(got) & torch.isnan(want))
    if bool(bad.any()):
        i = int(bad.nonzero()[0])
        raise HipExtensionError("TA_DEBUG_PARTIALS=verify: the |g| sums attached to this gradient are stale (image %d: attached "
                                "%.9g, recomputed %.9g) -- the tensor was modified without torch noticing; call "
                                "_hip.invalidate_partials(grad) after such a write" % (i, float(got[i]), float(want[i])))


def _new_ws(like, count):
    return torch.empty(max(int(count), 1), dtype=torch.float32, device=like.device)


# ------------------------------------------------------------------------------------------- update stack
def momentum(grad, momentum_in, momentum_out, decay, variance=None):
    n, e = _batch(grad)
    ws = workspace.l1(grad, n, e)
    _wrote(momentum_out)
    _call("ta_momentum", grad, _ptr(grad, name="grad"), _ptr(variance, name="variance"),
          _ptr(momentum_in, name="momentum"), _ptr(momentum_out, name="momentum_out"), _ptr(ws), decay, n, e)


def update_delta_linf(delta_in, data, momentum_, alpha, epsilon, delta_out, x_adv=None):
    alpha_t = alpha if isinstance(alpha, torch.Tensor) else None
    if alpha_t is not None and alpha_t.shape != delta_in.shape:
        alpha_t = alpha_t.expand_as(delta_in).contiguous()
    _wrote(delta_out, x_adv)
    _call("ta_update_delta_linf", delta_in, _ptr(delta_in, name="delta"), _ptr(data, name="data"),
          _ptr(momentum_, name="grad"), 0.0 if alpha_t is not None else float(alpha), _ptr(alpha_t, name="alpha"),
          float(epsilon), _ptr(delta_out, name="delta_out"), _ptr(x_adv, name="x_adv"), delta_in.numel())


def update_delta_l2(delta_in, data, grad, alpha, epsilon, delta_out):
    n, e = _batch(delta_in)
    ws = workspace.l1(delta_in, n, e)
    _wrote(delta_out)
    _call("ta_update_delta_l2", delta_in, _ptr(delta_in, name="delta"), _ptr(data, name="data"), _ptr(grad, name="grad"),
          float(alpha), float(epsilon), _ptr(delta_out, name="delta_out"), _ptr(ws), n, e)


# bench.py sets this to a list to time every fused-update launch with HIP events on the launch stream: marker events
# (hipEventRecord before / after the call) recorded here, and -- between timing_begin() and timing_end() -- events bound
# to the dispatch packets of the call's own kernels (ta_timing_begin, include/ta_hip.h)
profile_sink = None
_timing_capacity = 0


def timing_begin(capacity):
    global _timing_capacity
    rc = load().ta_timing_begin(int(capacity))
    if rc != 0:
        raise HipExtensionError("ta_timing_begin: %s" % load().ta_last_error().decode())
    _timing_capacity = int(capacity)


def timing_end():
    """Milliseconds (first kernel's begin -> update kernel's end) of every fused update since timing_begin, in call order."""
    global _timing_capacity
    buf = (ctypes.c_float * max(_timing_capacity, 1))()
    count = ctypes.c_int(0)
    rc = load().ta_timing_end(ctypes.cast(buf, _vp), _timing_capacity, ctypes.cast(ctypes.pointer(count), _vp))
    _timing_capacity = 0
    if rc != 0:
        raise HipExtensionError("ta_timing_end: %s" % load().ta_last_error().decode())
    return [float(buf[i]) for i in range(count.value)]


def u8_source_probe(data):
    """(bytes, mismatch flag) of an image batch for ``mi_update(..., data_u8=...)``: ``bytes = round(data * 255)`` (uint8,
    data's shape) and a device int that the kernel sets to 1 unless EVERY element of ``data`` is float(byte) / 255 bit for
    bit -- true for PNG-decoded images (utils.py:136).  Asynchronous; nobody on the host reads the flag."""
    if not data.is_contiguous():
        raise ValueError("data must be contiguous")
    u8 = torch.empty(data.shape, dtype=torch.uint8, device=data.device)
    flag = torch.empty(1, dtype=torch.int32, device=data.device)
    _call("ta_u8_source_probe", data, _ptr(data, name="data"), _ptr(u8, torch.uint8, name="data_u8"),
          _ptr(flag, torch.int32, name="mismatch"), data.numel())
    return u8, flag


def mi_update(grad, momentum_in, momentum_out, delta, data, decay, alpha, epsilon, variance=None, x_adv=None, data_u8=None,
              std=None):
    """Fused get_momentum + update_delta; ``delta`` is updated in place, momentum_out may alias momentum_in.
    ``momentum_in`` None = first iteration; ``momentum_out`` None = the momentum is not kept (decay == 0);
    ``x_adv`` (optional) receives data + delta', the next iteration's input; ``data_u8`` (optional) is
    ``u8_source_probe(data)``: the kernel then reads one byte instead of four per element of ``data`` whenever the probe
    found the batch byte-valued.  ``std`` (optional, fp32 [C]): ``grad`` is the gradient with respect to the NORMALISED
    input -- the backbone's own output -- and the kernel divides it by std[c] inline (Normalize's backward folded in:
    ``ta_mi_update_std``; no variance term, no ``x_adv`` in that form)."""
    n, e = _batch(grad)
    if profile_sink is not None:
        dev = grad.device
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record(torch.cuda.current_stream(dev))
        k1_before = stats["k1_passes"]
        _mi_update(grad, momentum_in, momentum_out, delta, data, decay, alpha, epsilon, variance, x_adv, n, e, data_u8, std)
        end.record(torch.cuda.current_stream(dev))
        bytes_per_elem = 4 * (3 + (variance is not None) + (momentum_in is not None) + 1 + (momentum_out is not None)
                              + (x_adv is not None))       # r g,(v),(m),d,x  w (m),d,(x_adv): the ALGORITHMIC bytes
        # (..., byte source offered, std form, did THIS launch run its own sum pass over the gradient)
        profile_sink.append((start, end, n, e, bytes_per_elem, data_u8 is not None, std is not None,
                             stats["k1_passes"] > k1_before))
        return
    _mi_update(grad, momentum_in, momentum_out, delta, data, decay, alpha, epsilon, variance, x_adv, n, e, data_u8, std)


def _mi_update(grad, momentum_in, momentum_out, delta, data, decay, alpha, epsilon, variance, x_adv, n, e, data_u8=None, std=None):
    ready = _take_partials(grad, variance, std)
    stats["partials_reused" if ready is not None else "k1_passes"] += 1
    ws, slots = ready if ready is not None else (workspace.l1(grad, n, e), 0)
    if std is not None:
        if variance is not None or x_adv is not None:
            raise ValueError("the std form of the fused update takes neither a variance term nor an x_adv buffer")
        c = grad.shape[1]
        if std.numel() != c or grad.dim() < 2:
            raise ValueError("std has %d entries for a gradient with %d channels" % (std.numel(), c))
        u8, flag = data_u8 if data_u8 is not None else (None, None)
        if u8 is not None and (u8.shape != data.shape or u8.device != data.device):
            raise ValueError("data_u8 does not belong to data")
        stats["std_form_launches"] += 1
        stats["u8_source_launches"] += u8 is not None
        _call("ta_mi_update_std", grad, _ptr(grad, name="grad"), _ptr(std, name="std"), _ptr(momentum_in, name="momentum"),
              _ptr(momentum_out, name="momentum_out"), _ptr(delta, name="delta"), _ptr(data, name="data"),
              _ptr(u8, torch.uint8, name="data_u8"), _ptr(flag, torch.int32, name="mismatch"), _ptr(ws), slots, float(decay),
              float(alpha), float(epsilon), n, c, e // c)
        return
    if data_u8 is not None:
        u8, flag = data_u8
        if u8.shape != data.shape or u8.device != data.device:
            raise ValueError("data_u8 does not belong to data")
        stats["u8_source_launches"] += 1
        _call("ta_mi_update_u8", grad, _ptr(grad, name="grad"), _ptr(variance, name="variance"),
              _ptr(momentum_in, name="momentum"), _ptr(momentum_out, name="momentum_out"), _ptr(delta, name="delta"),
              _ptr(data, name="data"), _ptr(u8, torch.uint8, name="data_u8"), _ptr(flag, torch.int32, name="mismatch"),
              _ptr(x_adv, name="x_adv"), _ptr(ws), slots, float(decay), float(alpha), float(epsilon), n, e)
        return
    _call("ta_mi_update", grad, _ptr(grad, name="grad"), _ptr(variance, name="variance"),
          _ptr(momentum_in, name="momentum"), _ptr(momentum_out, name="momentum_out"), _ptr(delta, name="delta"),
          _ptr(data, name="data"), _ptr(x_adv, name="x_adv"), _ptr(ws), slots, float(decay), float(alpha),
          float(epsilon), n, e)


def abs_sum_partials(grad, variance=None):
    """K1 alone: (ws, sums per image) for ``grad`` (+ variance), registered as the partials of ``grad``."""
    n, e = _batch(grad)
    slots = load().ta_update_tiles(e)
    ws = _new_ws(grad, n * slots)
    _call("ta_abs_sum_partials", grad, _ptr(grad, name="grad"), _ptr(variance, name="variance"), _ptr(ws), n, e)
    if variance is None:
        _register_partials(grad, ws, slots)
    return ws, slots


def abs_sum_partials_std(grad, std):
    """K1 over grad / std[c]: (ws, sums per image), registered as the partials of ``grad`` for ``mi_update(..., std=std)``"""
    n, e = _batch(grad)
    c = grad.shape[1]
    slots = load().ta_update_tiles(e)
    ws = _new_ws(grad, n * slots)
    _call("ta_abs_sum_partials_std", grad, _ptr(grad, name="grad"), _ptr(std, name="std"), _ptr(ws), n, c, e // c)
    _register_partials(grad, ws, slots, std=std)
    return ws, slots


def normalize_adv_fwd(data, delta, y, mean, std, data_u8=None):
    """y = ((data + delta) - mean[c]) / std[c]: the add of attack.py:88 and PreprocessingModel's Normalize in one pass;
    ``data_u8`` = ``u8_source_probe(data)``: the image is read as bytes when the probe found the batch byte-valued"""
    n, c = data.shape[0], data.shape[1]
    u8, flag = data_u8 if data_u8 is not None else (None, None)
    if u8 is not None and (u8.shape != data.shape or u8.device != data.device):
        raise ValueError("data_u8 does not belong to data")
    if mean.numel() != c or std.numel() != c:
        raise ValueError("mean / std have %d / %d entries for %d channels" % (mean.numel(), std.numel(), c))
    _wrote(y)
    if not y.is_contiguous() and y.is_contiguous(memory_format=torch.channels_last):
        # y in NHWC memory for a channels_last surrogate (its first convolution reads it without a layout copy)
        _call("ta_normalize_adv_fwd_nhwc", data, _ptr(data, name="data"), _ptr(u8, torch.uint8, name="data_u8"),
              _ptr(flag, torch.int32, name="mismatch"), _ptr(delta, name="delta"), _ptr_any(y, "y"), _ptr(mean, name="mean"),
              _ptr(std, name="std"), n, c, data[0, 0].numel())
        return
    _call("ta_normalize_adv_fwd", data, _ptr(data, name="data"), _ptr(u8, torch.uint8, name="data_u8"),
          _ptr(flag, torch.int32, name="mismatch"), _ptr(delta, name="delta"), _ptr(y, name="y"), _ptr(mean, name="mean"),
          _ptr(std, name="std"), n, c, data[0, 0].numel())


def normalize_fwd(x, y, mean, std):
    n, c = x.shape[0], x.shape[1]
    _wrote(y)
    _call("ta_normalize_fwd", x, _ptr(x, name="x"), _ptr(y, name="y"), _ptr(mean, name="mean"), _ptr(std, name="std"),
          n, c, x[0, 0].numel())


def normalize_bwd(gy, gx, std, variance=None):
    """gx = gy / std[c]; also registers the |gx| tile sums (|gx + variance| when a variance tensor is given: VMI-FGSM) for
    the fused update that consumes gx next."""
    n, c = gy.shape[0], gy.shape[1]
    slots = load().ta_update_tiles(gy[0].numel())
    ws = _new_ws(gy, n * slots)
    _call("ta_normalize_bwd", gy, _ptr(gy, name="gy"), _ptr(gx, name="gx"), _ptr(std, name="std"),
          _ptr(variance, name="variance"), _ptr(ws), n, c, gy[0, 0].numel())
    _register_partials(gx, ws, slots, variance)


def resize_normalize_fwd(x, y, mean, std):
    """y[n, c, out, out] = (bilinear(x[n, c, in, in]) - mean[c]) / std[c] -- PreprocessingModel with a Resize (Inception-v3)"""
    n, c, in_size, out_size = x.shape[0], x.shape[1], x.shape[-1], y.shape[-1]
    _wrote(y)
    _call("ta_resize_normalize_fwd", x, _ptr(x, name="x"), _ptr(y, name="y"), _ptr(mean, name="mean"), _ptr(std, name="std"),
          n, c, in_size, out_size)


def resize_normalize_bwd(gy, gx, std):
    """gx[n, c, in, in] = upsample_bilinear2d_backward(gy[n, c, out, out] / std[c]); registers the |gx| tile sums"""
    n, c, in_size, out_size = gx.shape[0], gx.shape[1], gx.shape[-1], gy.shape[-1]
    tiles = load().ta_resize_tiles(in_size)
    ws = _new_ws(gy, n * c * tiles)
    _call("ta_resize_normalize_bwd", gy, _ptr(gy, name="gy"), _ptr(gx, name="gx"), _ptr(std, name="std"), _ptr(ws), n, c,
          in_size, out_size)
    _register_partials(gx, ws, c * tiles)


def init_delta_uniform(delta, data, epsilon, seed=0, offset=0, noise=None):
    _wrote(delta)
    _call("ta_init_delta_uniform", delta, _ptr(delta, name="delta"), _ptr(data, name="data"), _ptr(noise, name="noise"),
          float(epsilon), seed, offset, delta.numel())


# --------------------------------------------------------------------------------------------- transforms
def depthwise_conv2d_same(inp, out, weight2d):
    """out = depthwise k x k 'same' correlation of inp; registers the |out| tile sums (TIM.get_grad produces the
    gradient the update consumes)."""
    k = weight2d.shape[-1]
    h, w = inp.shape[-2:]
    planes = inp.numel() // (h * w)
    per_image = inp[0].numel() // (h * w) if inp.dim() == 4 else 0
    tiles = load().ta_conv_tiles(k, h, w)
    ws = _new_ws(inp, planes * tiles) if per_image else None
    _call("ta_depthwise_conv2d_same", inp, _ptr(inp, name="grad"), _ptr(out, name="out"), _ptr(weight2d, name="kernel"),
          _ptr(ws), k, planes, h, w)
    if ws is not None:
        _register_partials(out, ws, per_image * tiles)
    else:
        _wrote(out)


def dim_fwd(x, y, resize, rnd, top, left):
    size = x.shape[-1]
    _wrote(y)
    _call("ta_dim_fwd", x, _ptr(x, name="x"), _ptr(y, name="y"), x.numel() // (size * size), size, resize, rnd, top, left)


def dim_bwd(gy, gx, resize, rnd, top, left):
    size = gy.shape[-1]
    planes = gy.numel() // (size * size)
    per_image = gy[0].numel() // (size * size) if gy.dim() == 4 else 0
    tiles = load().ta_dim_bwd_tiles(size, resize)
    ws = _new_ws(gy, planes * tiles) if per_image else None
    _call("ta_dim_bwd", gy, _ptr(gy, name="gy"), _ptr(gx, name="gx"), _ptr(ws), planes, size, resize, rnd, top, left)
    if ws is not None:
        _register_partials(gx, ws, per_image * tiles)
    else:
        _wrote(gx)


def scale_copies_fwd(x, y, num_scale):
    n, e = _batch(x)
    _wrote(y)
    _call("ta_scale_copies_fwd", x, _ptr(x, name="x"), _ptr(y, name="y"), n, e, num_scale)


def _image_sums(like, n, e):
    slots = load().ta_update_tiles(e)
    return _new_ws(like, n * slots), slots


def scale_copies_bwd(gy, gx, num_scale):
    n, e = _batch(gx)
    ws, slots = _image_sums(gx, n, e)
    _call("ta_scale_copies_bwd", gy, _ptr(gy, name="gy"), _ptr(gx, name="gx"), _ptr(ws), n, e, num_scale)
    _register_partials(gx, ws, slots)


def sum_copies_bwd(gy, gx, copies):
    n, e = _batch(gx)
    ws, slots = _image_sums(gx, n, e)
    _call("ta_sum_copies_bwd", gy, _ptr(gy, name="gy"), _ptr(gx, name="gx"), _ptr(ws), n, e, copies)
    _register_partials(gx, ws, slots)


def admix_fwd(x, perm, y, num_admix, num_scale, strength):
    n, e = _batch(x)
    _wrote(y)
    _call("ta_admix_fwd", x, _ptr(x, name="x"), _ptr(perm, torch.int64, "perm"), _ptr(y, name="y"), n, e, num_admix,
          num_scale, float(strength))


def admix_bwd(gy, gx, num_admix, num_scale):
    n, e = _batch(gx)
    ws, slots = _image_sums(gx, n, e)
    _call("ta_admix_bwd", gy, _ptr(gy, name="gy"), _ptr(gx, name="gx"), _ptr(ws), n, e, num_admix, num_scale)
    _register_partials(gx, ws, slots)


def sum_members(grads, gx):
    """gx = ((g[m-1] + g[m-2]) + ...) + g[0] -- the members' input gradients of an EnsembleModel, in autograd's
    accumulation order; registers the |gx| tile sums."""
    grads = list(grads)
    while len(grads) > 8:                 # the entry takes eight operands: fold the LAST eight (they are added first) into one
        head, tail = grads[:-8], grads[-8:]
        partial = torch.empty_like(gx)
        sum_members(tail, partial)
        grads = head + [partial]
    m = len(grads)
    n, e = _batch(gx)
    ws, slots = _image_sums(gx, n, e)
    ptrs = (ctypes.c_void_p * m)(*[_ptr(g, name="member gradient") for g in grads])
    _call("ta_sum_members", gx, ptrs, m, _ptr(gx, name="gx"), _ptr(ws), n, e)
    _register_partials(gx, ws, slots)


def sia_fwd(x, plan, y, copies, num_block, noise_radius, seed=0, offset=0, noise=None):
    h, w = x.shape[-2:]
    _wrote(y)
    _call("ta_sia_fwd", x, _ptr(x, name="x"), _ptr(plan, torch.int32, "plan"), _ptr(noise, name="noise"), _ptr(y, name="y"),
          x.numel() // (h * w), h, w, copies, num_block, float(noise_radius), seed, offset)


def sia_bwd(gy, plan, x, gx, copies, num_block, noise_radius, seed=0, offset=0, noise=None):
    h, w = x.shape[-2:]
    _wrote(gx)
    _call("ta_sia_bwd", gy, _ptr(gy, name="gy"), _ptr(plan, torch.int32, "plan"), _ptr(x, name="x"),
          _ptr(noise, name="noise"), _ptr(gx, name="gx"), x.numel() // (h * w), h, w, copies, num_block,
          float(noise_radius), seed, offset)


def bsr_fwd(x, plan, y, copies, num_block):
    h, w = x.shape[-2:]
    _wrote(y)
    _call("ta_bsr_fwd", x, _ptr(x, name="x"), _ptr(plan, torch.int32, "plan"), _ptr(y, name="y"), x.numel() // (h * w), h, w,
          copies, num_block)


def bsr_bwd(gy, plan, gx, copies, num_block):
    h, w = gx.shape[-2:]
    planes = gx.numel() // (h * w)
    per_image = gx[0].numel() // (h * w) if gx.dim() == 4 else 0
    tiles = load().ta_bsr_tiles(h)
    ws = _new_ws(gx, planes * tiles) if per_image else None
    _call("ta_bsr_bwd", gy, _ptr(gy, name="gy"), _ptr(plan, torch.int32, "plan"), _ptr(gx, name="gx"), _ptr(ws), planes, h, w,
          copies, num_block)
    if ws is not None:
        _register_partials(gx, ws, per_image * tiles)
    else:
        _wrote(gx)


def dct_pair(inp, add, mul, out, lmat, rmat):
    """out = (L . (inp + add) . R^T) * mul on every n x n plane (add / mul may be None)"""
    n = inp.shape[-1]
    _wrote(out)
    _call("ta_dct_pair", inp, _ptr(inp, name="in"), _ptr(add, name="add"), _ptr(mul, name="mul"), _ptr(out, name="out"),
          _ptr(lmat, name="L"), _ptr(rmat, name="R"), inp.numel() // (n * n), n)


# ---------------------------------------------------------------------------------------------- VMI / NI
def vmi_neighbor(data, delta, out, radius, seed=0, offset=0, noise=None):
    _wrote(out)
    _call("ta_vmi_neighbor", data, _ptr(data, name="data"), _ptr(delta, name="delta"), _ptr(noise, name="noise"),
          _ptr(out, name="out"), float(radius), seed, offset, data.numel())


def vmi_neighbor_normalized(data, delta, out, mean, std, radius, seed=0, offset=0, noise=None):
    """out = (((data + delta) + U(-radius, radius)) - mean[c]) / std[c]: the neighbour sample, already normalised for the
    backbone (``ta_vmi_neighbor`` + ``ta_normalize_fwd`` in one pass, same bits)."""
    n, c = data.shape[0], data.shape[1]
    _wrote(out)
    _call("ta_vmi_neighbor_normalized", data, _ptr(data, name="data"), _ptr(delta, name="delta"), _ptr(noise, name="noise"),
          _ptr(out, name="out"), _ptr(mean, name="mean"), _ptr(std, name="std"), float(radius), seed, offset, n, c,
          data[0, 0].numel())


def normalize_bwd_accumulate(gy, acc, std, first):
    """acc (+)= gy / std[c]: the backward of Normalize and VMI's gradient accumulation in one pass"""
    n, c = gy.shape[0], gy.shape[1]
    _wrote(acc)
    _call("ta_normalize_bwd_accumulate", gy, _ptr(gy, name="gy"), _ptr(acc, name="acc"), _ptr(std, name="std"),
          1 if first else 0, n, c, gy[0, 0].numel())


def grad_accumulate(acc, grad, first):
    _wrote(acc)
    _call("ta_grad_accumulate", acc, _ptr(acc, name="acc"), _ptr(grad, name="grad"), 1 if first else 0, acc.numel())


def variance_finalize(acc, cur_grad, out, count):
    _wrote(out)
    _call("ta_variance_finalize", acc, _ptr(acc, name="acc"), _ptr(cur_grad, name="grad"), _ptr(out, name="variance"),
          float(count), acc.numel())


def axpy(x, m, coeff, out):
    _wrote(out)
    _call("ta_axpy", x, _ptr(x, name="x"), _ptr(m, name="momentum"), float(coeff), _ptr(out, name="out"), x.numel())


# ------------------------------------------------------------------------------------------- surrogate glue
def _glue_layout(t):
    """(channels, inner) of a dense 4-D activation: inner = 1 for channels_last memory, H*W for contiguous NCHW"""
    n, c, h, w = t.shape
    if t.is_contiguous():
        return c, h * w
    if t.is_contiguous(memory_format=torch.channels_last):
        return c, 1
    raise ValueError("activation must be dense NCHW or channels_last, got strides %s" % (t.stride(),))


def pass_bits_like(y):
    """an (uninitialised) buffer for the pass bits of activation ``y``: one bit per element, or None if y.numel() % 8"""
    if y.numel() % 8:
        return None
    return torch.empty(y.numel() // 8, dtype=torch.uint8, device=y.device)


def bias_act_(y, bias, relu=True, mask=None):
    """y <- clamp_min(y + bias[c], 0) (or just the bias add) in place: one pass where ATen makes two.  ``mask`` (optional,
    ``pass_bits_like(y)``) receives one bit per element: does threshold_backward let the gradient pass there"""
    c, inner = _glue_layout(y)
    _wrote(y)
    _call("ta_bias_act", y, _ptr_any(y, "y"), _ptr(bias, name="bias"), 1 if relu else 0, _ptr(mask, torch.uint8, "mask"),
          y.numel(), c, inner)
    return y


def bias_add_relu_(y, bias, other, bias_other=None, mask=None):
    """y <- clamp_min((y + bias[c]) + (other [+ bias_other[c]]), 0) in place: a bottleneck's third convolution, its shortcut
    and the ReLU in one pass; ``mask`` as in ``bias_act_``"""
    c, inner = _glue_layout(y)
    if _glue_layout(other) != (c, inner) or other.shape != y.shape:
        raise ValueError("shortcut and main branch differ in shape or memory format")
    _wrote(y)
    _call("ta_bias_add_relu", y, _ptr_any(y, "y"), _ptr(bias, name="bias"), _ptr_any(other, "other"),
          _ptr(bias_other, name="bias_other"), _ptr(mask, torch.uint8, "mask"), y.numel(), c, inner)
    return y


def relu_mask(ga, y, out, gb=None, mask=None):
    """out <- threshold_backward(ga [+ gb], y, 0); ``out`` may be ``ga``.  All operands share one dense layout.  With ``mask``
    (the pass bits the forward kernel left for y) the activation itself is not read: 1 bit instead of 4 bytes per element"""
    layout = _glue_layout(y)
    for t in (ga, gb, out):
        if t is not None and (t.shape != y.shape or _glue_layout(t) != layout):
            raise ValueError("operands differ in shape or memory format")
    if mask is not None and (mask.numel() * 8 != y.numel() or mask.device != y.device):
        raise ValueError("pass bits do not belong to this activation")
    _wrote(out)
    _call("ta_relu_mask", y, _ptr_any(ga, "ga"), None if gb is None else _ptr_any(gb, "gb"),
          None if mask is not None else _ptr_any(y, "y"), _ptr(mask, torch.uint8, "mask"), _ptr_any(out, "out"), y.numel())
    return out


def stem7s2_prepare(weight):
    """the stem convolution's weights [64, 3, 7, 7] rearranged for ``stem7s2_input_grad`` (once per model)"""
    if tuple(weight.shape) != (64, 3, 7, 7) or weight.dtype != torch.float32:
        raise ValueError("stem weights must be fp32 [64, 3, 7, 7], got %s" % (tuple(weight.shape),))
    w = weight.detach().contiguous()
    w2 = torch.empty(16 * 16 * 4 * 16, dtype=torch.float32, device=w.device)
    _call("ta_stem7s2_prepare", w, _ptr(w, name="weight"), _ptr(w2, name="w2"))
    return w2


def stem7s2_input_grad(dy, w2, dx, std=None):
    """dx [n, 3, 2*oh, 2*ow] (NCHW) = d/d(input) of the 7x7 / stride 2 / padding 3 stem convolution for the output gradient
    ``dy`` [n, 64, oh, ow] in channels_last memory or -- the plain module path of an NCHW surrogate -- in NCHW memory.  ``std`` (fp32
    [3], optional): the consumer will divide dx by std[c] (``mi_update(..., std=std)``) -- the kernel then also leaves the sums of
    |dx / std[c]| per workgroup and they are attached to ``dx`` as its partials, so the update reads dx exactly once and nothing
    else does."""
    n, k, oh, ow = dy.shape
    nchw = dy.is_contiguous()
    if k != 64 or not (nchw or dy.is_contiguous(memory_format=torch.channels_last)) or tuple(dx.shape) != (n, 3, 2 * oh, 2 * ow):
        raise ValueError("stem7s2_input_grad: dy must be dense [n, 64, oh, ow] (NCHW or channels_last) and dx [n, 3, 2*oh, 2*ow]")
    entry = "ta_stem7s2_input_grad_nchw" if nchw else "ta_stem7s2_input_grad"
    p_dy = _ptr(dy, name="dy") if nchw else _ptr_any(dy, "dy")
    _wrote(dx)
    if std is not None and std.numel() == 3 and std.dtype == torch.float32 and std.device == dx.device and std.is_contiguous():
        slots = load().ta_stem_tiles(oh, ow)
        ws = _new_ws(dx, n * slots)
        _call(entry, dy, p_dy, _ptr(w2, name="w2"), _ptr(dx, name="dx"), _ptr(std, name="std"), _ptr(ws), n, oh, ow)
        _register_partials(dx, ws, slots, std=std)
        return dx
    _call(entry, dy, p_dy, _ptr(w2, name="w2"), _ptr(dx, name="dx"), None, None, n, oh, ow)
    return dx


def maxpool_bwd_relu(ga, idx, y, out, kernel, stride, padding, gb=None):
    """out <- threshold_backward(max_pool2d_with_indices_backward(ga [+ gb], idx), y, 0): the backward of ReLU -> max-pool in one
    gather pass.  channels_last tensors: ga / gb / idx [n, c, ph, pw], y / out [n, c, h, w]."""
    n, c, h, w = y.shape
    ph, pw = ga.shape[-2:]
    cl = torch.channels_last
    ok = (y.is_contiguous(memory_format=cl) and out.is_contiguous(memory_format=cl) and ga.is_contiguous(memory_format=cl)
          and idx.is_contiguous(memory_format=cl) and (gb is None or gb.is_contiguous(memory_format=cl))
          and idx.dtype == torch.int64 and tuple(idx.shape) == tuple(ga.shape) and out.shape == y.shape and c % 4 == 0)
    if not ok:
        raise ValueError("maxpool_bwd_relu needs channels_last operands with channels % 4 == 0")
    _wrote(out)
    flat = lambda t, dt: _ptr(t.detach().as_strided((t.numel(),), (1,)), dt, "operand")       # noqa: E731
    _call("ta_maxpool_bwd_relu", y, flat(ga, torch.float32), None if gb is None else flat(gb, torch.float32),
          flat(idx, torch.int64), flat(y, torch.float32), flat(out, torch.float32), n, c, h, w, ph, pw, kernel, stride, padding)
    return out


def maxpool3s2_takes(y, pool):
    """is ``pool`` (a MaxPool2d) over the channels_last activation ``y`` the stem pool ``maxpool3s2_fwd`` implements"""
    pair = lambda v: (v, v) if isinstance(v, int) else tuple(v)                                  # noqa: E731
    return (pair(pool.kernel_size) == (3, 3) and pair(pool.stride) == (2, 2) and pair(pool.padding) == (1, 1)
            and pair(pool.dilation) == (1, 1) and not pool.ceil_mode and y.dim() == 4 and y.dtype == torch.float32
            and y.shape[1] % 8 == 0 and y.shape[2] % 2 == 0 and y.shape[3] % 2 == 0 and y.shape[2] >= 2 and y.shape[3] >= 2
            and y.is_contiguous(memory_format=torch.channels_last) and not y.is_contiguous() and y.numel() < 2 ** 32 - 2 ** 12)


def maxpool3s2_fwd(y):
    """(pooled, arg, mask) of the 3 x 3 / stride 2 / padding 1 max-pool over the channels_last activation ``y`` [n, c, h, w]:
    ``pooled`` = F.max_pool2d(y, 3, 2, 1) (channels_last), ``arg`` [n, c, h/2, w/2] uint8 (channels_last) the winning tap
    kh * 3 + kw of each window -- the element max_pool2d_with_indices names -- and ``mask`` the pass bits of ``y`` itself;
    what ``maxpool3s2_bwd_relu`` needs, 1 1/8 byte per element where (int64 indices, y) are 12"""
    n, c, h, w = y.shape
    cl = torch.channels_last
    pooled = torch.empty((n, c, h // 2, w // 2), dtype=torch.float32, device=y.device, memory_format=cl)
    arg = torch.empty((n, c, h // 2, w // 2), dtype=torch.uint8, device=y.device, memory_format=cl)
    mask = torch.empty(y.numel() // 8, dtype=torch.uint8, device=y.device)
    flat = lambda t, dt: _ptr(t.detach().as_strided((t.numel(),), (1,)), dt, "operand")       # noqa: E731
    _call("ta_maxpool3s2_fwd", y, flat(y, torch.float32), flat(pooled, torch.float32), flat(arg, torch.uint8), _ptr(mask, torch.uint8, "mask"),
          n, c, h, w)
    return pooled, arg, mask


def maxpool3s2_bwd_relu(ga, arg, mask, out, gb=None):
    """out <- threshold_backward(max_pool2d_with_indices_backward(ga [+ gb], .), y, 0) from what ``maxpool3s2_fwd(y)`` left"""
    n, c, h, w = out.shape
    cl = torch.channels_last
    ok = (all(t.is_contiguous(memory_format=cl) for t in (ga, arg, out) + (() if gb is None else (gb,)))
          and tuple(ga.shape) == tuple(arg.shape) == (n, c, h // 2, w // 2) and (gb is None or gb.shape == ga.shape)
          and arg.dtype == torch.uint8 and mask.dtype == torch.uint8 and mask.numel() * 8 == out.numel() and c % 8 == 0
          and h % 2 == 0 and w % 2 == 0)
    if not ok:
        raise ValueError("maxpool3s2_bwd_relu: operands do not belong to one maxpool3s2_fwd call")
    _wrote(out)
    flat = lambda t, dt: _ptr(t.detach().as_strided((t.numel(),), (1,)), dt, "operand")       # noqa: E731
    _call("ta_maxpool3s2_bwd_relu", out, flat(ga, torch.float32), None if gb is None else flat(gb, torch.float32),
          flat(arg, torch.uint8), _ptr(mask, torch.uint8, "mask"), flat(out, torch.float32), n, c, h, w)
    return out


def _ptr_any(t, name):
    """device pointer of a dense fp32 tensor in either memory format (``_ptr`` insists on NCHW-contiguous)"""
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if t.dtype != torch.float32:
        raise TypeError("%s must be float32" % name)
    _glue_layout(t)
    return _ptr(t.detach().as_strided((t.numel(),), (1,)), name=name)


# ------------------------------------------------------------------------------------------------- output
def quantize_u8_nhwc(data, delta, out):
    n, c, h, w = data.shape
    _call("ta_quantize_u8_nhwc", data, _ptr(data, name="data"), _ptr(delta, name="delta"), _ptr(out, torch.uint8, "out"),
          n, c, h, w)
