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
    "ta_l1_workspace_floats": (_i64, [_i64, _i64]),
    "ta_abs_sum_partials": (_int, [_vp, _vp, _vp, _i64, _i64, _vp]),
    "ta_momentum": (_int, [_vp, _vp, _vp, _vp, _vp, _f32, _i64, _i64, _vp]),
    "ta_update_delta_linf": (_int, [_vp, _vp, _vp, _f32, _vp, _f32, _vp, _vp, _i64, _vp]),
    "ta_update_delta_l2": (_int, [_vp, _vp, _vp, _f32, _f32, _vp, _vp, _i64, _i64, _vp]),
    "ta_mi_update": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _f32, _f32, _f32, _i64, _i64, _vp]),
    "ta_normalize_fwd": (_int, [_vp, _vp, _vp, _vp, _i64, _int, _i64, _vp]),
    "ta_normalize_bwd": (_int, [_vp, _vp, _vp, _vp, _i64, _int, _i64, _vp]),
    "ta_fused_sync_bytes": (_i64, [_i64, _i64]),
    "ta_mi_update_fused": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _f32, _i64, _i64, _vp]),
    "ta_fused_sync_error": (_int, [_vp, _i64, _i64, _vp]),
    "ta_init_delta_uniform": (_int, [_vp, _vp, _vp, _f32, _u64, _u64, _i64, _vp]),
    "ta_depthwise_conv2d_same": (_int, [_vp, _vp, _vp, _int, _i64, _int, _int, _vp]),
    "ta_depthwise_conv2d_same_separable": (_int, [_vp, _vp, _vp, _vp, _int, _i64, _int, _int, _vp]),
    "ta_dim_fwd": (_int, [_vp, _vp, _i64, _int, _int, _int, _int, _int, _vp]),
    "ta_dim_bwd": (_int, [_vp, _vp, _i64, _int, _int, _int, _int, _int, _vp]),
    "ta_scale_copies_fwd": (_int, [_vp, _vp, _i64, _i64, _int, _vp]),
    "ta_scale_copies_bwd": (_int, [_vp, _vp, _i64, _i64, _int, _vp]),
    "ta_sum_copies_bwd": (_int, [_vp, _vp, _i64, _i64, _int, _vp]),
    "ta_admix_fwd": (_int, [_vp, _vp, _vp, _i64, _i64, _int, _int, _f32, _vp]),
    "ta_admix_bwd": (_int, [_vp, _vp, _i64, _i64, _int, _int, _vp]),
    "ta_sia_fwd": (_int, [_vp, _vp, _vp, _vp, _i64, _int, _int, _int, _int, _f32, _u64, _u64, _vp]),
    "ta_sia_bwd": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _int, _int, _int, _int, _f32, _u64, _u64, _vp]),
    "ta_vmi_neighbor": (_int, [_vp, _vp, _vp, _vp, _f32, _u64, _u64, _i64, _vp]),
    "ta_grad_accumulate": (_int, [_vp, _vp, _int, _i64, _vp]),
    "ta_variance_finalize": (_int, [_vp, _vp, _vp, _f32, _i64, _vp]),
    "ta_axpy": (_int, [_vp, _vp, _f32, _vp, _i64, _vp]),
    "ta_quantize_u8_nhwc": (_int, [_vp, _vp, _vp, _i64, _int, _int, _int, _vp]),
}

ABI_VERSION = 4


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


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _check(rc, what):
    if rc != 0:
        msg = load().ta_last_error().decode("utf-8", "replace")
        raise HipExtensionError("%s failed (rc=%d): %s" % (what, rc, msg))


def _batch(t):
    return t.shape[0], t[0].numel()


# ------------------------------------------------------------------------------------------ workspaces
class Workspace:
    """Per-(device, stream) scratch for the reductions / the in-kernel exchange; grown on demand."""

    def __init__(self):
        self._l1 = {}
        self._sync = {}

    def l1(self, like, n, e):
        _ptr(like)
        key = (like.device, torch.cuda.current_stream().cuda_stream)
        need = load().ta_l1_workspace_floats(n, e)
        buf = self._l1.get(key)
        if buf is None or buf.numel() < need:
            buf = torch.empty(max(need, 1024), dtype=torch.float32, device=like.device)
            self._l1[key] = buf
        return buf

    def sync(self, like, n, e):
        _ptr(like)
        key = (like.device, torch.cuda.current_stream().cuda_stream, n, e)
        buf = self._sync.get(key)
        if buf is None:
            nbytes = load().ta_fused_sync_bytes(n, e)
            buf = torch.zeros((nbytes + 7) // 8, dtype=torch.int64, device=like.device)
            self._sync[key] = buf
        return buf


workspace = Workspace()


# ------------------------------------------------------------------------------------------- update stack
def momentum(grad, momentum_in, momentum_out, decay, variance=None):
    n, e = _batch(grad)
    ws = workspace.l1(grad, n, e)
    _check(load().ta_momentum(_ptr(grad, name="grad"), _ptr(variance, name="variance"),
                              _ptr(momentum_in, name="momentum"), _ptr(momentum_out, name="momentum_out"),
                              _ptr(ws), decay, n, e, _stream()), "ta_momentum")


def update_delta_linf(delta_in, data, momentum_, alpha, epsilon, delta_out, x_adv=None):
    alpha_t = alpha if isinstance(alpha, torch.Tensor) else None
    if alpha_t is not None and alpha_t.shape != delta_in.shape:
        alpha_t = alpha_t.expand_as(delta_in).contiguous()
    _check(load().ta_update_delta_linf(_ptr(delta_in, name="delta"), _ptr(data, name="data"),
                                       _ptr(momentum_, name="grad"), 0.0 if alpha_t is not None else float(alpha),
                                       _ptr(alpha_t, name="alpha"), float(epsilon), _ptr(delta_out, name="delta_out"),
                                       _ptr(x_adv, name="x_adv"), delta_in.numel(), _stream()),
           "ta_update_delta_linf")


def update_delta_l2(delta_in, data, grad, alpha, epsilon, delta_out):
    n, e = _batch(delta_in)
    ws = workspace.l1(delta_in, n, e)
    _check(load().ta_update_delta_l2(_ptr(delta_in, name="delta"), _ptr(data, name="data"), _ptr(grad, name="grad"),
                                     float(alpha), float(epsilon), _ptr(delta_out, name="delta_out"), _ptr(ws), n, e,
                                     _stream()), "ta_update_delta_l2")


# bench.py sets this to a list to time every fused-update launch with HIP events on the launch stream
profile_sink = None


def mi_update(grad, momentum_in, momentum_out, delta, data, decay, alpha, epsilon, variance=None, x_adv=None,
              single_launch=False):
    """Fused get_momentum + update_delta; ``delta`` is updated in place, momentum_out may alias momentum_in."""
    n, e = _batch(grad)
    if profile_sink is not None:
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        _mi_update(grad, momentum_in, momentum_out, delta, data, decay, alpha, epsilon, variance, x_adv, single_launch,
                   n, e)
        end.record()
        profile_sink.append((start, end, n, e))
        return
    _mi_update(grad, momentum_in, momentum_out, delta, data, decay, alpha, epsilon, variance, x_adv, single_launch, n, e)


def _mi_update(grad, momentum_in, momentum_out, delta, data, decay, alpha, epsilon, variance, x_adv, single_launch,
               n, e):
    args = (_ptr(grad, name="grad"), _ptr(variance, name="variance"), _ptr(momentum_in, name="momentum"),
            _ptr(momentum_out, name="momentum_out"), _ptr(delta, name="delta"), _ptr(data, name="data"),
            _ptr(x_adv, name="x_adv"))
    if single_launch:
        sync = workspace.sync(grad, n, e)
        _check(load().ta_mi_update_fused(*args, sync.data_ptr(), float(decay), float(alpha), float(epsilon), n, e,
                                         _stream()), "ta_mi_update_fused")
    else:
        ready = _take_partials(grad) if variance is None else None
        stats["partials_reused" if ready is not None else "k1_passes"] += 1
        ws = ready if ready is not None else workspace.l1(grad, n, e)
        _check(load().ta_mi_update(*args, _ptr(ws), 1 if ready is not None else 0, float(decay), float(alpha),
                                   float(epsilon), n, e, _stream()), "ta_mi_update")


# ---- producer-side |g| partial sums ---------------------------------------------------------------------
# normalize_bwd (the last kernel of the surrogate's backward) leaves the per-tile sums of |g| of the gradient it
# produced here; mi_update consumes them if -- and only if -- it is handed that very tensor, unmodified.  The
# entry holds a strong reference to the gradient, so its memory cannot be recycled while the entry is live.
_partials = None            # (grad tensor, grad._version, ws tensor)
stats = {"partials_reused": 0, "k1_passes": 0}


def _take_partials(grad):
    global _partials
    entry, _partials = _partials, None
    if entry is None:
        return None
    tensor, version, ws = entry
    if (tensor.data_ptr() == grad.data_ptr() and tensor.shape == grad.shape and grad._version == version
            and tensor._version == version):
        return ws
    if os.environ.get("TA_DEBUG_PARTIALS"):
        print("partials not reused: ptr %x vs %x, shape %s vs %s, version %d/%d vs %d" % (
            tensor.data_ptr(), grad.data_ptr(), tuple(tensor.shape), tuple(grad.shape), tensor._version,
            grad._version, version), flush=True)
    return None


def normalize_fwd(x, y, mean, std):
    n, c = x.shape[0], x.shape[1]
    _check(load().ta_normalize_fwd(_ptr(x, name="x"), _ptr(y, name="y"), _ptr(mean, name="mean"), _ptr(std, name="std"),
                                   n, c, x[0, 0].numel(), _stream()), "ta_normalize_fwd")


def normalize_bwd(gy, gx, std):
    """gx = gy / std[c]; also registers the |gx| tile sums for the fused update that consumes gx next."""
    global _partials
    n, c = gy.shape[0], gy.shape[1]
    ws = torch.empty(max(load().ta_l1_workspace_floats(n, gy[0].numel()), 1), dtype=torch.float32, device=gy.device)
    _check(load().ta_normalize_bwd(_ptr(gy, name="gy"), _ptr(gx, name="gx"), _ptr(std, name="std"), _ptr(ws), n, c,
                                   gy[0, 0].numel(), _stream()), "ta_normalize_bwd")
    _partials = (gx, gx._version, ws)


def fused_sync_check(like, n, e):
    sync = workspace.sync(like, n, e)
    _check(load().ta_fused_sync_error(sync.data_ptr(), n, e, _stream()), "ta_fused_sync_error")


def init_delta_uniform(delta, data, epsilon, seed=0, offset=0, noise=None):
    _check(load().ta_init_delta_uniform(_ptr(delta, name="delta"), _ptr(data, name="data"), _ptr(noise, name="noise"),
                                        float(epsilon), seed, offset, delta.numel(), _stream()),
           "ta_init_delta_uniform")


# --------------------------------------------------------------------------------------------- transforms
def depthwise_conv2d_same(inp, out, weight2d):
    k = weight2d.shape[-1]
    h, w = inp.shape[-2:]
    _check(load().ta_depthwise_conv2d_same(_ptr(inp, name="grad"), _ptr(out, name="out"), _ptr(weight2d, name="kernel"),
                                           k, inp.numel() // (h * w), h, w, _stream()), "ta_depthwise_conv2d_same")


def depthwise_conv2d_same_separable(inp, out, wy, wx):
    """opt-in two-pass form for outer-product kernels (rounding differs from the reference's direct convolution)"""
    h, w = inp.shape[-2:]
    _check(load().ta_depthwise_conv2d_same_separable(_ptr(inp, name="grad"), _ptr(out, name="out"), _ptr(wy, name="wy"),
                                                     _ptr(wx, name="wx"), wy.numel(), inp.numel() // (h * w), h, w,
                                                     _stream()), "ta_depthwise_conv2d_same_separable")


def dim_fwd(x, y, resize, rnd, top, left):
    size = x.shape[-1]
    _check(load().ta_dim_fwd(_ptr(x, name="x"), _ptr(y, name="y"), x.numel() // (size * size), size, resize, rnd, top,
                             left, _stream()), "ta_dim_fwd")


def dim_bwd(gy, gx, resize, rnd, top, left):
    size = gy.shape[-1]
    _check(load().ta_dim_bwd(_ptr(gy, name="gy"), _ptr(gx, name="gx"), gy.numel() // (size * size), size, resize, rnd,
                             top, left, _stream()), "ta_dim_bwd")


def scale_copies_fwd(x, y, num_scale):
    n, e = _batch(x)
    _check(load().ta_scale_copies_fwd(_ptr(x, name="x"), _ptr(y, name="y"), n, e, num_scale, _stream()),
           "ta_scale_copies_fwd")


def scale_copies_bwd(gy, gx, num_scale):
    n, e = _batch(gx)
    _check(load().ta_scale_copies_bwd(_ptr(gy, name="gy"), _ptr(gx, name="gx"), n, e, num_scale, _stream()),
           "ta_scale_copies_bwd")


def sum_copies_bwd(gy, gx, copies):
    n, e = _batch(gx)
    _check(load().ta_sum_copies_bwd(_ptr(gy, name="gy"), _ptr(gx, name="gx"), n, e, copies, _stream()),
           "ta_sum_copies_bwd")


def admix_fwd(x, perm, y, num_admix, num_scale, strength):
    n, e = _batch(x)
    _check(load().ta_admix_fwd(_ptr(x, name="x"), _ptr(perm, torch.int64, "perm"), _ptr(y, name="y"), n, e, num_admix,
                               num_scale, float(strength), _stream()), "ta_admix_fwd")


def admix_bwd(gy, gx, num_admix, num_scale):
    n, e = _batch(gx)
    _check(load().ta_admix_bwd(_ptr(gy, name="gy"), _ptr(gx, name="gx"), n, e, num_admix, num_scale, _stream()),
           "ta_admix_bwd")


def sia_fwd(x, plan, y, copies, num_block, noise_radius, seed=0, offset=0, noise=None):
    h, w = x.shape[-2:]
    _check(load().ta_sia_fwd(_ptr(x, name="x"), _ptr(plan, torch.int32, "plan"), _ptr(noise, name="noise"), _ptr(y, name="y"),
                             x.numel() // (h * w), h, w, copies, num_block, float(noise_radius), seed, offset, _stream()),
           "ta_sia_fwd")


def sia_bwd(gy, plan, x, gx, copies, num_block, noise_radius, seed=0, offset=0, noise=None):
    h, w = x.shape[-2:]
    _check(load().ta_sia_bwd(_ptr(gy, name="gy"), _ptr(plan, torch.int32, "plan"), _ptr(x, name="x"), _ptr(noise, name="noise"),
                             _ptr(gx, name="gx"), x.numel() // (h * w), h, w, copies, num_block, float(noise_radius), seed,
                             offset, _stream()), "ta_sia_bwd")


# ---------------------------------------------------------------------------------------------- VMI / NI
def vmi_neighbor(data, delta, out, radius, seed=0, offset=0, noise=None):
    _check(load().ta_vmi_neighbor(_ptr(data, name="data"), _ptr(delta, name="delta"), _ptr(noise, name="noise"),
                                  _ptr(out, name="out"), float(radius), seed, offset, data.numel(), _stream()),
           "ta_vmi_neighbor")


def grad_accumulate(acc, grad, first):
    _check(load().ta_grad_accumulate(_ptr(acc, name="acc"), _ptr(grad, name="grad"), 1 if first else 0, acc.numel(),
                                     _stream()), "ta_grad_accumulate")


def variance_finalize(acc, cur_grad, out, count):
    _check(load().ta_variance_finalize(_ptr(acc, name="acc"), _ptr(cur_grad, name="grad"), _ptr(out, name="variance"),
                                       float(count), acc.numel(), _stream()), "ta_variance_finalize")


def axpy(x, m, coeff, out):
    _check(load().ta_axpy(_ptr(x, name="x"), _ptr(m, name="momentum"), float(coeff), _ptr(out, name="out"), x.numel(),
                          _stream()), "ta_axpy")


# ------------------------------------------------------------------------------------------------- output
def quantize_u8_nhwc(data, delta, out):
    n, c, h, w = data.shape
    _check(load().ta_quantize_u8_nhwc(_ptr(data, name="data"), _ptr(delta, name="delta"),
                                      _ptr(out, torch.uint8, "out"), n, c, h, w, _stream()), "ta_quantize_u8_nhwc")
