"""TEST INFRASTRUCTURE -- ctypes view of oracle/_build/libta_oracle.so (built from oracle/ta_oracle.c).

numpy in, numpy out.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline use this.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libta_oracle.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_u8p = ctypes.POINTER(ctypes.c_uint8)


def build():
    subprocess.run(["make", "-C", _HERE, "--no-print-directory"], check=True, stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.isfile(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.ta_oracle_aten_row_sum.restype = ctypes.c_float
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_f32p)


def aten_row_sum(row, lanes=8):
    row, p = _f(row)
    return np.float32(lib().ta_oracle_aten_row_sum(p, ctypes.c_int64(row.size), ctypes.c_int(lanes)))


def momentum(g, m_in, decay, lanes=8):
    g, pg = _f(g)
    n, e = g.shape[0], g[0].size
    out = np.empty_like(g)
    if m_in is None:
        pm = None
    else:
        m_in, pm = _f(m_in)
    lib().ta_oracle_momentum(pg, pm, out.ctypes.data_as(_f32p), ctypes.c_float(decay), ctypes.c_int64(n),
                             ctypes.c_int64(e), ctypes.c_int(lanes))
    return out


def update_delta_linf(delta, x, m, alpha, eps, alpha_t=None):
    delta, pd = _f(delta)
    x, px = _f(x)
    m, pm = _f(m)
    pa = None
    if alpha_t is not None:
        alpha_t, pa = _f(alpha_t)
    out = np.empty_like(delta)
    lib().ta_oracle_update_delta_linf(pd, px, pm, ctypes.c_float(alpha), pa, ctypes.c_float(eps),
                                      out.ctypes.data_as(_f32p), ctypes.c_int64(delta.size))
    return out


def quantize_u8_nhwc(x, delta):
    x, px = _f(x)
    delta, pd = _f(delta)
    n, c, h, w = x.shape
    out = np.empty((n, h, w, c), dtype=np.uint8)
    lib().ta_oracle_quantize_u8_nhwc(px, pd, out.ctypes.data_as(_u8p), ctypes.c_int64(n), c, h, w)
    return out


def depthwise_conv2d_same(x, w2d):
    x, px = _f(x)
    w2d, pw = _f(w2d)
    k = w2d.shape[-1]
    h, wd = x.shape[-2:]
    planes = x.size // (h * wd)
    out = np.empty_like(x)
    lib().ta_oracle_depthwise_conv2d_same(px, out.ctypes.data_as(_f32p), pw, k, ctypes.c_int64(planes), h, wd)
    return out


def bilinear_fwd(x, out_size):
    x, px = _f(x)
    size = x.shape[-1]
    planes = x.size // (size * size)
    out = np.empty(x.shape[:-2] + (out_size, out_size), dtype=np.float32)
    lib().ta_oracle_bilinear_fwd(px, out.ctypes.data_as(_f32p), ctypes.c_int64(planes), size, out_size)
    return out


def bilinear_bwd(gy, in_size, mode=0):
    gy, pg = _f(gy)
    out_size = gy.shape[-1]
    planes = gy.size // (out_size * out_size)
    gx = np.empty(gy.shape[:-2] + (in_size, in_size), dtype=np.float32)
    lib().ta_oracle_bilinear_bwd(pg, gx.ctypes.data_as(_f32p), ctypes.c_int64(planes), in_size, out_size, mode)
    return gx


def dim_fwd(x, geom, resize):
    apply, rnd, top, left = geom
    x, px = _f(x)
    if not apply:
        return x.copy()
    size = x.shape[-1]
    planes = x.size // (size * size)
    out = np.empty_like(x)
    lib().ta_oracle_dim_fwd(px, out.ctypes.data_as(_f32p), ctypes.c_int64(planes), size, resize, rnd, top, left)
    return out


def dim_bwd(gy, geom, resize, mode=0):
    apply, rnd, top, left = geom
    gy, pg = _f(gy)
    if not apply:
        return gy.copy()
    size = gy.shape[-1]
    planes = gy.size // (size * size)
    out = np.empty_like(gy)
    lib().ta_oracle_dim_bwd(pg, out.ctypes.data_as(_f32p), ctypes.c_int64(planes), size, resize, rnd, top, left,
                            mode)
    return out


def philox_uniform(numel, seed, offset, r):
    out = np.empty(numel, dtype=np.float32)
    lib().ta_oracle_philox_uniform(out.ctypes.data_as(_f32p), ctypes.c_int64(numel), ctypes.c_uint64(seed),
                                   ctypes.c_uint64(offset), ctypes.c_float(r))
    return out


_i64p = ctypes.POINTER(ctypes.c_int64)


def scale_copies_fwd(x, num_scale):
    x, px = _f(x)
    out = np.empty((num_scale * x.shape[0],) + x.shape[1:], dtype=np.float32)
    lib().ta_oracle_scale_copies_fwd(px, out.ctypes.data_as(_f32p), ctypes.c_int64(x.size), num_scale)
    return out


def scale_copies_bwd(gy, num_scale):
    gy, pg = _f(gy)
    out = np.empty((gy.shape[0] // num_scale,) + gy.shape[1:], dtype=np.float32)
    lib().ta_oracle_scale_copies_bwd(pg, out.ctypes.data_as(_f32p), ctypes.c_int64(out.size), num_scale)
    return out


def admix_fwd(x, perm, num_admix, num_scale, strength):
    x, px = _f(x)
    perm = np.ascontiguousarray(perm, dtype=np.int64)
    n, e = x.shape[0], x[0].size
    out = np.empty((num_scale * num_admix * n,) + x.shape[1:], dtype=np.float32)
    lib().ta_oracle_admix_fwd(px, perm.ctypes.data_as(_i64p), out.ctypes.data_as(_f32p), ctypes.c_int64(n),
                              ctypes.c_int64(e), num_admix, num_scale, ctypes.c_float(strength))
    return out


def admix_bwd(gy, num_admix, num_scale):
    gy, pg = _f(gy)
    n = gy.shape[0] // (num_admix * num_scale)
    out = np.empty((n,) + gy.shape[1:], dtype=np.float32)
    lib().ta_oracle_admix_bwd(pg, out.ctypes.data_as(_f32p), ctypes.c_int64(n), ctypes.c_int64(out[0].size),
                              num_admix, num_scale)
    return out


def normalize_fwd(x, mean, std):
    x, px = _f(x)
    mean, pm = _f(mean)
    std, ps = _f(std)
    out = np.empty_like(x)
    lib().ta_oracle_normalize_fwd(px, out.ctypes.data_as(_f32p), pm, ps, ctypes.c_int64(x.shape[0]), x.shape[1],
                                  ctypes.c_int64(x[0, 0].size))
    return out


def normalize_bwd(gy, std):
    gy, pg = _f(gy)
    std, ps = _f(std)
    out = np.empty_like(gy)
    lib().ta_oracle_normalize_bwd(pg, out.ctypes.data_as(_f32p), ps, ctypes.c_int64(gy.shape[0]), gy.shape[1],
                                  ctypes.c_int64(gy[0, 0].size))
    return out


def variance_finalize(acc, cur, count):
    acc, pa = _f(acc)
    cur, pc = _f(cur)
    out = np.empty_like(acc)
    lib().ta_oracle_variance_finalize(pa, pc, out.ctypes.data_as(_f32p), ctypes.c_float(count),
                                      ctypes.c_int64(acc.size))
    return out


_i32p = ctypes.POINTER(ctypes.c_int32)


def sia_fwd(x, plan, noise, num_block):
    """x [N,C,H,W], plan int32 [copies, stride] (transforms.sia_draw), noise shaped like the output stack"""
    x, px = _f(x)
    noise, pn = _f(noise)
    plan = np.ascontiguousarray(plan, dtype=np.int32)
    copies = plan.shape[0]
    h, w = x.shape[-2:]
    y = np.empty((copies * x.shape[0],) + x.shape[1:], dtype=np.float32)
    lib().ta_oracle_sia_fwd(px, plan.ctypes.data_as(_i32p), pn, y.ctypes.data_as(_f32p),
                            ctypes.c_int64(x.size // (h * w)), h, w, copies, num_block)
    return y


def sia_bwd(gy, plan, x, noise, num_block):
    gy, pg = _f(gy)
    x, px = _f(x)
    noise, pn = _f(noise)
    plan = np.ascontiguousarray(plan, dtype=np.int32)
    h, w = x.shape[-2:]
    gx = np.empty_like(x)
    lib().ta_oracle_sia_bwd(pg, plan.ctypes.data_as(_i32p), px, pn, gx.ctypes.data_as(_f32p),
                            ctypes.c_int64(x.size // (h * w)), h, w, plan.shape[0], num_block)
    return gx
