"""ctypes binding of libta_ck.so (include/ta_ck.h): convolutions of a ResNet surrogate with the glue pass that follows them as
the convolution's epilogue -- composable_kernel instances, the library kernels MIOpen itself dispatches for these layers.

Opt-in (``TA_CK_EPILOGUE=1``; ``bench.py`` switches it on and says so) and used by ``backbones/fused.py`` only.  Per
(epilogue, shape) the first call times the two-kernel form it would replace -- MIOpen's convolution + the glue kernel of
libta_hip.so -- against every tile configuration of the fused form and keeps the faster (``choose``): a layer for which MIOpen's
assembly kernels win stays on them (a few 3 x 3 layers of the deep stages do, profiles/r06/ck_site_decisions_b125_r6h.txt).
Backward sites run the FORWARD kernels on the rewritten problem (``backward_as_forward``): composable_kernel's own backward-data
kernels lost at every site (and zero-fill their output first).  Like MIOpen's
find mode this makes the choice of kernel, hence the last bits of an activation, a property of the process: under
``TA_DETERMINISTIC=1`` the path is off.  No fallback inside: a missing library raises when the path is asked for.
"""
import ctypes
import os
import sys

import torch

from . import _hip

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TA_CK_LIB", os.path.join(_HERE, "lib", "libta_ck.so"))
ABI_VERSION = 2

FWD_BIAS_RELU, FWD_BIAS_ADD_RELU, FWD_BIAS_ADD_BIAS_RELU, FWD_MASK, FWD_ADD_MASK = 1, 2, 3, 4, 5
UNSUPPORTED = 1

_int, _vp = ctypes.c_int, ctypes.c_void_p
SIGNATURES = {
    "ta_ck_abi_version": (_int, []),
    "ta_ck_last_error": (ctypes.c_char_p, []),
    "ta_ck_instances": (_int, [_int, _int, _int, _int]),
    "ta_ck_instance_name": (ctypes.c_char_p, [_int, _int, _int, _int, _int]),
    "ta_ck_conv": (_int, [_int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _int, _int, _vp]),
}

_lib = None
stats = {"fused_launches": 0, "tuned_sites": 0, "sites_on_ck": 0, "sites_from_disk": 0}
plans = {}                       # site key -> configuration index, or None = the two-kernel form is faster there


def enabled():
    return os.environ.get("TA_CK_EPILOGUE", "0") == "1" and os.environ.get("TA_DETERMINISTIC", "0") != "1"


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise _hip.HipExtensionError("TA_CK_EPILOGUE=1 but %s is missing: build it with `make -C transferattack_amd/csrc` "
                                     "(python -c 'import __graft_entry__ as g; g.build()')" % LIB_PATH)
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as exc:
        raise _hip.HipExtensionError("cannot load %s: %s" % (LIB_PATH, exc))
    for name, (restype, argtypes) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise _hip.HipExtensionError("%s does not export %s (stale build?)" % (LIB_PATH, name))
        fn.restype, fn.argtypes = restype, argtypes
    if lib.ta_ck_abi_version() != ABI_VERSION:
        raise _hip.HipExtensionError("libta_ck.so ABI %d, binding %d" % (lib.ta_ck_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def weight_kyxc(conv):
    """the convolution's weight as dense [k, y, x, c] memory, cached on the module (frozen weights: built once)"""
    w = conv.weight
    cached = getattr(conv, "_ta_ck_weight", None)
    if cached is None or cached[0] != w._version or cached[1].device != w.device:
        cached = (w._version, w.detach().permute(0, 2, 3, 1).contiguous())
        conv._ta_ck_weight = cached
    return cached[1]


def geometry(x_shape, conv):
    """(n, c, h, w, k, ksize, stride, pad) of ``conv`` on an input of ``x_shape``, or None if the entry points do not take it"""
    k, c, ky, kx = conv.weight.shape
    st, pd, dl = tuple(conv.stride), tuple(conv.padding), tuple(conv.dilation)
    if ky != kx or st[0] != st[1] or pd[0] != pd[1] or dl != (1, 1) or conv.groups != 1 or isinstance(conv.padding, str):
        return None
    n, cin, h, w = x_shape
    if cin != c:
        return None
    geom = (int(n), int(c), int(h), int(w), int(k), int(ky), int(st[0]), int(pd[0]))
    ho, wo = out_hw(geom)
    if max(n * h * w * c, n * ho * wo * k) * 4 >= 2 ** 31:       # the kernels address with 32-bit byte offsets: larger maps stay on MIOpen
        return None
    return geom


def out_hw(geom):
    n, c, h, w, k, ks, st, pd = geom
    return (h + 2 * pd - ks) // st + 1, (w + 2 * pd - ks) // st + 1


def conv(kind, index, a, w, d0, d1, d2, e, geom, probing=False):
    """one launch on ``a``'s device and current stream.  ``probing`` (the tuner): -> 0 launched / UNSUPPORTED (this configuration
    does not take the problem); otherwise anything but a launch raises -- a planned configuration must run"""
    p = lambda t: None if t is None else t.data_ptr()      # noqa: E731
    dev = a.device
    if dev.index is not None and dev.index != torch.cuda.current_device():
        with torch.cuda.device(dev):                        # the launch needs the tensor's device current in THIS thread
            rc = load().ta_ck_conv(kind, index, p(a), p(w), p(d0), p(d1), p(d2), p(e), *geom, torch.cuda.current_stream(dev).cuda_stream)
    else:
        rc = load().ta_ck_conv(kind, index, p(a), p(w), p(d0), p(d1), p(d2), p(e), *geom, torch.cuda.current_stream(dev).cuda_stream)
    if rc != 0 and not (probing and rc == UNSUPPORTED):
        raise _hip.HipExtensionError("ta_ck_conv(kind %d, configuration %d, %s) failed (rc=%d): %s" % (
            kind, index, geom, rc, load().ta_ck_last_error().decode("utf-8", "replace") or "the configuration does not take the problem"))
    if rc == 0:
        stats["fused_launches"] += 1
    return rc


def _time(fn, reps=6):
    """milliseconds per call on the device.  A first call outside the clock, then ``reps`` queued back to back: the host's own
    cost of a call (torch's dispatcher for the two-kernel form) hides behind the previous call's kernels, as it does in the loop"""
    fn()
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    start.record()
    for _ in range(reps):
        fn()
    end.record()
    end.synchronize()
    return start.elapsed_time(end) / reps


def weight_flipped_cyxk(conv):
    """the filter of the FORWARD convolution that computes ``conv``'s input gradient (stride 1): w'[c][Y-1-y][X-1-x][k] = w[k][c][y][x],
    dense [c, y, x, k] memory -- its "output" channels are conv's input channels; cached on the module"""
    w = conv.weight
    cached = getattr(conv, "_ta_ck_weight_t", None)
    if cached is None or cached[0] != w._version or cached[1].device != w.device:
        cached = (w._version, w.detach().flip(2, 3).permute(1, 2, 3, 0).contiguous())
        conv._ta_ck_weight_t = cached
    return cached[1]


def backward_as_forward(geom):
    """the forward problem whose result is the input gradient of the stride-1 convolution ``geom``, or None (strided convolution)"""
    n, c, h, w, k, ks, st, pd = geom
    if st != 1 or ks - 1 - pd < 0:
        return None
    ho, wo = out_hw(geom)
    return (n, k, ho, wo, c, ks, 1, ks - 1 - pd)


# ---- the decisions persist (as MIOpen's find results do in its user find-db): tuning the ~40 sites of a ResNet-50 takes 1-2 s per
# batch shape, more than a 1000-image job saves (main.py end to end: 2.4 s of attack time, profiles/r06/e2e_main_1000png_r6p.jsonl).
# One JSON file per (device, library build): {site key: [family, configuration] | null}.  TA_CK_PLAN_CACHE=<path> moves it, =0
# switches it off.
_disk = None


def _plan_file():
    where = os.environ.get("TA_CK_PLAN_CACHE", "")
    if where == "0":
        return None
    if where:
        return where
    import zlib
    lib = load()
    names = "|".join(lib.ta_ck_instance_name(kind, ks, 1, ks // 2, i).decode() for kind in (FWD_BIAS_RELU, FWD_MASK, FWD_ADD_MASK)
                     for ks in (1, 3) for i in range(lib.ta_ck_instances(kind, ks, 1, ks // 2)))
    props = torch.cuda.get_device_properties(torch.cuda.current_device())
    tag = "%s_%dcu_abi%d_%08x" % (getattr(props, "gcnArchName", props.name).split(":")[0], props.multi_processor_count, ABI_VERSION,
                                   zlib.crc32(names.encode()))
    return os.path.join(os.path.expanduser("~"), ".cache", "transferattack_amd", "ck_plans_%s.json" % tag)


def _disk_plans():
    global _disk
    if _disk is None:
        _disk = {}
        path = _plan_file()
        if path and os.path.isfile(path):
            try:
                import json
                _disk = json.load(open(path))
            except (OSError, ValueError):
                _disk = {}
    return _disk


def _remember(key, best):
    path = _plan_file()
    if not path:
        return
    import json
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        merged = {}
        if os.path.isfile(path):                           # another rank may have written since this process read the file
            try:
                merged = json.load(open(path))
            except ValueError:
                merged = {}
        merged[repr(key)] = None if best is None else list(best)
        _disk_plans()[repr(key)] = merged[repr(key)]
        tmp = "%s.%d.tmp" % (path, os.getpid())
        json.dump(merged, open(tmp, "w"))
        os.replace(tmp, path)
    except OSError:
        pass                                               # a read-only home: the decisions simply do not persist


def choose(key, families, run_two_kernels):
    """-> (family, configuration index) of the fastest fused form for this site, or None where the two-kernel form is at least as
    fast.  ``families``: [(kind, geometry, run(index) -> rc)] -- alternative kernel families for one site.  Decided once per key by timing all of them on the caller's own tensors (none may modify its inputs); a
    fused form must win by 3 % to be taken."""
    if key in plans:
        return plans[key]
    stored = _disk_plans().get(repr(key), "absent")
    if stored != "absent":                                  # decided by an earlier process on this device with this library build
        best = None if stored is None else (int(stored[0]), int(stored[1]))
        if best is None or (best[0] < len(families) and best[1] < load().ta_ck_instances(families[best[0]][0], *families[best[0]][1][5:8])):
            plans[key] = best
            stats["sites_from_disk"] += 1
            stats["sites_on_ck"] += best is not None
            return best
    best, best_ms, base_ms = None, float("inf"), None
    for fam, (kind, geom, run) in enumerate(families):
        n_cfg = load().ta_ck_instances(kind, geom[5], geom[6], geom[7])
        if n_cfg <= 0:
            continue
        if base_ms is None:
            torch.cuda.synchronize()
            base_ms = _time(run_two_kernels)
        for idx in range(n_cfg):
            if run(idx) != 0:                               # (the site's run passes probing=True)
                continue
            ms = _time(lambda: run(idx))
            if ms < best_ms:
                best, best_ms = (fam, idx), ms
    if best is not None and best_ms > 0.97 * base_ms:
        best = None
    if base_ms is not None and os.environ.get("TA_CK_DEBUG"):
        name = "none faster" if best is None else "%.1f us (%s, kind %d)" % (
            best_ms * 1e3, load().ta_ck_instance_name(families[best[0]][0], *families[best[0]][1][5:8], best[1]).decode(), families[best[0]][0])
        print("ck site %s: two kernels %.1f us, fused %s" % (key, base_ms * 1e3, name), file=sys.stderr, flush=True)
    plans[key] = best
    if base_ms is not None:
        _remember(key, best)
    stats["tuned_sites"] += base_ms is not None
    stats["sites_on_ck"] += best is not None
    return best
