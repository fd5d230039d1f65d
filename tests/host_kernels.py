"""TEST INFRASTRUCTURE: run the REAL binding (`transferattack_amd._hip`, argument marshalling included) against the
kernel sources compiled for the host (tests/hipcpu) instead of libta_hip.so, on CPU tensors.

``install(monkeypatch)`` swaps, for one test, the loaded library, the device check of the pointer helper and the
stream lookup.  The product never sees this module; without it `_hip` still refuses CPU tensors."""
import os
import sys
import types

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "hipcpu"))
import hipcpu_build                    # noqa: E402
from transferattack_amd import _hip     # noqa: E402

_libs = {}


def _bind(tag):
    if tag not in _libs:
        lib = hipcpu_build.load(tag)
        for name, (restype, argtypes) in _hip.SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = restype, argtypes
        assert lib.ta_abi_version() == _hip.ABI_VERSION
        _libs[tag] = lib
    return _libs[tag]


def _ptr(t, dtype=torch.float32, name="tensor"):
    if t is None:
        return None
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    return t.data_ptr()


def install(monkeypatch, tag=None, env=None):
    """``env``: environment knobs for the duration of the test; ``tag``: a private copy of the library (own statics)."""
    for key, value in (env or {}).items():
        monkeypatch.setenv(key, value)
    monkeypatch.setattr(_hip, "_lib", _bind(tag))
    monkeypatch.setattr(_hip, "_ptr", _ptr)
    monkeypatch.setattr(_hip, "_stream", lambda like=None: None)

    def call(name, like, *args):                 # no device / stream on the host: the "launch" runs synchronously
        _hip._check(getattr(_hip._sync_options(_hip.load()), name)(*args, None), name)

    monkeypatch.setattr(_hip, "_call", call)
    monkeypatch.setattr(_hip, "workspace", _hip.Workspace())
    from transferattack_amd import attack as ta_attack, utils as ta_utils
    cpu = lambda: torch.device("cpu")            # noqa: E731  -- "the device this process drives" is the host here
    monkeypatch.setattr(ta_utils, "default_device", cpu)
    monkeypatch.setattr(ta_attack, "default_device", cpu)
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a, **k: types.SimpleNamespace(cuda_stream=0))
