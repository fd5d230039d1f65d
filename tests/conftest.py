import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, ORACLE_DIR):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")
    config.addinivalue_line("markers", "gpu_long: the 1000-image forms of the two slowest ASR jobs (MI355X, ~8 min; -m gpu_long): "
                                       "kept out of -m gpu so that tier fits the driver's step limit")
    # test infrastructure may build what it checks (a fresh checkout has no .so: they are git-ignored); the product
    # itself never builds or falls back at run time
    import subprocess
    lib = os.path.join(ROOT, "transferattack_amd", "lib", "libta_hip.so")
    if not os.path.isfile(lib):
        subprocess.run(["make", "-C", os.path.join(ROOT, "transferattack_amd", "csrc"), "-j8"], check=True,
                       stdout=subprocess.DEVNULL)
    if not os.path.isfile(os.path.join(ORACLE_DIR, "_build", "libta_oracle.so")):
        subprocess.run(["make", "-C", ORACLE_DIR], check=True, stdout=subprocess.DEVNULL)


# Collection order of the -m gpu tier.  The driver runs it with -x, so ONE failure hides everything collected after it
# (round 3: a surrogate-conditioning bound tripped at test 60 and the 106 kernel tests behind it never ran).  Order by
# how deterministic the checked quantity is: bit-exact kernel parity first, loops with replayed reference gradients next,
# whole surrogates on the device (MIOpen algorithm choice, atomics: run-to-run noise) last, statistics (ASR) at the very end.
_GPU_TIERS = (
    ("test_hip_kernels.py", None, 0),
    ("test_hip_ck.py", None, 0),
    ("test_hip_configs.py", ("kernel_on_device",), 0),
    ("test_zz_hip_widened.py", ("kernels", "plane_groups", "properties", "full_size", "spectrum", "largest_ratio",
                                "registry_rules", "reference_sum_order"), 0),
    ("test_hip_loops_golden.py", None, 1),
    ("test_hip_attacks.py", ("replay",), 1),
    ("test_hip_configs.py", ("replay",), 2),
    ("test_zz_hip_widened.py", None, 3),
    ("test_hip_attacks.py", ("gradient_accuracy",), 6),
    ("test_hip_attacks.py", None, 4),
    ("test_hip_rccl.py", None, 5),
    ("test_hip_configs.py", None, 6),
    ("test_hip_asr1000.py", None, 7),
    ("test_hip_asr_trained.py", None, 7),
)


def _gpu_tier(item):
    fname = os.path.basename(str(item.fspath))
    for f, keys, tier in _GPU_TIERS:
        if f == fname and (keys is None or any(k in item.name for k in keys)):
            return tier
    return 4


def pytest_collection_modifyitems(config, items):
    order = {id(it): i for i, it in enumerate(items)}
    items.sort(key=lambda it: ((_gpu_tier(it), order[id(it)]) if "gpu" in it.keywords else (-1, order[id(it)])))
    if "gpu_long" not in (config.getoption("-m") or ""):          # ... and never as a side effect of -m "not gpu"
        unasked = pytest.mark.skip(reason="runs only when asked for: -m gpu_long")
        for item in items:
            if "gpu_long" in item.keywords:
                item.add_marker(unasked)
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords or "gpu_long" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    return load


def u8_images(n, size, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, 256, (n, 3, size, size), generator=g, dtype=torch.uint8)


def ulp_diff(a, b):
    """max distance in units-in-the-last-place between two fp32 arrays (NaNs must coincide)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    nan_a, nan_b = np.isnan(a), np.isnan(b)
    assert np.array_equal(nan_a, nan_b), "NaN patterns differ"
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia)
    ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    d = np.abs(ia - ib)
    d[nan_a] = 0
    return int(d.max()) if d.size else 0


def assert_momentum_close(m_hip, m_ref, grad, m_prev, decay):
    """m' = m*decay + g/mean|g|: the HIP path adds the |g| partial sums in its own fixed order, so the mean (and
    q = g/mean) can differ from ATen's in the last bits; the error of m' is bounded by a few ulp of its two
    terms (NOT of m' itself -- the terms may cancel).  Bound used: 8 * 2^-24 * (|m*decay| + |q|)."""
    m_hip, m_ref = np.asarray(m_hip, dtype=np.float64), np.asarray(m_ref, dtype=np.float64)
    assert np.array_equal(np.isnan(m_hip), np.isnan(m_ref)), "NaN patterns differ"
    g = np.asarray(grad, dtype=np.float64)
    mean = np.abs(g).reshape(g.shape[0], -1).mean(axis=1).reshape((-1,) + (1,) * (g.ndim - 1))
    with np.errstate(divide="ignore", invalid="ignore"):
        q = np.abs(g / mean)
    prev = 0.0 if m_prev is None else np.abs(np.asarray(m_prev, dtype=np.float64) * decay)
    bound = 8 * 2.0 ** -24 * (prev + q) + 1e-30
    ok = np.isnan(m_ref) | (np.abs(m_hip - m_ref) <= bound)
    assert ok.all(), "momentum off by %.3e (bound %.3e)" % (np.nanmax(np.abs(m_hip - m_ref)), np.nanmax(bound))
