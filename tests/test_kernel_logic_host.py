"""Kernel LOGIC on the CPU.  The .hip sources (update, elementwise, TIM, DIM) are compiled for the host under a
thread-per-lane stand-in for the HIP runtime (tests/hipcpu) and driven through the real ctypes binding
(tests/host_kernels.py); the assertions are the ones the `-m gpu` tests make on MI355X (tests/test_hip_kernels.py,
re-used here function by function at sizes the stand-in finishes quickly).

What this proves: indexing, tiling, staging through the shared arrays, barriers, tail handling, rounding order and
the binding's argument marshalling.  What it does not: anything about the gfx950 build (another compiler, no real
wave execution, no non-temporal paths) or about speed -- the GPU tests stay the parity
gate.  Test infrastructure only."""
import numpy as np
import pytest
import torch

import c_oracle as C
import host_kernels
import test_hip_kernels as G


@pytest.fixture(autouse=True)
def host_backend(monkeypatch):
    host_kernels.install(monkeypatch)
    monkeypatch.setattr(G, "DEV", "cpu")


# ---- re-used as they are (golden tensors of the reference; small ragged shapes)
test_update_stack_golden = G.test_update_stack_golden
test_update_delta_variants_golden = G.test_update_delta_variants_golden
test_quantiser = G.test_quantiser
test_tim_golden = G.test_tim_golden
test_dim_golden = G.test_dim_golden
test_sim_admix_golden = G.test_sim_admix_golden
test_sum_copies_bwd = G.test_sum_copies_bwd
test_sim_admix_ragged = G.test_sim_admix_ragged
test_vmi_kernels_and_philox = G.test_vmi_kernels_and_philox


test_normalize_and_producer_side_partials = G.test_normalize_and_producer_side_partials
test_tim_random = G.test_tim_random
test_dim_random = G.test_dim_random


@pytest.mark.parametrize("shape", [(32, 3, 224, 224), (5, 3, 37, 41), (1, 3, 8, 8), (3, 3, 299, 299), (2, 1, 1, 7)])
def test_fused_update_random(shape):
    G.test_fused_update_random(shape)


test_producer_side_partials_all_kernels = G.test_producer_side_partials_all_kernels
test_sum_members = G.test_sum_members
test_update_without_momentum = G.test_update_without_momentum
test_byte_source_of_the_fused_update = G.test_byte_source_of_the_fused_update
test_normalize_folded_update = G.test_normalize_folded_update


@pytest.mark.parametrize("n,oh,ow", [(2, 16, 16), (1, 9, 37)])
def test_stem_kernel_leaves_the_sums(n, oh, ow):
    G.test_stem_kernel_leaves_the_sums(n, oh, ow)
test_resize_normalize_kernels = G.test_resize_normalize_kernels




def test_bad_arguments_fail_loudly():
    """the library's own argument checks (TA_EINVAL + message) and the binding's dtype check; the device check of the
    binding is the one thing the stand-in replaces, see tests/test_host_logic.py::test_no_cpu_fallback for it"""
    from transferattack_amd import _hip
    x = torch.rand(2, 3, 8, 8)
    with pytest.raises(_hip.HipExtensionError, match="alias"):
        _hip.depthwise_conv2d_same(x, x, torch.rand(3, 3))
    with pytest.raises(_hip.HipExtensionError, match="geometry"):
        _hip.dim_fwd(x, torch.empty_like(x), 8, 9, 0, 0)
    with pytest.raises(TypeError):
        _hip.momentum(x.double(), None, torch.empty_like(x), 1.0)


def test_dim_lane_and_table_kernels(golden):
    """resize ratios 1.1 / 1.5 run the lane-per-column kernels, ratio 2 the table-driven gathers"""
    G.test_dim_golden(golden)
    G.test_dim_random(224, 1.1, [(224, 0, 0), (224, 22, 22), (245, 0, 1), (245, 1, 0), (237, 3, 5), (230, 16, 0)])
    G.test_dim_random(64, 1.5, [(64, 0, 31), (95, 0, 0), (80, 7, 9)])
    G.test_dim_random(33, 2.0, [(40, 5, 20), (65, 0, 1)])


# ---- out-of-bounds reads: inputs placed flush against inaccessible pages (a stray load faults instead of passing)
def _guarded(array, at_end):
    """copy of ``array`` inside an anonymous mapping, its last (or first) byte adjacent to a PROT_NONE page"""
    import ctypes
    import mmap
    page = mmap.PAGESIZE
    nbytes = array.nbytes
    body = (nbytes + page - 1) // page * page
    m = mmap.mmap(-1, body + 2 * page)
    base = ctypes.addressof(ctypes.c_char.from_buffer(m))
    libc = ctypes.CDLL(None, use_errno=True)
    libc.mprotect.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    assert libc.mprotect(base, page, 0) == 0 and libc.mprotect(base + page + body, page, 0) == 0
    start = page + (body - nbytes if at_end else 0)
    view = np.frombuffer(m, dtype=array.dtype, count=array.size, offset=start).reshape(array.shape)
    view[...] = array
    return view, m


def test_dim_reads_stay_in_bounds():
    from transferattack_amd import _hip
    gen = torch.Generator().manual_seed(1)
    for size, resize, geoms in ((224, 246, [(245, 0, 1), (224, 22, 0), (230, 0, 16)]), (33, 66, [(40, 5, 20), (65, 0, 1)])):
        x = torch.rand(1, 1, size, size, generator=gen).numpy()
        for at_end in (True, False):
            xg, keep = _guarded(x, at_end)
            xt = torch.from_numpy(xg)
            for rnd, top, left in geoms:
                yg, keep_y = _guarded(np.zeros_like(x), at_end)            # the outputs are fenced in as well
                gg, keep_g = _guarded(np.zeros_like(x), at_end)
                y, gx = torch.from_numpy(yg), torch.from_numpy(gg)
                _hip.dim_fwd(xt, y, resize, rnd, top, left)
                _hip.dim_bwd(xt, gx, resize, rnd, top, left)
                assert np.array_equal(y.numpy(), C.dim_fwd(x, (True, rnd, top, left), resize))
                assert np.array_equal(gx.numpy(), C.dim_bwd(x, (True, rnd, top, left), resize))
                del y, gx, yg, gg
            del xt, xg
            keep = None


def test_tim_reads_stay_in_bounds():
    from transferattack_amd import _hip
    gen = torch.Generator().manual_seed(2)
    for shape, k in (((1, 1, 224, 224), 15), ((1, 1, 37, 41), 15), ((1, 1, 64, 64), 7), ((1, 1, 33, 33), 4)):
        grad = torch.randn(shape, generator=gen).numpy()
        w = torch.rand(k, k, generator=gen)
        w = (w / w.sum()).contiguous()
        for at_end in (True, False):
            gg, keep = _guarded(grad, at_end)
            og, keep_o = _guarded(np.zeros_like(grad), at_end)
            out = torch.from_numpy(og)
            _hip.depthwise_conv2d_same(torch.from_numpy(gg), out, w)
            assert np.array_equal(out.numpy(), C.depthwise_conv2d_same(grad, w.numpy()))
            del gg, out, og
            keep = None


# ---- SIA: the GPU tests of tests/test_zz_hip_widened.py, on the host stand-in
import test_zz_hip_widened as W          # noqa: E402


@pytest.fixture
def widened_on_host(monkeypatch):
    monkeypatch.setattr(W, "DEV", "cpu")
    monkeypatch.setattr(W, "BOUND", 0.0)


def test_sia_kernels_golden(golden, widened_on_host):
    W.test_sia_kernels_golden(golden)


@pytest.mark.parametrize("shape,nb,copies", [((2, 3, 224, 224), 3, 5), ((1, 3, 37, 41), 3, 4), ((3, 1, 16, 100), 2, 3),
                                             ((1, 2, 9, 9), 1, 2), ((1, 3, 64, 64), 5, 6), ((2, 3, 299, 299), 3, 2),
                                             ((1, 1, 5, 700), 4, 3)])
def test_sia_kernels_random(widened_on_host, shape, nb, copies):
    W.test_sia_kernels_random(shape, nb, copies)


@pytest.mark.parametrize("shape", [(4096, 3, 4, 8), (2050, 3, 4, 8), (1538, 3, 4, 8), (769, 3, 4, 8), (1025, 2, 4, 8)])
def test_bsr_plane_groups(widened_on_host, shape):
    W.test_bsr_plane_groups(shape)


@pytest.mark.parametrize("size,rate,geoms", [(224, 2.9, [(648, 0, 0), (300, 100, 249), (224, 424, 0)]),
                                             (32, 2.9, [(91, 0, 0), (40, 20, 51)])])
def test_dim_largest_ratio(size, rate, geoms):
    G.test_dim_random(size, rate, geoms)


def test_partials_registry_rules(widened_on_host, monkeypatch):
    W.test_partials_registry_rules(monkeypatch)


def test_bsr_kernels_golden(golden, widened_on_host, monkeypatch):
    W.test_bsr_kernels_golden(golden, monkeypatch)


@pytest.mark.parametrize("shape,nb,copies", [((2, 3, 224, 224), 3, 4), ((1, 3, 37, 41), 3, 3), ((2, 1, 16, 100), 2, 3),
                                             ((1, 2, 64, 64), 5, 2), ((1, 1, 9, 300), 1, 2), ((2, 2, 40, 40), 8, 2)])
def test_bsr_kernels_random(widened_on_host, monkeypatch, shape, nb, copies):
    W.test_bsr_kernels_random(shape, nb, copies, monkeypatch)


def test_full_size_property_tests_at_reduced_size(widened_on_host, monkeypatch):
    """the BASELINE-size property tests of the GPU tier, with the sizes cut to what the stand-in finishes quickly"""
    monkeypatch.setattr(W, "FULL_N", 12)
    monkeypatch.setattr(W, "SHARD_N", 5)
    W.test_update_full_size_properties()
    W.test_quantiser_full_size()
    W.test_transform_properties_at_shard_size()


@pytest.mark.parametrize("size,rate,geoms", [
    (224, 1.05, [(224, 0, 0), (234, 0, 0), (230, 3, 2)]),
    (224, 1.2, [(268, 0, 0), (240, 14, 28), (225, 43, 0)]),
    (100, 1.1, [(100, 5, 5), (109, 0, 1)]),
    (299, 1.1, [(300, 10, 20), (327, 0, 0)]),
    (40, 1.45, [(57, 0, 0), (41, 8, 16)]),
])
def test_dim_lane_kernels_other_geometries(size, rate, geoms):
    """tile widths, row counts (RPW 10 and 17) and hit-run lengths the default 224 / 1.1 setting does not reach"""
    G.test_dim_random(size, rate, geoms)


def test_kernels_under_reverse_lane_order(monkeypatch, golden, widened_on_host):
    """HIPCPU_ORDER=reverse runs the lanes of a workgroup last-to-first between barriers.  A kernel that needs a barrier
    it does not have (lanes of different waves meeting in LDS) answers differently under the two orders; all of them
    must give the oracle's bytes under both."""
    host_kernels.install(monkeypatch, tag="rev", env={"HIPCPU_ORDER": "reverse"})
    G.test_update_stack_golden(golden, "d09", 0.9, False)
    G.test_fused_update_random((5, 3, 37, 41))
    G.test_normalize_and_producer_side_partials((3, 3, 37, 41))
    G.test_tim_random((4, 3, 224, 224), 15)
    G.test_tim_random((2, 3, 50, 70), 9)
    G.test_dim_golden(golden)
    G.test_dim_random(224, 1.1, [(245, 0, 1), (237, 3, 5)])
    G.test_sim_admix_golden(golden)
    G.test_vmi_kernels_and_philox()
    G.test_dim_random(33, 2.0, [(40, 5, 20)])                  # the table-driven kernels
    G.test_producer_side_partials_all_kernels()
    W.test_sia_kernels_golden(golden)


@pytest.mark.parametrize("at_end", [True, False])
@pytest.mark.parametrize("shape", [(3, 3, 37, 41), (2, 1, 1, 7), (1, 3, 224, 224), (2, 3, 5, 6)])
def test_streaming_kernels_stay_in_bounds(shape, at_end):
    """every operand of the update / elementwise kernels -- inputs AND outputs -- flush against inaccessible pages:
    a vector access that runs past a ragged tail, before the first element or beyond the last faults"""
    from transferattack_amd import _hip
    gen = torch.Generator().manual_seed(sum(shape))
    keep = []

    def guarded(tensor):
        view, owner = _guarded(tensor.numpy(), at_end)
        keep.append(owner)
        return torch.from_numpy(view)

    n = shape[0]
    x = guarded(torch.rand(shape, generator=gen))
    grad = guarded(torch.randn(shape, generator=gen) * 1e-3)
    mom = guarded(torch.randn(shape, generator=gen))
    var = guarded(torch.randn(shape, generator=gen) * 1e-4)
    delta = guarded(torch.zeros(shape))
    xadv = guarded(torch.zeros(shape))
    out = guarded(torch.zeros(shape))
    _hip.mi_update(grad, mom, mom, delta, x, 0.9, 1.6 / 255, 16 / 255, variance=var, x_adv=xadv)
    _hip.momentum(grad, None, out, 1.0)
    _hip.update_delta_linf(delta, x, mom, 1.6 / 255, 16 / 255, out, x_adv=xadv)
    _hip.update_delta_linf(delta, x, mom, var.abs(), 16 / 255, out)
    _hip.update_delta_l2(delta, x, grad, 0.3, 3.0, out)
    c = shape[1]
    mean, std = torch.tensor([0.485, 0.456, 0.406][:c]), torch.tensor([0.229, 0.224, 0.225][:c])
    _hip.normalize_fwd(x, out, mean, std)
    _hip.normalize_bwd(grad, out, std)
    _hip.init_delta_uniform(delta, x, 16 / 255, seed=3, offset=1)
    _hip.vmi_neighbor(x, delta, out, 0.1, seed=3, offset=2)
    _hip.grad_accumulate(out, grad, first=True)
    _hip.grad_accumulate(out, grad, first=False)
    _hip.variance_finalize(out, grad, xadv, 20)
    _hip.axpy(x, mom, 0.01, out)
    copies = guarded(torch.zeros((5 * n,) + shape[1:]))
    _hip.scale_copies_fwd(x, copies, 5)
    _hip.scale_copies_bwd(copies, out, 5)
    _hip.sum_copies_bwd(copies, out, 5)
    mixed = guarded(torch.zeros((2 * 3 * n,) + shape[1:]))
    perm = torch.cat([torch.randperm(n, generator=gen) for _ in range(3)])
    _hip.admix_fwd(x, perm, mixed, 3, 2, 0.2)
    _hip.admix_bwd(mixed, out, 3, 2)
    u8 = torch.zeros((n,) + shape[2:] + (shape[1],), dtype=torch.uint8)
    _hip.quantize_u8_nhwc(x, delta, u8)
    del keep


@pytest.mark.parametrize("at_end", [True, False])
def test_sia_stays_in_bounds(at_end):
    from transferattack_amd import _hip
    from transferattack_amd.transforms import SIA_NOISE, sia_draw
    keep = []

    def guarded(array):
        view, owner = _guarded(np.ascontiguousarray(array), at_end)
        keep.append(owner)
        return torch.from_numpy(view)

    for shape, nb, copies in (((1, 2, 37, 41), 3, 3), ((1, 1, 9, 130), 2, 2)):
        np.random.seed(nb)
        torch.manual_seed(copies)
        plan, noise = sia_draw(shape, nb, copies, lambda s, lo, hi: torch.zeros(s).uniform_(lo, hi))
        x = torch.rand(shape)
        xg, plan_g, noise_g = guarded(x.numpy()), guarded(plan), guarded(noise.numpy())
        y = guarded(np.zeros((copies * shape[0],) + shape[1:], dtype=np.float32))
        _hip.sia_fwd(xg, plan_g, y, copies, nb, SIA_NOISE, noise=noise_g)
        assert np.array_equal(y.numpy(), C.sia_fwd(x.numpy(), plan, noise.numpy(), nb))
        gx = guarded(np.zeros(shape, dtype=np.float32))
        _hip.sia_bwd(y, plan_g, xg, gx, copies, nb, SIA_NOISE, noise=noise_g)
        assert np.array_equal(gx.numpy(), C.sia_bwd(y.numpy(), plan, x.numpy(), noise.numpy(), nb))


# ---- opt-in: |g| summed in the reference's (ATen cascade) order -> the momentum carries the reference's bits
@pytest.mark.parametrize("lanes", [8, 16])
def test_reference_order_sum(monkeypatch, golden, lanes):
    from transferattack_amd import _hip
    host_kernels.install(monkeypatch, tag="aten%d" % lanes, env={"TA_ATEN_SUM_LANES": str(lanes)})
    gen = torch.Generator().manual_seed(lanes)
    for shape in ((3, 3, 224, 224), (2, 3, 37, 41), (2, 1, 1, 7), (1, 3, 8, 8), (2, 2, 64, 65), (1, 1, 1, 1)):
        grad = torch.randn(shape, generator=gen)
        var = torch.randn(shape, generator=gen) * 0.1
        n, e = shape[0], grad[0].numel()
        for v in (None, var):
            ws = torch.full((int(_hip.load().ta_l1_workspace_floats(n, e)),), float("nan"))
            lib = _hip._sync_options(_hip.load())           # a raw call: push TA_ATEN_SUM_LANES through the ABI as the wrappers do
            rc = lib.ta_abs_sum_partials(grad.data_ptr(), None if v is None else v.data_ptr(), ws.data_ptr(), n, e, None)
            assert rc == 0
            tiles = ws.numel() // (2 * n)
            src = (grad if v is None else grad + v).abs().reshape(n, -1).numpy()
            for b in range(n):
                row = ws[b * tiles:(b + 1) * tiles].numpy()
                assert row[0] == C.aten_row_sum(src[b], lanes) and not row[1:].any()
    if lanes == 8:                                  # the goldens were written by an AVX2-order reference
        g = golden("update_stack")
        grad, mom, delta, x = (torch.from_numpy(g[k]) for k in ("grad", "momentum", "delta", "x"))
        for tag, decay, first in (("first", 1.0, True), ("d1", 1.0, False), ("d09", 0.9, False), ("d0", 0.0, False)):
            m = torch.empty_like(grad)
            _hip.momentum(grad, None if first else mom, m, decay)
            assert np.array_equal(m.numpy(), g["m_" + tag], equal_nan=True)            # bit for bit, not "close"
            d, m2 = delta.clone(), torch.empty_like(grad)
            _hip.mi_update(grad, None if first else mom.clone(), m2, d, x, decay, G.ALPHA, G.EPS)
            assert np.array_equal(m2.numpy(), g["m_" + tag], equal_nan=True)
            assert np.array_equal(d.numpy(), g["delta_" + tag])


def test_reference_sum_order_gpu_test_on_host(golden, monkeypatch, widened_on_host):
    W.test_reference_sum_order(golden, monkeypatch)




@pytest.mark.parametrize("shape", [(1, 1, 224, 224), (2, 1, 32, 32), (1, 2, 64, 64)])
def test_spectrum_kernel(widened_on_host, shape):
    W.test_spectrum_kernel(shape)
