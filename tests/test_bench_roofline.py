"""CPU: bench.py's ``roofline`` object from a synthetic record of update launches -- which bytes each field prices (review r4:
the headline ``frac`` must be the bytes the kernel requests, not an algorithmic bill the instantiation does not move)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


class _Evt:
    def __init__(self, t_ms):
        self.t = t_ms

    def elapsed_time(self, other):
        return other.t - self.t


class _Hip:
    stats = {"k1_passes": 0, "partials_reused": 10}


def _sink(n, e, us_steady, us_first):
    """K = 10 launches of the default loop: the first has no momentum to read (20 B/element algorithmic), the rest 24; all take
    the image from the byte source and the std form"""
    recs = [(_Evt(0.0), _Evt(us_first / 1e3 + 0.003), n, e, 20, True, True)]
    recs += [(_Evt(0.0), _Evt(us_steady / 1e3 + 0.003), n, e, 24, True, True) for _ in range(9)]
    return recs, [us_first / 1e3] + [us_steady / 1e3] * 9


def test_roofline_prices_executed_bytes():
    n, e = 125, 3 * 224 * 224
    sink, dispatch = _sink(n, e, 64.7, 54.2)
    r = bench.roofline(None, sink, dispatch, None, _Hip, byte_source_taken=True)
    steady_exec, steady_alg = 21 * n * e, 24 * n * e
    assert r["steady_state_launch"]["executed_bytes"] == steady_exec and r["steady_state_launch"]["algorithmic_bytes"] == steady_alg
    total_exec = 17 * n * e + 9 * steady_exec
    total_us = 54.2 + 9 * 64.7
    assert abs(r["achieved"] - total_exec / total_us / 1e3) < 0.2                      # GB/s over all launches, executed bytes
    assert r["frac"] == r["executed"]["frac"] == round(r["achieved"] / 8000.0, 4)
    assert abs(r["frac_at_24B_contract"] - steady_alg / 64.7 / 1e3 / 8000.0) < 1e-3    # 24 B/element over the steady launches
    assert r["frac_algorithmic"] > r["frac"] and r["std_form_launches"] == 10 and r["byte_source_launches"] == 10
    assert r["clock"].startswith("HIP events bound") and r["k1_passes"] == 0
    # without the byte source the executed bytes ARE the algorithmic ones
    r2 = bench.roofline(None, sink, dispatch, None, _Hip, byte_source_taken=False)
    assert r2["frac"] == r2["frac_algorithmic"] and r2["byte_source_launches"] == 0
    # an inconsistent dispatch clock falls back to the markers and says so
    r3 = bench.roofline(None, sink, [d * 10 for d in dispatch], None, _Hip, byte_source_taken=True)
    assert r3["clock"].startswith("hipEventRecord")


def test_roofline_bills_the_sum_pass_to_the_launches_that_ran_one():
    """records of round 6 carry, per launch, whether the call ran its own sum pass over the gradient (the module path's backward
    leaves no sums): those launches request 4 B/element more, the others do not -- whatever the process-wide counter says"""
    n, e = 32, 3 * 224 * 224
    sink, dispatch = _sink(n, e, 30.0, 26.0)
    with_k1 = [rec + (True,) for rec in sink]
    r = bench.roofline(None, with_k1, dispatch, None, _Hip, byte_source_taken=True)
    assert r["k1_passes"] == 10 and r["k1_pass_skipped_launches"] == 0
    assert r["steady_state_launch"]["executed_bytes"] == 25 * n * e
    mixed = [rec + (i % 2 == 0,) for i, rec in enumerate(sink)]
    r2 = bench.roofline(None, mixed, dispatch, None, _Hip, byte_source_taken=True)
    assert r2["k1_passes"] == 5 and r["pricing"] == r2["pricing"] == "executed-bytes/2"
    none = [rec + (False,) for rec in sink]
    assert bench.roofline(None, none, dispatch, None, _Hip, byte_source_taken=True)["frac"] == \
        bench.roofline(None, sink, dispatch, None, _Hip, byte_source_taken=True)["frac"]


def test_committed_pmc_file_is_the_newest_rounds():
    """profiles/pmc_update_kernel.json -- which bench.py reads for ``roofline.traffic`` -- must be the file the newest round's
    PMC run produced (tools/gpu_check.sh pmc writes both): a copy that drifts from its source fails here"""
    import glob
    import json
    rounds = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]", "pmc_update_kernel*.json")))
    assert rounds, "no per-round PMC summary under profiles/rNN/"
    newest_dir = os.path.dirname(rounds[-1])
    newest = sorted(p for p in rounds if os.path.dirname(p) == newest_dir)[-1]
    top = json.load(open(os.path.join(ROOT, "profiles", "pmc_update_kernel.json")))
    src = json.load(open(newest))
    assert top["kernels"] == src["kernels"] and top["fetch_correction"] == src["fetch_correction"], \
        "profiles/pmc_update_kernel.json differs from %s" % os.path.relpath(newest, ROOT)
