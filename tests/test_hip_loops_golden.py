"""GPU (-m gpu): loop-level goldens for the branches of the base hooks that the plain L-inf / scalar-step loops never reach
(review rows a3 / a4): the REAL reference ran whole attacks on the toy CNN with every hook call recorded
(oracle/gen_golden.py::gen_loops_hooks -> tests/golden/loops_hooks.npz):

    mifgsm_l2            update_delta's L2 branch                          transferattack/attack.py:148-151
    mifgsm_l2_random     + init_delta's L2 random start                    transferattack/attack.py:136-140
    mifgsm_linf_random   init_delta's uniform random start + box clamp     transferattack/attack.py:133-134,141
    gra                  update_delta with a TENSOR step M * alpha         transferattack/gradient/gra.py:149
    cwa                  update_delta with a NEGATIVE step, random start   transferattack/ensemble/cwa.py:69

Two checks per loop.  (1) hook by hook: every recorded call of init_delta / get_momentum / update_delta is repeated through
the product's hook on the device with the recorded inputs -- the L-inf results must carry the reference's bits (scalar,
negative and tensor step alike), the momentum is within the summation-order bound of conftest.assert_momentum_close, the L2
results within 4 ulp of the ball's radius (the per-image norm is summed in another order).  (2) the loop: the product's own
``forward`` runs on the device with the recorded gradients replayed call by call and the reference's CPU draws injected; the
final delta must equal the reference's (bit for bit where every step is sign-driven; see ``LOOP_EXACT``).

tests/test_attack_loops_host.py runs the same functions on the CPU through the host build of the kernels."""
import numpy as np
import pytest
import torch

import transferattack_amd as ta
from conftest import assert_momentum_close
from transferattack_amd import backbones
from transferattack_amd.utils import EnsembleModel, wrap_model

pytestmark = pytest.mark.gpu
DEV = "cuda"

CASES = {
    "mifgsm_l2": ("mifgsm", dict(norm="l2", epsilon=3.0, alpha=0.6), 1),
    "mifgsm_l2_random": ("mifgsm", dict(norm="l2", epsilon=3.0, alpha=0.6, random_start=True), 1),
    "mifgsm_linf_random": ("mifgsm", dict(random_start=True), 1),
    "gra": ("gra", dict(num_neighbor=2, epoch=5), 1),
    "cwa": ("cwa", dict(epoch=4), 2),
}
# loops whose every step is sign(m) * step: replayed gradients give the reference's delta bit for bit.  GRA blends two
# gradients by a cosine similarity and CWA steps along an L2-normalised momentum: their device reductions add in another
# order, the last bit of the blend / step moves, and an element may land on the other side of a clamp or a sign.
LOOP_EXACT = {"mifgsm_linf_random"}


def load(golden, tag):
    g = golden("loops_hooks")

    def get(key):
        v = g[key]
        return get(str(v)) if v.dtype.kind in "US" else v          # a stored key = "same bits as that earlier tensor"

    calls = []
    for k, hook in enumerate(g[tag + ".hooks"]):
        prefix = "%s.%d." % (tag, k)
        calls.append((str(hook), {key[len(prefix):]: get(key) for key in g.files if key.startswith(prefix)}))
    x = torch.from_numpy(g["x_u8"]).float() / 255
    return x, torch.from_numpy(g["label"]), calls, g[tag + ".delta"], int(g["seed"])


def make(tag):
    name, kw, members = CASES[tag]
    base = ta.load_attack_class(name)
    nets = [backbones.create("toy_cnn", seed=3 + i, verbose=False) for i in range(members)]

    def load_model(self, model_name):
        wrapped = [wrap_model(m.eval().to(DEV)) for m in nets]
        return wrapped[0] if members == 1 else EnsembleModel(wrapped)

    cls = type("Gpu" + base.__name__, (base,), {"load_model": load_model})
    atk = cls(model_name=["a", "b"] if members > 1 else "injected", **kw)
    atk.noise_source = lambda shape, lo, hi: torch.zeros(shape).uniform_(lo, hi)         # the reference's CPU draws ...
    atk.normal_source = lambda shape, mean, std: torch.zeros(shape).normal_(mean, std)    # ... in the reference's order
    return atk


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def close_l2(got, ref, eps, what):
    """L2 branch: grad / (norm + 1e-20), renorm to the eps-ball -- the norms are device reductions"""
    err = float(np.abs(got.astype(np.float64) - ref).max())
    bound = 4 * 2.0 ** -23 * max(eps, float(np.abs(ref).max()))
    assert err <= bound, "%s: off by %.3e (bound %.3e)" % (what, err, bound)
    return err


@pytest.mark.parametrize("tag", list(CASES))
def test_recorded_hook_calls(golden, tag):
    x, label, calls, _, seed = load(golden, tag)
    atk = make(tag)
    l2 = atk.norm == "l2"
    xd = x.to(DEV)
    torch.manual_seed(seed)
    seen = {"init_delta": 0, "get_momentum": 0, "update_delta": 0}
    steps, worst = set(), 0.0
    for hook, a in calls:
        if hook == "init_delta":
            got = atk.init_delta(xd)
            assert got.requires_grad and got.is_leaf
            if l2:
                worst = max(worst, close_l2(got.detach().cpu().numpy(), a["out"], atk.epsilon, "init_delta"))
            else:
                assert np.array_equal(got.detach().cpu().numpy(), a["out"]), "init_delta differs from the reference's"
        elif hook == "get_momentum":
            m_prev = a.get("momentum")
            got = atk.get_momentum(dev(a["grad"]), dev(m_prev) if m_prev is not None else 0)
            assert_momentum_close(got.cpu().numpy(), a["out"], a["grad"], m_prev, atk.decay)
        elif hook == "update_delta":
            alpha = a["alpha"]
            step = dev(alpha) if alpha.ndim else float(alpha)
            steps.add("tensor" if alpha.ndim else ("negative" if float(alpha) < 0 else "positive"))
            got = atk.update_delta(dev(a["delta"]).requires_grad_(True), xd, dev(a["grad"]), step)
            assert got.requires_grad and got.is_leaf                       # attack.py:153: a fresh leaf
            if l2:
                worst = max(worst, close_l2(got.detach().cpu().numpy(), a["out"], atk.epsilon, "update_delta"))
            else:
                assert np.array_equal(got.detach().cpu().numpy(), a["out"]), "update_delta differs from the reference's"
        else:
            continue
        seen[hook] += 1
    print("%s: %s; step kinds %s%s" % (tag, seen, sorted(steps), "; largest L2-branch deviation %.2e" % worst if l2 else ""))
    assert seen["init_delta"] == 1 and seen["update_delta"] >= 4
    if tag == "gra":
        assert steps == {"tensor"}
    if tag == "cwa":
        assert steps == {"negative", "positive"}


@pytest.mark.parametrize("tag", list(CASES))
def test_loop_with_replayed_gradients(golden, tag):
    x, label, calls, delta_ref, seed = load(golden, tag)
    atk = make(tag)
    grads = [a["out"] for hook, a in calls if hook == "get_grad"]
    it = [0]
    orig = type(atk).get_grad

    def get_grad(self, loss, delta, **kw):
        own = orig(self, loss, delta, **kw)                   # the device's own backward still runs (and is discarded)
        assert own.shape == delta.shape
        it[0] += 1
        return dev(grads[it[0] - 1])

    type(atk).get_grad = get_grad
    torch.manual_seed(seed)
    delta = atk(x, label).cpu().numpy()
    assert it[0] == len(grads), "the loop asked for %d gradients, the reference's for %d" % (it[0], len(grads))
    diff = np.abs(delta.astype(np.float64) - delta_ref)
    print("%s: %d replayed gradients; final delta vs the reference's: %d of %d elements differ, max |diff| %.3e"
          % (tag, len(grads), int((delta != delta_ref).sum()), delta.size, float(diff.max())))
    if tag in LOOP_EXACT:
        assert np.array_equal(delta, delta_ref)
    elif atk.norm == "l2":
        assert float(diff.max()) <= 64 * 2.0 ** -23 * atk.epsilon          # K = 10 steps of <= 4 ulp each, not amplified
    else:
        assert float((delta != delta_ref).mean()) <= 2e-3 and float(diff.max()) <= 2 * atk.epsilon
