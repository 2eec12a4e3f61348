import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")


def pytest_collection_modifyitems(config, items):
    """A hung GPU test (a rank waiting for a peer that died, a spinning kernel) must not eat the GPU box's time budget:
    every test gets a wall-clock limit when pytest-timeout is installed."""
    if config.pluginmanager.hasplugin("timeout"):
        for it in items:
            if it.get_closest_marker("timeout") is None:
                it.add_marker(pytest.mark.timeout(420 if "headline" in it.nodeid else 240))


def load_golden(name):
    """Load tests/golden/<name>.npz (fixtures produced by oracle/make_golden.py from the reference)."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def sub(d, prefix):
    p = prefix + "/"
    return {k[len(p):]: v for k, v in d.items() if k.startswith(p)}


@pytest.fixture(scope="session")
def oracle():
    from oracle import xrl_oracle
    return xrl_oracle


_PARITY_LOG = []          # (test id, what, err / tensor scale, err / max(1,|ref|), tol) of every comparison of the session


def _record(what, err_tensor, err_legacy, tol, n, **extra):
    _PARITY_LOG.append({"test": os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0], "what": what,
                        "err": float(err_tensor), "err_legacy": float(err_legacy), "tol": float(tol), "n": int(n), **extra})


def pytest_sessionfinish(session, exitstatus):
    """XRL_PARITY_REPORT=<path>: every comparison's measured error (relative to the tensor's own scale) as JSON lines --
    the committed record behind the tolerances (profiles/r03_parity_errors_*.jsonl)."""
    path = os.environ.get("XRL_PARITY_REPORT")
    if path and _PARITY_LOG:
        import json
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, "a") as fh:
            for r in _PARITY_LOG:
                fh.write(json.dumps(r) + "\n")


def assert_close(a, b, tol=1e-5, what="", scale=None):
    """max|a-b| <= tol * S over the whole tensor, S = the tensor's OWN scale max|b| -- never max(1, |b|): a gradient of
    magnitude 1e-3 or an Adam second moment of magnitude 1e-9 is checked to 1e-5 of itself, not of 1.

    ``scale`` replaces S only where the quantity is a sum of operands larger than itself, and then it is that operand
    magnitude, named at the call site (a loss that is the mean of O(1) terms cancelling to 1e-4; a Gaussian log-prob of
    magnitude ~50 carries an fp32 rounding floor of ~50 * 2^-23 * few: the reference itself sits 8.6e-6 from a float64
    evaluation there).  XRL_PARITY_LEGACY=1 asserts round 2's looser elementwise rule instead (|a-b| <= tol*max(1,|b|));
    it exists to collect the error record of a full run in one go, not for the driver's runs."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    assert np.all(np.isfinite(a)), f"{what}: non-finite values"
    diff = np.abs(a - b)
    S = float(scale) if scale is not None else float(np.abs(b).max(initial=0.0))
    err = float(diff.max(initial=0.0)) / S if S > 0 else float(diff.max(initial=0.0))
    legacy = float((diff / np.maximum(1.0 if scale is None else max(1.0, float(scale)), np.abs(b))).max(initial=0.0))
    _record(what, err, legacy, tol, a.size)
    if os.environ.get("XRL_PARITY_LEGACY") == "1":
        assert legacy <= tol, f"{what}: legacy max err {legacy:.3e} > {tol}"
        return
    assert err <= tol, f"{what}: max |a-b| / scale = {err:.3e} > {tol} (scale {S:.3e})"


def assert_step_close(before, after, ref_before, ref_after, tol=1e-5, what=""):
    """Parameter STEPS (after - before) compared at the scale of the step, not of the parameter: a parameter of
    magnitude 0.1 moves by <= lr = 4e-4 per update, so comparing parameters at 1e-5 of their own scale would check the
    Adam step to a few per cent only.  The stored fp32 parameter quantises the step (two exact implementations may land
    on neighbouring floats), hence the one-ulp allowance: |da - db| <= tol * max|db| + ulp32(|param|)."""
    before, after = np.asarray(before, np.float32), np.asarray(after, np.float32)
    ref_before, ref_after = np.asarray(ref_before, np.float32), np.asarray(ref_after, np.float32)
    da = after.astype(np.float64) - before.astype(np.float64)
    db = ref_after.astype(np.float64) - ref_before.astype(np.float64)
    S = float(np.abs(db).max(initial=0.0))
    ulp = np.spacing(np.maximum(np.abs(ref_after), np.abs(ref_before))).astype(np.float64)
    excess = np.maximum(np.abs(da - db) - ulp, 0.0)
    err = float(excess.max(initial=0.0)) / S if S > 0 else float(excess.max(initial=0.0))
    _record("step " + what, err, float(np.abs(after.astype(np.float64) - ref_after).max(initial=0.0)), tol, da.size)
    if os.environ.get("XRL_PARITY_LEGACY") == "1":
        return
    assert err <= tol, f"step {what}: max (|da-db| - ulp) / max|db| = {err:.3e} > {tol} (max|db| {S:.3e})"


def assert_grad_close(a, ref32, ref64=None, tol=1e-5, what=""):
    """A gradient tensor against the reference's (float32) one at the tensor's own scale -- and, where the reference's float32
    evaluation is itself further than `tol` from the exact value, against the reference's float64 twin instead (fixtures
    `u*/grad64/*`: the reference's learner on model.double(), oracle/make_golden.py: float64_twin): a weight gradient summed
    over 8 192 rows with heavy cancellation (critic.values.0.weight of the C2 fixture: |sum| / sum|terms| ~ 1e-2) carries
    sgemm's float32 summation noise -- the reference sits 1.5e-4 of the tensor's scale from its float64 twin there, and no
    other summation order can reproduce that noise.  Rule: within tol of the float32 reference, OR no further from the float64
    twin than the float32 reference is (and never worse than that).  Returns the absolute error bound this tensor was held to
    (the parameter-step check propagates it through Adam)."""
    a, ref32 = np.asarray(a, np.float64), np.asarray(ref32, np.float64)
    assert a.shape == ref32.shape, f"{what}: shape {a.shape} vs {ref32.shape}"
    assert np.all(np.isfinite(a)), f"{what}: non-finite values"
    S = float(np.abs(ref32).max(initial=0.0)) or 1.0
    d32 = float(np.abs(a - ref32).max(initial=0.0)) / S
    if ref64 is None:
        _record(what, d32, d32, tol, a.size)
        if os.environ.get("XRL_PARITY_LEGACY") != "1":
            assert d32 <= tol, f"{what}: max |a-ref| / max|ref| = {d32:.3e} > {tol}"
        return tol * S
    ref64 = np.asarray(ref64, np.float64)
    d64 = float(np.abs(a - ref64).max(initial=0.0)) / S
    r64 = float(np.abs(ref32 - ref64).max(initial=0.0)) / S            # the reference's own float32 error
    # ONE record per comparison, saying which clause decided: "f32" = within tol of the reference's float32 gradient (then the float64
    # numbers are information only: the two float64 distances of an update later than the first also contain the distance between the
    # float32 and float64 TRAJECTORIES); "f64" = an exception to the 1e-5 float32 rule, listed by tools/compact_parity_report.py
    decided = "f32" if d32 <= tol else "f64"
    _record(what + (" [vs f32 ref]" if decided == "f32" else " [EXCEPTION: vs f64 twin; vs f32 ref %.3e; the f32 reference itself: %.3e]" % (d32, r64)),
            d32 if decided == "f32" else d64, d32, tol if decided == "f32" else max(tol, r64), a.size,
            decided_by=decided, d32=d32, d64=d64, ref32_vs_ref64=r64)
    if os.environ.get("XRL_PARITY_LEGACY") != "1":
        assert d32 <= tol or d64 <= max(tol, r64), \
            f"{what}: {d32:.3e} from the float32 reference, {d64:.3e} from its float64 twin (the reference itself: {r64:.3e}), tol {tol}"
    return max(tol, r64) * S


class AdamReplay:
    """float64 restatement of what the reference's optimiser does to one parameter set, fed with the REFERENCE's gradients of
    each update (torch.optim.Adam, single-tensor form: exp_avg.lerp_(g, 1 - b1); exp_avg_sq.mul_(b2).addcmul_(g, g, 1 - b2);
    denom = sqrt(exp_avg_sq) / sqrt(1 - b2^t) + eps; p -= lr_t / (1 - b1^t) * exp_avg / denom; LinearLR(start 1, end_factor,
    total_iters) stepped after every update, ppo_learner.py:19-22,63-67).  It yields, per update and element, the step the
    reference takes and the CONDITIONING of that step: if every gradient of the history is known to +-d (per tensor), then
    |d exp_avg| <= d (1 - b1^t), |d sqrt(exp_avg_sq)| <= d sqrt(1 - b2^t) (triangle inequality of the weighted 2-norm), hence
        |d step_i| <= lr_t * d * (1 / denom_i + |m_hat_i| / denom_i^2)        (first order in d / denom).
    Entries whose gradient sits at the level of eps = 1e-5 move by ~lr whatever their gradient's sign: two evaluations that
    agree to 1e-5 of the gradient's scale may step them differently by a large fraction of lr -- the bound says by how much."""

    def __init__(self, lr, eps=1e-5, end_factor=1.0, total_iters=1, betas=(0.9, 0.999), weight_decay=0.0):
        self.lr0, self.eps, self.ef, self.total = float(lr), float(eps), float(end_factor), max(int(total_iters), 1)
        self.b1, self.b2, self.wd = float(betas[0]), float(betas[1]), float(weight_decay)
        self.t, self.m, self.v, self.gmax = 0, {}, {}, {}

    def lr(self):
        return self.lr0 * (1.0 + (self.ef - 1.0) * min(self.t, self.total) / self.total)

    def step(self, grads, params=None):
        """grads: name -> the reference's gradient of this update (after clipping).  -> name -> (step, sensitivity)."""
        lr_t = self.lr()
        self.t += 1
        bc1, bc2 = 1.0 - self.b1 ** self.t, 1.0 - self.b2 ** self.t
        out = {}
        for n, g in grads.items():
            g = np.asarray(g, np.float64)
            if self.wd:
                g = g + self.wd * np.asarray(params[n], np.float64)
            m = self.m.get(n, 0.0) * self.b1 + (1.0 - self.b1) * g
            v = self.v.get(n, 0.0) * self.b2 + (1.0 - self.b2) * g * g
            self.m[n], self.v[n] = m, v
            self.gmax[n] = max(self.gmax.get(n, 0.0), float(np.abs(g).max(initial=0.0)))
            den = np.sqrt(v) / np.sqrt(bc2) + self.eps
            mhat = m / bc1
            out[n] = (lr_t * mhat / den, lr_t * (1.0 / den + np.abs(mhat) / den ** 2))
        return out


class LearnerFixtureCheck:
    """One learner fixture (oracle/make_golden.py: run_learner_updates) replayed against an engine: per update the gradients
    (assert_grad_close: the tensor's own scale, float64-anchored where the fixture has a twin), the parameter STEPS (at the
    step's scale, through AdamReplay's conditioning, + one ulp of the stored float32 parameter), and at the end Adam's moments
    (exp_avg: linear in the gradients, same bound; exp_avg_sq: |d v_i| <= 2 d sqrt(1 - b2^t) sqrt(v_i))."""

    def __init__(self, g, init, lr, eps=1e-5, end_factor=1.0, total_iters=1, weight_decay=0.0, tol=1e-5, init_extra=None,
                 tol_except=None):
        """tol: the bound of every tensor; tol_except: {tensor name: its own bound} for NAMED, explained exceptions (nothing else
        rides on them).  `rows/<name>` entries of the fixture: only those rows of that (large) tensor are stored -- the engine's tensor is cut to
        the same rows wherever it is compared; init_extra: the reference's initial values of tensors the fixture rebuilds from
        a formula instead of storing them (oracle/make_golden.py: golden_ppo_cnn)."""
        self.g, self.tol, self.tol_except = g, tol, dict(tol_except or {})
        self.rows = {k[len("rows/"):]: np.asarray(v) for k, v in g.items() if k.startswith("rows/")}
        self.adam = AdamReplay(lr, eps, end_factor, total_iters, weight_decay=weight_decay)
        self.before = {k: self._sl(k, np.asarray(v, np.float32)).copy() for k, v in init.items()}   # the ENGINE's parameters
        self.ref_before = {k: np.asarray(v, np.float32) for k, v in sub(g, "init").items()}
        self.ref_before.update({k: self._sl(k, np.asarray(v, np.float32)) for k, v in (init_extra or {}).items()})
        self.delta = {}                                                                      # name -> gradient error bound held
        self.replay_checked = 0
        self.hist, self.ref_hist = [dict(self.before)], [dict(self.ref_before)]              # parameter sets: init, after u0, ...

    def _sl(self, name, arr):
        return arr[self.rows[name]] if name in self.rows else arr

    def _tol(self, name):
        return self.tol_except.get(name, self.tol)

    def update(self, u, grads, params_after):
        g = self.g
        grads = {k: self._sl(k, np.asarray(v)) for k, v in grads.items()}
        params_after = {k: self._sl(k, np.asarray(v)) for k, v in params_after.items()}
        ref_g, ref_g64 = sub(g, f"u{u}/grad"), sub(g, f"u{u}/grad64")
        ref_after = sub(g, f"u{u}/param")
        for n, rg in ref_g.items():
            d = assert_grad_close(grads[n], rg, ref_g64.get(n), self._tol(n), f"grad {n} (update {u})")
            self.delta[n] = max(self.delta.get(n, 0.0), d)
        steps = self.adam.step(ref_g, self.ref_before)
        for n, rp in ref_after.items():
            after = np.asarray(params_after[n], np.float32)
            if n not in ref_g:
                # no gradient in the reference: a target copy or a frozen tensor.  The reference's value is bit-equal to some
                # tensor of an earlier (or this) parameter set -- the engine's must be bit-equal to ITS tensor of that set
                # (the copy itself is exact; how far that source is from the reference is checked where it was stepped)
                src = None
                ref_sets = self.ref_hist + [ref_after]
                for v in range(len(ref_sets) - 1, -1, -1):
                    for m_ in [n] + [k for k in ref_sets[v] if k != n]:
                        if m_ in ref_sets[v] and ref_sets[v][m_].shape == rp.shape and np.array_equal(ref_sets[v][m_], rp) \
                                and (m_ in ref_g or v < len(ref_sets) - 1):
                            src = (v, m_)
                            break
                    if src:
                        break
                assert src is not None, f"param {n} after update {u}: the reference's value is no copy of any known tensor"
                eng_sets = self.hist + [params_after]
                assert np.array_equal(after, np.asarray(eng_sets[src[0]][src[1]], np.float32)), \
                    f"param {n} after update {u} is not the copy of {src[1]} (parameter set {src[0]}) it is in the reference"
                _record(f"copy {n} (update {u})", 0.0, 0.0, 0.0, after.size)
                continue
            step, sens = steps[n]
            db = rp.astype(np.float64) - self.ref_before[n].astype(np.float64)
            ulp = np.spacing(np.maximum(np.abs(rp), np.abs(self.ref_before[n]))).astype(np.float64)
            # the replay itself is pinned by the reference: its float32 parameters moved by the replayed step (to an ulp)
            # (torch keeps exp_avg / exp_avg_sq in float32: 2e-6 of the tensor's largest step covers their rounding)
            assert np.all(np.abs(db + step) <= 2.0 * ulp + 2e-6 * np.abs(step).max()), f"AdamReplay does not reproduce the reference's step of {n}"
            self.replay_checked += 1
            da = after.astype(np.float64) - self.before[n].astype(np.float64)
            S = float(np.abs(db).max(initial=0.0)) or 1.0
            allowed = self._tol(n) * S + sens * self.delta[n] + 2.0 * ulp
            excess = float(np.max((np.abs(da - db) - allowed) / S, initial=-1.0))
            _record(f"step {n} (update {u}) [max |da-db|/max|db| = {np.abs(da - db).max() / S:.3e}]", max(excess, 0.0) , 0.0, 0.0, da.size)
            if os.environ.get("XRL_PARITY_LEGACY") != "1":
                assert excess <= 0.0, (f"step of {n} in update {u}: |da-db| exceeds tol*max|db| + (Adam conditioning x gradient bound) + 2 ulp "
                                       f"by {excess:.3e} of max|db| = {S:.3e}")
        self.before = {k: np.asarray(v, np.float32).copy() for k, v in params_after.items()}
        self.ref_before = {k: np.asarray(v, np.float32) for k, v in ref_after.items()}
        self.hist.append(dict(self.before))
        self.ref_hist.append(dict(self.ref_before))

    def moments(self, exp_avg, exp_avg_sq):
        """name -> engine tensors after the last update, against `adam/exp_avg[_sq]/<name>` of the fixture."""
        bc2 = 1.0 - self.adam.b2 ** self.adam.t
        exp_avg = {k: self._sl(k, np.asarray(v)) for k, v in exp_avg.items()}
        exp_avg_sq = {k: self._sl(k, np.asarray(v)) for k, v in exp_avg_sq.items()}
        for n, a in exp_avg.items():
            if f"adam/exp_avg/{n}" not in self.g:
                continue
            G = max(self.adam.gmax.get(n, 0.0), 1e-30)
            d = self.delta.get(n, self._tol(n) * G) / G
            assert_close(a, self.g[f"adam/exp_avg/{n}"], max(self._tol(n), d), f"exp_avg {n}", scale=G)
            rv = self.g[f"adam/exp_avg_sq/{n}"]
            assert_close(exp_avg_sq[n], rv, 2.0 * max(self._tol(n), d), f"exp_avg_sq {n}",
                         scale=G * np.sqrt(bc2) * float(np.sqrt(np.abs(rv).max(initial=0.0))) or 1.0)


class EngineFixtureCheck(LearnerFixtureCheck):
    """LearnerFixtureCheck fed from a xuance_amd learner: gradients = views of the flat (clipped, reduced) gradient buffer the
    optimiser launch leaves behind, parameters = net.state_dict(), moments = learner.optimizer.state_dict()."""

    def __init__(self, g, net, learner, lr, end_factor=1.0, total_iters=1, weight_decay=0.0, tol=1e-5, state_source=None,
                 init_extra=None, tol_except=None):
        self.net, self.learner = net, learner
        self.state_source = net if state_source is None else state_source       # (seam tests: the caller's own nn.Module)
        super().__init__(g, self._params(), lr, end_factor=end_factor, total_iters=total_iters, weight_decay=weight_decay, tol=tol,
                         init_extra=init_extra, tol_except=tol_except)

    def _params(self):
        return {k: v.detach().cpu().numpy().copy() for k, v in self.state_source.state_dict().items()}

    def after_update(self, u):
        names = list(sub(self.g, f"u{u}/grad"))
        grads = {k: self.net.params.view(k, self.learner.optimizer.grad).cpu().numpy() for k in names}
        self.update(u, grads, self._params())

    def finish(self, order=None):
        osd = self.learner.optimizer.state_dict()
        order = list(self.net.ref_order if order is None else order)
        m, v = {}, {}
        for i, k in enumerate(order):
            if f"adam/exp_avg/{k}" not in self.g:            # the reference never produced a gradient for this tensor
                if i in osd["state"]:
                    assert not osd["state"][i]["exp_avg"].any() and not osd["state"][i]["exp_avg_sq"].any(), k
                continue
            m[k], v[k] = osd["state"][i]["exp_avg"].cpu().numpy(), osd["state"][i]["exp_avg_sq"].cpu().numpy()
        assert m, "no Adam moments compared"
        self.moments(m, v)


class ChainCheck:
    """Parameters after a short chain of updates the engine ran in one go (a captured update phase), against a float32 oracle
    that replayed the same minibatches: only the end state is visible, so the allowance is the SUM over the chain's updates of
    what one update may differ by when every gradient agrees to `tol` of its tensor's scale -- tol * max|step| + Adam's
    conditioning (AdamReplay) * tol * max|grad| -- plus an ulp of the stored parameter per update.  Feed it the oracle's clipped
    gradients of every update (oracle.ppo_update -> info["clipped_grads"])."""

    def __init__(self, lr, tol=1e-5, **adam_kwargs):
        self.adam, self.tol, self.allow = AdamReplay(lr, **adam_kwargs), tol, {}

    def step(self, grads):
        for n, (step, sens) in self.adam.step(grads).items():
            self.allow[n] = self.allow.get(n, 0.0) + self.tol * float(np.abs(step).max(initial=0.0)) + sens * self.tol * self.adam.gmax[n]

    def check(self, got, ref_after, init, what="param"):
        for n, ref in ref_after.items():
            a, r, i0 = (np.asarray(x[n], np.float32).astype(np.float64) for x in (got, ref_after, init))
            if n not in self.allow:
                assert np.array_equal(a, r), f"{what} {n}: never stepped in the oracle but differs"
                continue
            ulp = np.spacing(np.maximum(np.abs(np.asarray(ref, np.float32)), np.abs(np.asarray(init[n], np.float32)))).astype(np.float64)
            S = float(np.abs(r - i0).max(initial=0.0)) or 1.0
            excess = float(np.max((np.abs(a - r) - self.allow[n] - (self.adam.t + 1) * ulp) / S, initial=-1.0))
            _record(f"chain {what} {n} ({self.adam.t} updates) [max |a-ref| / max|moved| = {np.abs(a - r).max() / S:.3e}]",
                    max(excess, 0.0), float(np.abs(a - r).max()), 0.0, a.size)
            if os.environ.get("XRL_PARITY_LEGACY") != "1":
                assert excess <= 0.0, f"{what} {n} after {self.adam.t} updates: beyond the propagated bound by {excess:.3e} of the distance moved ({S:.3e})"


def free_port():
    """A TCP port nobody listens on right now (rendezvous of the multi-process tests: a fixed, pid-derived port could
    still be held by the previous test's process group)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]
