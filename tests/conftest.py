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


def assert_close(a, b, tol=1e-5, what="", scale=1.0):
    """|a-b| <= tol * max(scale, |b|) elementwise -- the north-star's 1e-5 fp32 bar.

    ``scale`` is the magnitude of the operands a quantity was accumulated from when that is larger
    than the quantity itself (e.g. a Gaussian log-prob of magnitude ~50 carries an fp32 rounding
    floor of ~50 * 2^-23 * few: the reference itself sits 8.6e-6 from a float64 evaluation there)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    err = np.abs(a - b) / np.maximum(float(scale), np.abs(b))
    assert np.all(np.isfinite(a)), f"{what}: non-finite values"
    assert err.max(initial=0.0) <= tol, f"{what}: max err {err.max():.3e} > {tol}"


def free_port():
    """A TCP port nobody listens on right now (rendezvous of the multi-process tests: a fixed, pid-derived port could
    still be held by the previous test's process group)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]
