import os
import sys

import numpy as np
import pytest

# torch ships its own copy of the HIP runtime; when libxritdemod_amd.so (linked against /opt/rocm) initialises
# HIP first, torch's copy then reports "No HIP GPUs are available".  Loading torch first makes both share one
# runtime.  Only the GPU tests that allocate through torch need it; the product itself never imports torch.
try:
    import torch  # noqa: F401
except ImportError:
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "experiments: needs a library built with make EXTRA=-DXRIT_EXPERIMENTS (the A/B "
                            "switches of DESIGN.md section 7 are not in the shipped build)")


def pytest_collection_modifyitems(config, items):
    """Tests of the measurement switches run only against a library that carries them."""
    try:
        import xritdemod_amd
        have = xritdemod_amd.build_experiments()
    except Exception:
        have = False
    if have:
        return
    skip = pytest.mark.skip(reason="library built without -DXRIT_EXPERIMENTS")
    for it in items:
        if "experiments" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build()
    return oracle


_SIG_CACHE = {}


def synth_signal(n, **kw):
    """Deterministic synthetic burst (numpy spec generator), cached per parameter set."""
    import synth
    key = (n, tuple(sorted(kw.items())))
    if key not in _SIG_CACHE:
        _SIG_CACHE[key] = synth.generate(synth.SynthParams(**kw), n)
    return _SIG_CACHE[key]


@pytest.fixture(scope="session")
def lrit_1m():
    return synth_signal(1 << 20)


def rms(a):
    a = np.asarray(a)
    return float(np.sqrt(np.mean(np.abs(a) ** 2))) if a.size else 0.0
