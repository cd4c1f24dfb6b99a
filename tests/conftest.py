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


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build()
    return oracle


_SIG_CACHE = {}


def synth_signal(n, **kw):
    """Deterministic synthetic burst (numpy spec generator), cached per parameter set."""
    from xritdemod_amd import synth
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
