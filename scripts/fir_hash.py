"""Hashes of the FIR stage's outputs (decimations 16 / 32 / 64 take the polyphase kernel; 5 and 1 the windowed one) on fixed random
input, several calls with history: two library builds that print the same lines compute the same words.  python scripts/fir_hash.py"""
import hashlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import xritdemod_amd as xa
from xritdemod_amd import _capi
for D, fs, nt in [(16, 20e6, None), (32, 40e6, None), (64, 80e6, None), (32, 40e6, 3), (5, 6.25e6, None), (1, 1.25e6, None)]:
    taps = np.asarray(_capi.Filters.lowPass(1.0, fs, 625e3, 100e3) if nt is None else [0.5, -0.25, 0.125], np.float32)
    rng = np.random.default_rng(100 + D)
    n_out = [300001, 1, 777, 0, 123457]
    x = (rng.standard_normal(sum(n_out) * D) + 1j * rng.standard_normal(sum(n_out) * D)).astype(np.complex64)
    f, pos, h = xa.FirFilter(D, taps), 0, hashlib.sha256()
    for n in n_out:
        y = f.Work(x[pos:pos + n * D], n)
        pos += n * D
        h.update(np.ascontiguousarray(y).tobytes())
    print(D, len(taps), h.hexdigest()[:24])
