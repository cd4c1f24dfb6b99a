"""The matrix-pipe decimator (XRIT_MFMA_DEC=1, csrc/fir.hip) against the straight-line VALU one: outputs must be the same
words over several calls (history) and ragged lengths."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import xritdemod_amd as xa
lp = np.zeros(151, np.float32)
n = xa.lib().xrit_lowpass_taps(1.0, 6.25e6, 625e3, 100e3, lp.ctypes.data_as(__import__("ctypes").c_void_p), 151)
assert n == 151
rng = np.random.default_rng(12)
n_out = [100003, 5, 0, 4097, 70000, 768 * 40]
x = (rng.standard_normal(sum(n_out) * 5) + 1j * rng.standard_normal(sum(n_out) * 5)).astype(np.complex64)
x[1000:1200] = 0


def run():
    f, pos, out = xa.FirFilter(5, lp), 0, []
    for k in n_out:
        out.append(f.Work(x[pos:pos + k * 5], k)); pos += k * 5
    return np.concatenate(out)


a = run()
os.environ["XRIT_MFMA_DEC"] = "1"
b = run()
d = a.view(np.uint32) != b.view(np.uint32)
print("outputs", len(a), "differing words", int(d.sum()), "max abs diff", float(np.abs(a - b).max()))
