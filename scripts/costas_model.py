"""How far a model of the Costas loop over runs of R samples stays from the loop itself (CPU, oracle only).

The oracle's chain gives the matched-filter output z and the de-rotated stream y; the loop's phase at every sample is
arg(z conj(y)).  Compared at the chain boundaries (every 256 samples):
  * the block-average guess of costas_guess_kernel (1/2 arg sum z^2 per chain, interpolated),
  * the model recurrence of costas_model_pass_kernel, run serially over the stream from a chain in lock:
        e = 1/2 Im(sum_run z^2 e^{-2j phi_mid}),  f += beta e,  phi += R f + (alpha + beta (R + 1) / 2) e
Usage: python scripts/costas_model.py [n_input_samples]      (C2's rates: 6.25 Msps, decimation 5)
"""
import sys
import os
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import oracle                                    # noqa: E402
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.normpath(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), '../tests')))
import synth  # tests/synth.py: the NumPy specification of the synthetic burst (test infrastructure)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
fs, D, L = 6.25e6, 5, 256
x = synth.generate(synth.SynthParams(fs_in=fs), n)
dem = oracle.Demod(oracle.config("lrit", fs, D))
dem.process(x)
z = dem.stage("rrc").astype(np.complex128)
y = dem.stage("costas").astype(np.complex128)
phi = np.angle(z * np.conj(y))
K = len(z) // L
lb = 0.0037
damp = np.sqrt(2) / 2
den = 1 + 2 * damp * lb + lb * lb
alpha, beta = 4 * damp * lb / den, 4 * lb * lb / den
z2 = z * z


def wrap(v, p=np.pi):
    return (v + p / 2) % p - p / 2


c = z2[:K * L].reshape(K, L).sum(1)
th2 = np.unwrap(np.angle(c))
ga, fa = np.zeros(K), np.zeros(K)
for k in range(1, K):
    ga[k] = 0.25 * (th2[k - 1] + th2[k])
    a, b = max(0, k - 2), min(K - 1, k + 1)
    fa[k] = 0.5 * (th2[b] - th2[a]) / ((b - a) * L)
true = phi[np.arange(K) * L]
e = wrap(ga - true)[160:]
print("block-average guess: rms %.3e max %.3e rad" % (np.sqrt(np.mean(e ** 2)), np.abs(e).max()))
for R in (8, 16, 32, 64):
    nb = len(z) // R
    s = z2[:nb * R].reshape(nb, R).sum(1)
    bpc = L // R
    out = np.zeros(K)
    p, f = ga[60], fa[60]
    for k in range(60, K):
        out[k] = p
        for b in range(k * bpc, (k + 1) * bpc):
            pm = p + f * (R - 1) * 0.5
            er = 0.5 * np.imag(s[b] * np.exp(-2j * pm))
            p = p + R * f + (alpha + beta * (R + 1) * 0.5) * er
            f = f + beta * er
    e = wrap(out - true)[160:]
    print("model, runs of %2d:    rms %.3e max %.3e rad" % (R, np.sqrt(np.mean(e ** 2)), np.abs(e).max()))
