"""One stream through many calls of random sizes (a few samples .. a few million), the oracle beside it call by call: symbol counts,
hard decisions, soft rms.  Shakes the transitions between the clock recovery's plans (one exact walk / hand-off + relay / relay from the
timing guess), tiny and empty calls, and the carried state across them.  python scripts/random_calls.py [calls] [seed] [mode] [D]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import xritdemod_amd as xa
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.normpath(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), '../tests')))
import synth  # tests/synth.py: the NumPy specification of the synthetic burst (test infrastructure)
import oracle

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 150
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
mode = sys.argv[3] if len(sys.argv) > 3 else "lrit"
D = int(sys.argv[4]) if len(sys.argv) > 4 else 5
rng = np.random.default_rng(seed)
fs = (1.25e6 if mode == "lrit" else 2.5e6) * D
sym, alpha = (293883.0, 0.5) if mode == "lrit" else (927000.0, 0.3)
sizes = []
for c in range(calls):
    k = rng.integers(0, 10)
    if k == 0: sizes.append(int(rng.integers(0, 40)))                    # empty / a few samples
    elif k <= 3: sizes.append(int(rng.integers(40, 20000)))
    elif k <= 7: sizes.append(int(rng.integers(20000, 600000)))
    else: sizes.append(int(rng.integers(600000, 4000000)))
total = sum(sizes)
x = synth.generate(synth.SynthParams(fs_in=fs, symbol_rate=sym, alpha=alpha, seed=seed + 7, esn0_db=float(rng.uniform(6, 14))), total)
od, gd = oracle.Demod(oracle.config(mode, fs, D)), xa.Demodulator(xa.Demodulator.config(mode, fs, D))
pos, bad, worst, flips = 0, 0, 0.0, 0
W, G = [], []
for c, n in enumerate(sizes):
    seg = x[pos:pos + n]; pos += n
    w, g = od.process(seg), gd.process(seg)
    if len(w) != len(g):
        print("call", c, "n", n, "COUNT", len(w), len(g)); bad += 1; break
    W.append(w); G.append(g)
    if len(w):
        big = np.abs(w) > 1e-3
        f = int(np.sum(np.sign(w[big]) != np.sign(g[big])))
        flips += f
        r = float(np.sqrt(np.mean((w - g) ** 2)))
        worst = max(worst, r)
        if f or r > 1e-3:
            st = gd.stats()
            print("call", c, "n", n, "symbols", len(w), "rms %.2e" % r, "flips", f, "passes", st.costas_passes, st.clock_passes, st.clock_relay_passes, st.clock_relay_segments)
w, g = np.concatenate(W), np.concatenate(G)
print("calls", len(W), "samples", total, "symbols", len(w), "flips", flips, "rms over all %.2e" % float(np.sqrt(np.mean((w - g) ** 2))), "worst call %.2e" % worst, "count mismatches", bad)
