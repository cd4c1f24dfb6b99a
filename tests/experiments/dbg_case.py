import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, oracle, xritdemod_amd as xa
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.normpath(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), '..')))
import synth  # tests/synth.py: the NumPy specification of the synthetic burst (test infrastructure)
# reproduce one case of fuzz_chain.py (FUZZ_WIDE=1): python dbg_case.py <seed> <case>, stage by stage
SEED, CASE = int(sys.argv[1]), int(sys.argv[2])
snr_lo, snr_hi = (float(v) for v in os.environ.get("FUZZ_SNR", "8,20").split(","))
rng = np.random.default_rng(SEED)
for c in range(CASE + 1):
    mode = "lrit" if rng.random() < 0.7 else "hrit"
    D = int(rng.choice([1, 1, 2, 3, 5, 5, 8, 16, 32]))
    base = 1.25e6 if mode == "lrit" else 2.5e6
    fs = base * D
    n = int(rng.integers(1, int(os.environ.get("FUZZ_MAX", "400000")))) * D + int(rng.integers(0, D))
    typ = int(rng.choice([0, 0, 0, 1, 2]))
    seed = int(rng.integers(1, 1 << 30))
    extra = dict(esn0_db=float(rng.uniform(snr_lo, snr_hi)), carrier_hz=float(rng.uniform(-600, 600)), clock_ppm=float(rng.uniform(-100, 100)), timing_offset=float(rng.uniform(0, 1)), phase0=float(rng.uniform(-3.1, 3.1)))
    ncut = int(rng.integers(0, 4)); cutv = [int(v) for v in rng.integers(0, n + 1, ncut)]
    keep = rng.random() < 0.3
print(c, mode, D, n, typ, seed, extra, cutv, keep)
sym, alpha = (293883.0, 0.5) if mode == "lrit" else (927000.0, 0.3)
x = synth.generate(synth.SynthParams(fs_in=fs, symbol_rate=sym, alpha=alpha, amplitude=0.1, seed=seed, **extra), n)
cuts = sorted(set([0, n] + cutv))
for mp in (0,):
  for keepv in (True,):
    od, gd = oracle.Demod(oracle.config(mode, fs, D)), xa.Demodulator(xa.Demodulator.config(mode, fs, D, max_passes=mp))
    gd.keep_stages(keepv)
    print("max_passes", mp, "keep", keepv)
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        w, g = od.process(x[lo:hi], 0), gd.process(x[lo:hi], 0)
        st = gd.stats()
        big = np.abs(w) > 1e-3
        msg = "" if len(w) != len(g) or not len(w) else "rms %.2e sign %d" % (np.sqrt(np.mean((w - g) ** 2)), np.sum(np.sign(w[big]) != np.sign(g[big])))
        print("  call", lo, hi, "symbols", len(w), len(g), msg, "passes", st.costas_passes, st.clock_passes, "unconv", st.costas_unconverged, st.clock_unconverged, "resid %.2e %.2e" % (st.costas_max_residual, st.clock_max_residual))
        if keepv and len(w) == len(g) and len(w):
            for name in gd.STAGES:
                try:
                    a = gd.stage(name); b = od.stage(name)
                    m = min(len(a), len(b))
                    print("     stage", name, len(a), len(b), "max diff %.2e" % (np.abs(a[:m] - b[:m]).max() if m else 0))
                except Exception as e:
                    print("     stage", name, "err", e)
