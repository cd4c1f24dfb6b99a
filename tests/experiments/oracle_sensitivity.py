"""Dev aid (CPU only): how far the ORACLE moves when its input moves by a few 1e-6 (relative), on the fuzz cases that
fail the parity bar at low Es/N0.  python tests/experiments/oracle_sensitivity.py <seed> <case,case,...>
(same generator as fuzz_chain.py with FUZZ_WIDE=1 and the given FUZZ_SNR / FUZZ_MAX)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, oracle
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.normpath(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), '..')))
import synth  # tests/synth.py: the NumPy specification of the synthetic burst (test infrastructure)
oracle.build()
seed0 = int(sys.argv[1]); want = set(int(v) for v in sys.argv[2].split(","))
rng = np.random.default_rng(seed0)
snr_lo, snr_hi = (float(v) for v in os.environ.get("FUZZ_SNR", "8,20").split(","))
for c in range(max(want) + 1):
    mode = "lrit" if rng.random() < 0.7 else "hrit"
    D = int(rng.choice([1, 1, 2, 3, 5, 5, 8, 16, 32]))
    base = 1.25e6 if mode == "lrit" else 2.5e6
    fs = base * D
    n = int(rng.integers(1, int(os.environ.get("FUZZ_MAX", "400000")))) * D + int(rng.integers(0, D))
    typ = int(rng.choice([0, 0, 0, 1, 2]))
    sym, alpha = (293883.0, 0.5) if mode == "lrit" else (927000.0, 0.3)
    amp = 0.1 if typ == 0 else 0.3
    seed = int(rng.integers(1, 1 << 30))
    extra = dict(esn0_db=float(rng.uniform(snr_lo, snr_hi)), carrier_hz=float(rng.uniform(-600, 600)),
                 clock_ppm=float(rng.uniform(-100, 100)), timing_offset=float(rng.uniform(0, 1)), phase0=float(rng.uniform(-3.1, 3.1)))
    ncut = int(rng.integers(0, 4)); cutv = [int(v) for v in rng.integers(0, n + 1, ncut)]
    keep = rng.random() < 0.3
    if os.environ.get("FUZZ_CFG"):
        rng.choice([0, 64, 128, 256, 320, 512]); rng.choice([0, 32, 48, 64, 100, 256]); rng.choice([0, 1, 2, 3])
    if c not in want:
        continue
    x = synth.generate(synth.SynthParams(fs_in=fs, symbol_rate=sym, alpha=alpha, amplitude=amp, seed=seed, **extra), n)
    if typ == 1:      # what the s16 ingest makes of it, as floats
        x = (np.clip(np.round(x.view(np.float32) * 32768), -32768, 32767) / 32768.0).astype(np.float32).view(np.complex64)
    a = oracle.Demod(oracle.config(mode, fs, D)).process(x)
    res = []
    for trial in range(3):
        r2 = np.random.default_rng(trial)
        # every sample moved by a relative 4e-6 (random): about what two float32 implementations of the front end
        # differ by at the clock recovery's input (de-rotated stream: ~2e-6 absolute)
        xp = (x.view(np.float32) * (1.0 + 4e-6 * r2.standard_normal(2 * len(x))).astype(np.float32)).astype(np.float32)
        b = oracle.Demod(oracle.config(mode, fs, D)).process(xp.view(np.complex64))
        if len(a) != len(b):
            res.append("count differs")
            continue
        big = np.abs(a) > 1e-3
        res.append("rms %.2e sign %d" % (np.sqrt(np.mean((a - b) ** 2)), int(np.sum(np.sign(a[big]) != np.sign(b[big])))))
    print("case", c, mode, "D", D, "n", n, "Es/N0 %.1f" % extra["esn0_db"], "symbols", len(a), "| oracle vs oracle(input x (1 + 4e-6 N(0,1))):", res, flush=True)
