"""How long a float32 AGC walked from a start gain that is a few ulps off takes to become the true trajectory bit for bit (CPU,
the oracle's AGC): what a literally walked AGC on the device would need as warm-up in front of every chain.
    python tests/experiments/agc_merge_time.py      (LRIT at the circuit rate, 200 trials: median 2150 samples, 99 % 5667, max 6516)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.normpath(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), '..')))
import synth  # tests/synth.py: the NumPy specification of the synthetic burst (test infrastructure)

x = synth.generate(synth.SynthParams(fs_in=1.25e6), 600000)
cfg = oracle.config("lrit", 1.25e6, 1)
rng = np.random.default_rng(1)
m = []
for trial in range(200):
    k = 100000 + 2000 * trial
    t = oracle.AGC(cfg.agc_rate, cfg.agc_reference, cfg.agc_gain, cfg.agc_max_gain)
    t.Work(x[:k])
    ulps = int(rng.integers(-12, 13)) or 7
    gp = np.int32(np.float32(t.s.gain).view(np.int32) + ulps).view(np.float32)
    p = oracle.AGC(cfg.agc_rate, cfg.agc_reference, float(gp), cfg.agc_max_gain)
    yt, yp = t.Work(x[k:k + 20000]), p.Work(x[k:k + 20000])
    d = np.nonzero(yt.view(np.uint64) != yp.view(np.uint64))[0]
    m.append(int(d[-1]) + 1 if len(d) else 0)
m = np.array(m)
print("samples until the perturbed AGC output is the true one bit for bit: median %d, 90 %% %d, 99 %% %d, max %d (%d trials; 20000 = never)"
      % (np.median(m), np.percentile(m, 90), np.percentile(m, 99), m.max(), len(m)))
