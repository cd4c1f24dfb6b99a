"""Dev aid: the C2 decimator alone (stage object, 60 Mi samples), for rocprofv3 --kernel-trace timing of kernel variants."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import xritdemod_amd as xa
taps = np.asarray(xa.Filters.lowPass(1, 6.25e6, 625e3, 100e3), np.float32)
n_out = 12 * 1024 * 1024
x = (np.random.default_rng(0).standard_normal(2 * n_out * 5).astype(np.float32)).view(np.complex64)
f = xa.FirFilter(5, taps)
for _ in range(3):
    y = f.Work(x, n_out)
print(len(taps), float(abs(y).mean()))
