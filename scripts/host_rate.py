"""PCIe-inclusive rate of the host entry point xrit_demod_process: host cf32 buffer in,
host soft symbols out, one call per chunk."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import xritdemod_amd as xa
from xritdemod_amd import synth
fs, D = 6.25e6, 5
for log2 in (19, 22, 24, 26):
    n = 1 << log2
    x = synth.generate(synth.SynthParams(fs_in=fs), min(n, 1 << 22))
    x = np.tile(x, max(1, n // len(x)))[:n].copy()
    dem = xa.Demodulator(xa.Demodulator.config("lrit", fs, D))
    dem.process(x)
    reps = max(2, (1 << 27) // n)
    t0 = time.perf_counter()
    for _ in range(reps):
        dem.process(x)
    dt = (time.perf_counter() - t0) / reps
    print(f"chunk 2^{log2} samples ({n*8/2**20:.0f} MiB): {dt*1e3:.2f} ms/call, {n/dt/1e6:.0f} Msamples/s host to host")
