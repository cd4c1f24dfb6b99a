"""Rate of the oracle's literal correlator loops on one host core (the CPU figure quoted beside
scripts/bench_sync.py in DESIGN.md)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import oracle

d = np.random.default_rng(7).integers(-128, 128, size=64 * 16384).astype(np.int8)
t0 = time.perf_counter()
oracle.sync_correlate(d)
print(f"{len(d) / (time.perf_counter() - t0) / 1e6:.1f} Msymbols/s")
