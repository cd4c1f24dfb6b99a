"""Exact closure (cfg.clock_exact) against the serial device trajectory (cfg.clock_serial): bitwise comparison,
relay pass counts and time per call, for a few burst sizes and windows.
    python scripts/relay_check.py [log2 samples ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import xritdemod_amd as xa
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.normpath(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), '../tests')))
import synth  # tests/synth.py: the NumPy specification of the synthetic burst (test infrastructure)


def run(cfg_kw, x, calls=1, mode="lrit", fs=6.25e6, D=5):
    dem = xa.Demodulator(xa.Demodulator.config(mode, fs, D, **cfg_kw))
    outs, times, stats = [], [], []
    per = len(x) // calls
    for c in range(calls):
        t0 = time.perf_counter()
        outs.append(dem.process(x[c * per:(c + 1) * per]))
        times.append(time.perf_counter() - t0)
        stats.append(dem.stats())
    return outs, times, stats


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [19, 21, 23]
    for lg in sizes:
        n = 1 << lg
        p = synth.SynthParams(fs_in=6.25e6)
        x = synth.generate(p, n)
        ser, tser, _ = run(dict(clock_serial=1), x, calls=2)
        fast, tf, _ = run(dict(clock_exact=-2), x, calls=2)
        for w in (0, 8, 64):
            ex, tex, st = run(dict(clock_exact=1, clock_exact_window=w), x, calls=2)
            for c in range(2):
                same = len(ex[c]) == len(ser[c]) and np.array_equal(ex[c].view(np.uint32), ser[c].view(np.uint32))
                nd = int(np.sum(ex[c].view(np.uint32) != ser[c].view(np.uint32))) if len(ex[c]) == len(ser[c]) else -1
                r_fast = float(np.sqrt(np.mean((fast[c] - ser[c]) ** 2))) if len(fast[c]) == len(ser[c]) else -1
                print(f"2^{lg} call {c} window {w}: symbols {len(ex[c])} (serial {len(ser[c])}), bitwise equal {same} "
                      f"(differing {nd}), relay passes {st[c].clock_relay_passes} closed {st[c].clock_relay_closed} "
                      f"segments {st[c].clock_relay_segments}; exact {tex[c] * 1e3:.2f} ms, fast {tf[c] * 1e3:.2f} ms "
                      f"(rms vs serial {r_fast:.2e}), serial {tser[c] * 1e3:.1f} ms", flush=True)


if __name__ == "__main__":
    main()
