import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import xritdemod_amd as xa
from xritdemod_amd import synth
for seed, esn0, carrier, ppm, toff, ph, n in [(207702987, 3.45, 381.61, 9.98, 0.51, 1.61, 274735), (151577245, 2.66, -139.61, -7.34, 0.96, 2.74, 221449)]:
    p = synth.SynthParams(fs_in=12.5e6, symbol_rate=927000.0, alpha=0.3, amplitude=0.1, seed=seed, esn0_db=esn0, carrier_hz=carrier, clock_ppm=ppm, timing_offset=toff, phase0=ph)
    x = synth.generate(p, n)
    outs = {}
    for tag, kw in (("default", {}), ("serial", dict(clock_serial=1)), ("exact", dict(clock_exact=1)), ("tiled", dict(clock_exact=-1))):
        dem = xa.Demodulator(xa.Demodulator.config("hrit", 12.5e6, 5, **kw))
        outs[tag] = dem.process(x); st = dem.stats()
        print(tag, len(outs[tag]), "clock passes", st.clock_passes, "max_res", st.clock_max_residual, "open_large", st.clock_open_large, "unconv", st.clock_unconverged, "relay", st.clock_relay_passes, st.clock_relay_closed, flush=True)
    for tag in ("default", "exact", "tiled"):
        d = outs[tag] - outs["serial"]
        print("  ", tag, "vs serial rms", float(np.sqrt(np.mean(d**2))), "differing", int(np.sum(outs[tag] != outs["serial"])))
