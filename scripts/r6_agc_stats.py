"""Round 6: the exact AGC's block statistics (Picard rounds, lattice segments, fallbacks) on the chain's own signals:
the oracle's decimator output (C2) / the raw circuit-rate stream (C1, C3).  Runs on the GPU box."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle, synth
import xritdemod_amd as xa
for name, mode, fs, D, kw in (("C2", "lrit", 6.25e6, 5, {}), ("C1", "lrit", 1.25e6, 1, {}), ("C3", "hrit", 2.5e6, 1, dict(symbol_rate=927000.0, alpha=0.3))):
    x = synth.generate(synth.SynthParams(fs_in=fs, **kw), 3_000_000 * D)
    od = oracle.Demod(oracle.config(mode, fs, D)); od.process(x)
    inp = od.stage("decimator") if D > 1 else x
    ag = xa.AGC(0.01, 0.5, 1.0, 4000.0, exact=True)
    ag.Work(inp[:1_000_000])
    y = ag.Work(inp[1_000_000:])
    st = ag.exact_stats()
    print(name, "gain %.4f" % ag.gain, st, "rounds/block %.2f segs/scan %.2f fallback %.1f %%" % (st["picard_rounds"] / st["blocks"], st["lattice_segments"] / st["picard_rounds"], 100.0 * st["lattice_fallbacks"] / st["picard_rounds"]))
