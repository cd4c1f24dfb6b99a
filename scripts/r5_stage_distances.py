import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xritdemod_amd as xa
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.normpath(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), '../tests')))
import synth  # tests/synth.py: the NumPy specification of the synthetic burst (test infrastructure)
import oracle
def rms(a): return float(np.sqrt(np.mean(np.abs(a)**2)))
for name,(mode,fs,D,kw,n) in {"C2":("lrit",6.25e6,5,dict(fs_in=6.25e6),8000000),"C3":("hrit",2.5e6,1,dict(fs_in=2.5e6,symbol_rate=927000.0,alpha=0.3),3000000)}.items():
    x = synth.generate(synth.SynthParams(**kw), 2*n)
    ref = oracle.Demod(oracle.config(mode, fs, D)); dem = xa.Demodulator(xa.Demodulator.config(mode, fs, D)); dem.keep_stages(True)
    for part in range(2):
        w = ref.process(x[part*n:(part+1)*n]); g = dem.process(x[part*n:(part+1)*n])
        out = {}
        for st in (["decimator"] if D > 1 else []) + ["agc", "rrc", "costas"]:
            a, b = ref.stage(st), dem.stage(st)
            out[st] = "%.2e" % rms(a - b)
        print(name, "call", part, out, "soft rms %.3e" % rms(w - g), "costas passes", dem.stats().costas_passes, flush=True)
