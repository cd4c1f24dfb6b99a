"""cfg.front_exact = 0 / 1 against the oracle, stage by stage, on two consecutive calls of C2 and C3 (the second one is a
tracking call); clock recovery to closure (the serial trajectory) and as shipped.  python scripts/r5_front_exact_check.py [log2 of samples]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import xritdemod_amd as xa
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.normpath(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), '../tests')))
import synth  # tests/synth.py: the NumPy specification of the synthetic burst (test infrastructure)
import oracle

def rms(a): return float(np.sqrt(np.mean(np.abs(a) ** 2)))
log2 = int(sys.argv[1]) if len(sys.argv) > 1 else 24
for name, (mode, fs, D, kw) in {"C2": ("lrit", 6.25e6, 5, dict(fs_in=6.25e6)), "C3": ("hrit", 2.5e6, 1, dict(fs_in=2.5e6, symbol_rate=927000.0, alpha=0.3))}.items():
    n = (1 << log2) - (1 << log2) % D
    x = synth.generate(synth.SynthParams(**kw), 2 * n)
    ref = oracle.Demod(oracle.config(mode, fs, D))
    want = []
    for part in range(2):
        w = ref.process(x[part * n:(part + 1) * n])
        want.append((w, {st: ref.stage(st).copy() for st in (["decimator"] if D > 1 else []) + ["agc", "rrc", "costas"]}))
    for fe in (0, 1):
        for exact in (1, 0):
            dem = xa.Demodulator(xa.Demodulator.config(mode, fs, D, front_exact=fe, clock_exact=exact))
            dem.keep_stages(exact == 1)        # (the stage copies only with the clock to closure; the other run is the chain as it streams)
            for part in range(2):
                g = dem.process(x[part * n:(part + 1) * n])
                w, stages = want[part]
                out = {st: "%.2e" % rms(stages[st] - dem.stage(st)) for st in stages} if exact == 1 else {}
                stt = dem.stats()
                print(name, "front_exact", fe, "clock_exact", exact, "call", part, out, "soft rms %.3e" % (rms(w - g) if len(w) == len(g) else float("nan")),
                      "costas passes", stt.costas_passes, "agc serial", stt.agc_serial_fallback, flush=True)
