import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, oracle
import xritdemod_amd as xa
from xritdemod_amd import synth

def prof(e, parts=16):
    q = len(e) // parts
    return " ".join(f"{np.sqrt(np.mean(e[i*q:(i+1)*q]**2)):.1e}" for i in range(parts))

for (fs, D, N) in ((1.25e6, 1, 400000), (6.25e6, 5, 400000), (6.25e6, 5, 2000000)):
    p = synth.SynthParams(fs_in=fs); x = synth.generate(p, N)
    od = oracle.Demod(oracle.config("lrit", fs, D)); so = od.process(x)
    for mp in (8, 24, 64):
        gd = xa.Demodulator(xa.Demodulator.config("lrit", fs, D, max_passes=mp)); sg = gd.process(x)
        st = gd.stats()
        n = min(len(so), len(sg)); e = np.abs(so[:n] - sg[:n])
        print(f"fs={fs} D={D} N={N} max_passes={mp}: syms {len(so)}/{len(sg)} costas {st.costas_passes} clock {st.clock_passes} "
              f"unconv {st.clock_unconverged} maxres {st.clock_max_residual:.2e} rms {np.sqrt(np.mean(e**2)):.3e} max {e.max():.2e}")
        print("    profile:", prof(e))
