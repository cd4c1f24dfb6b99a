"""Generates the golden fixtures under tests/golden/.

The reference ships no test vectors (/root/reference/Makefile:91-92) and its DSP
library is absent, so these fixtures pin THIS repo's oracle against regressions:
a seeded synthetic burst and the oracle's output at every stage, plus the tap
sets.  Regenerate with:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle  # noqa: E402
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.normpath(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), '..')))
import synth  # tests/synth.py: the NumPy specification of the synthetic burst (test infrastructure)


def main():
    out = {}
    # C2-style: LRIT, decimation 5
    n = 40000
    p = synth.SynthParams(fs_in=6.25e6)
    x = synth.generate(p, n)
    d = oracle.Demod(oracle.config("lrit", 6.25e6, 5))
    soft = d.process(x)
    out["lrit_d5_in"] = x
    for st in oracle.Demod.STAGES:
        out["lrit_d5_" + st] = d.stage(st)
    out["lrit_d5_soft"] = soft
    out["lrit_d5_i8"] = oracle.quantize_i8(soft)
    out["lrit_d5_dec_taps"] = d.decimator_taps()
    out["lrit_rrc_taps"] = d.rrc_taps()
    # HRIT, decimation 1
    ph = synth.SynthParams(fs_in=2.5e6, symbol_rate=927000.0, alpha=0.3)
    xh = synth.generate(ph, 16000)
    dh = oracle.Demod(oracle.config("hrit", 2.5e6, 1))
    out["hrit_d1_in"] = xh
    out["hrit_d1_soft"] = dh.process(xh)
    out["hrit_rrc_taps"] = dh.rrc_taps()
    out["mmse_table"] = oracle.mmse_table()
    np.savez_compressed(os.path.join(HERE, "oracle_stages.npz"), **out)
    print("wrote", os.path.join(HERE, "oracle_stages.npz"), {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
