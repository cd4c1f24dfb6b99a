"""CPU experiment (round 5, VERDICT item 3d, first stage): how far does the arithmetic of the two FIRs ALONE move the chain?
The oracle is rebuilt with its FIR accumulating as a sequential fmaf chain (what an FMA machine or the GPU's FIR does) instead of
four interleaved partial sums without FMA; everything else is the same source.  Result (C2, 16 M samples; C3, 6 M samples):
the Costas output moves by 1.9e-7 / 1.6e-7 rms -- and the soft symbols by 5.0e-5 / 9.4e-5: ANY difference in front of a float32
M&M costs that much.  The device chain's Costas output is 1.1e-6 from the oracle's, of which the AGC's
composed maps account for 4.5e-7 and the Costas hand-off for ~1e-6 (scripts/r5_stage_distances.py): FIRs summed in the oracle's
order would remove a fifth of the distance and none of the floor.
Usage: python tests/experiments/fir_order_sensitivity.py"""
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.normpath(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), '..')))
import synth  # tests/synth.py: the NumPy specification of the synthetic burst (test infrastructure)

OLD = '''        float sr[4] = {0, 0, 0, 0}, si[4] = {0, 0, 0, 0};
        int i = 0;
        for (; i + 4 <= T; i += 4) {
            for (int j = 0; j < 4; j++) {
                sr[j] += rt[i + j] * w[i + j].re;
                si[j] += rt[i + j] * w[i + j].im;
            }
        }
        for (int j = 0; i < T; i++, j++) {
            sr[j] += rt[i] * w[i].re;
            si[j] += rt[i] * w[i].im;
        }
        out[m].re = (sr[0] + sr[1]) + (sr[2] + sr[3]);
        out[m].im = (si[0] + si[1]) + (si[2] + si[3]);'''
NEW = '''        float ar = 0.f, ai = 0.f;
        for (int i = 0; i < T; i++) { ar = __builtin_fmaf(rt[i], w[i].re, ar); ai = __builtin_fmaf(rt[i], w[i].im, ai); }
        out[m].re = ar; out[m].im = ai;'''


def rms(a):
    return float(np.sqrt(np.mean(np.abs(a) ** 2)))


def main():
    tmp = tempfile.mkdtemp()
    src = open(os.path.join(ROOT, "oracle", "xrit_oracle.c")).read()
    assert OLD in src
    open(os.path.join(tmp, "xrit_oracle.c"), "w").write(src.replace(OLD, NEW))
    shutil.copy(os.path.join(ROOT, "oracle", "xrit_oracle.h"), tmp)
    var = os.path.join(tmp, "libvar.so")
    subprocess.check_call(["gcc", "-O3", "-mavx2", "-mfma", "-ffp-contract=off", "-fno-math-errno", "-fPIC", "-std=gnu11", "-shared",
                           "-o", var, os.path.join(tmp, "xrit_oracle.c"), "-lm"])
    oracle.build()
    base = oracle._LIB_PATH

    def run(lib, mode, fs, D, x):
        oracle._lib, oracle._LIB_PATH = None, lib
        d = oracle.Demod(oracle.config(mode, fs, D))
        s = d.process(x)
        return s, d.stage("costas"), d.stage("rrc")

    cases = {"C2": ("lrit", 6.25e6, 5, dict(fs_in=6.25e6), 16000000),
             "C3": ("hrit", 2.5e6, 1, dict(fs_in=2.5e6, symbol_rate=927000.0, alpha=0.3), 6000000)}
    for name, (mode, fs, D, kw, n) in cases.items():
        x = synth.generate(synth.SynthParams(**kw), n)
        a, b = run(base, mode, fs, D, x), run(var, mode, fs, D, x)
        h = len(a[0]) // 2
        print(f"{name}: {len(a[0])} symbols; FIRs as fmaf chains: rrc stage {rms(a[2] - b[2]):.2e}, costas stage {rms(a[1] - b[1]):.2e}, "
              f"soft symbols (second half) {rms(a[0][h:] - b[0][h:]):.2e} rms")
    oracle._lib, oracle._LIB_PATH = None, base


if __name__ == "__main__":
    main()
