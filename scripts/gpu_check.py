"""Quick stage-by-stage comparison of the HIP path against the CPU oracle (dev aid)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
import xritdemod_amd as xa
from xritdemod_amd import synth


def rms(a):
    return float(np.sqrt(np.mean(np.abs(a) ** 2))) if len(a) else 0.0


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
    print(xa.version(), "devices", xa.device_count())
    rng = np.random.default_rng(0)
    # ---- FIR, random input
    for (D, taps) in ((5, oracle.lowpass_taps(1, 6.25e6, 625e3, 100e3)), (1, oracle.rrc_taps(1, 1.25e6, 293883, 0.5, 63)),
                      (32, oracle.lowpass_taps(1, 40e6, 625e3, 100e3)), (2, oracle.rrc_taps(1, 1.25e6, 293883, 0.5, 63))):
        x = (rng.standard_normal(50000 * D) + 1j * rng.standard_normal(50000 * D)).astype(np.complex64)
        o = oracle.FirFilter(D, taps); g = xa.FirFilter(D, taps)
        e = []
        for c in range(2):   # two calls: history
            seg = x[c * 25000 * D:(c + 1) * 25000 * D]
            e.append(np.abs(o.Work(seg, 25000) - g.Work(seg, 25000)).max())
        print(f"FIR D={D} T={len(taps)}: max err {max(e):.3e}")
    # ---- chain, synthetic LRIT d=1
    p = synth.SynthParams()
    x = synth.generate(p, N)
    od = oracle.Demod(oracle.config("lrit", 1.25e6, 1))
    so = od.process(x)
    gd = xa.Demodulator(xa.Demodulator.config("lrit", 1.25e6, 1))
    gd.keep_stages(True)
    t0 = time.time(); sg = gd.process(x); t1 = time.time()
    st = gd.stats()
    print(f"chain d=1 N={N}: oracle {len(so)} syms, gpu {len(sg)} syms, {t1-t0:.3f}s  costas passes {st.costas_passes} unconv {st.costas_unconverged} "
          f"maxres {st.costas_max_residual:.2e} | clock passes {st.clock_passes} unconv {st.clock_unconverged} maxres {st.clock_max_residual:.2e}")
    for name in ("agc", "rrc", "costas"):
        a, b = od.stage(name), gd.stage(name)
        n = min(len(a), len(b))
        print(f"  stage {name}: n {len(a)}/{len(b)} rms err {rms(a[:n]-b[:n]):.3e} max {np.abs(a[:n]-b[:n]).max():.3e}")
    a, b = od.stage("clock"), gd.stage("clock")
    n = min(len(a), len(b))
    if n:
        e = np.abs(a[:n] - b[:n])
        print(f"  stage clock: n {len(a)}/{len(b)} rms err {rms(e):.3e} max {e.max():.3e} frac>1e-3 {(e>1e-3).mean():.4f}")
    n = min(len(so), len(sg))
    if n:
        e = np.abs(so[:n] - sg[:n])
        big = np.abs(so[:n]) > 1e-3
        print(f"  soft: rms {rms(e):.3e} max {e.max():.3e} sign mismatches {(np.sign(so[:n])[big] != np.sign(sg[:n])[big]).sum()}")
    # second call (streaming state)
    x2 = synth.generate(p, 100000, start=N)
    so2 = od.process(x2); sg2 = gd.process(x2)
    n = min(len(so2), len(sg2))
    print(f"  2nd call: {len(so2)}/{len(sg2)} syms rms {rms(so2[:n]-sg2[:n]):.3e}", "stats", gd.stats().costas_passes, gd.stats().clock_passes)
    # ---- chain d=5
    p5 = synth.SynthParams(fs_in=6.25e6)
    x = synth.generate(p5, N)
    od = oracle.Demod(oracle.config("lrit", 6.25e6, 5)); so = od.process(x)
    gd = xa.Demodulator(xa.Demodulator.config("lrit", 6.25e6, 5)); gd.keep_stages(True); sg = gd.process(x)
    n = min(len(so), len(sg))
    print(f"chain d=5: {len(so)}/{len(sg)} syms")
    for name in ("decimator", "agc", "rrc", "costas"):
        a, b = od.stage(name), gd.stage(name)
        m = min(len(a), len(b))
        print(f"  stage {name}: n {len(a)}/{len(b)} rms err {rms(a[:m]-b[:m]):.3e} max {np.abs(a[:m]-b[:m]).max():.3e}")
    if n:
        e = np.abs(so[:n] - sg[:n]); big = np.abs(so[:n]) > 1e-3
        print(f"  soft: rms {rms(e):.3e} max {e.max():.3e} sign mismatches {(np.sign(so[:n])[big] != np.sign(sg[:n])[big]).sum()}")
    print("profile:")
    gd.profile(True); gd.process(synth.generate(p5, N, start=N))
    for r in gd.profile_read():
        print("   ", r)


if __name__ == "__main__":
    main()
