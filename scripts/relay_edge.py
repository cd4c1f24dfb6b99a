"""Edge cases of the exact closure against the serial wave: large samples-per-symbol (the walker that reads global memory),
tiny and empty calls, many small calls, kept stages (complex symbols)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import xritdemod_amd as xa
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.normpath(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), '../tests')))
import synth  # tests/synth.py: the NumPy specification of the synthetic burst (test infrastructure)

def pair(mode, fs, D, **kw):
    return (xa.Demodulator(xa.Demodulator.config(mode, fs, D, clock_serial=1, **kw)),
            xa.Demodulator(xa.Demodulator.config(mode, fs, D, clock_exact=1, **kw)))

def same(a, b):
    return len(a) == len(b) and np.array_equal(a.view(np.uint32), b.view(np.uint32))

# 1. large sps: LRIT at 6.25 Msps without decimation (sps 21.3) and 20 Msps (sps 68)
for fs in (6.25e6, 20e6):
    x = synth.generate(synth.SynthParams(fs_in=fs), 3000000)
    s, e = pair("lrit", fs, 1)
    ok = True
    for a, b in ((0, 1000001), (1000001, 3000000)):
        ys, ye = s.process(x[a:b]), e.process(x[a:b])
        ok &= same(ys, ye)
        st = e.stats()
    print(f"sps {fs/293883:.1f}: equal {ok}, symbols {len(ye)}, relay passes {st.clock_relay_passes} closed {st.clock_relay_closed}", flush=True)

# 2. tiny / empty / many small calls
x = synth.generate(synth.SynthParams(fs_in=6.25e6), 1200000)
s, e = pair("lrit", 6.25e6, 5)
cuts = [0, 0, 7, 40, 45, 300, 5000, 5005, 70000, 70000, 400000, 400020, 1200000]
ok, tot = True, 0
for a, b in zip(cuts[:-1], cuts[1:]):
    ys, ye = s.process(x[a:b]), e.process(x[a:b])
    ok &= same(ys, ye); tot += len(ye)
print(f"ragged calls: equal {ok}, symbols {tot}", flush=True)

# 3. kept stages: complex symbols
s, e = pair("lrit", 6.25e6, 5)
s.keep_stages(True); e.keep_stages(True)
ys, ye = s.process(x), e.process(x)
cs, ce = s.stage("clock"), e.stage("clock")
print(f"kept stages: soft equal {same(ys, ye)}, complex equal {np.array_equal(cs.view(np.uint32), ce.view(np.uint32))}, re == soft {np.array_equal(ce.real, ye)}", flush=True)

# 4. HRIT with decimation 5, s16 ingest, three calls
p = synth.SynthParams(fs_in=12.5e6, symbol_rate=927000.0, alpha=0.3)
x = synth.generate(p, 2000000)
xi = np.clip(np.round(x.view(np.float32) * 32768), -32768, 32767).astype(np.int16)
s, e = pair("hrit", 12.5e6, 5)
ok = True
for a, b in ((0, 600000), (600000, 600010), (600010, 2000000)):
    ys, ye = s.process(xi[2 * a:2 * b], 1), e.process(xi[2 * a:2 * b], 1)
    ok &= same(ys, ye)
print(f"hrit d=5 s16: equal {ok}", flush=True)
