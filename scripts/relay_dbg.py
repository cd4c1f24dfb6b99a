import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import xritdemod_amd as xa
from xritdemod_amd import synth
lg, exact, window = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
x = synth.generate(synth.SynthParams(fs_in=6.25e6), 1 << lg)
dem = xa.Demodulator(xa.Demodulator.config("lrit", 6.25e6, 5, clock_exact=exact, clock_exact_window=window))
for c in range(2):
    t0 = time.perf_counter(); y = dem.process(x); dt = time.perf_counter() - t0
    st = dem.stats()
    print(f"2^{lg} exact={exact} window={window} call {c}: {len(y)} symbols {dt*1e3:.2f} ms relay passes {st.clock_relay_passes} closed {st.clock_relay_closed} segments {st.clock_relay_segments}", flush=True)
