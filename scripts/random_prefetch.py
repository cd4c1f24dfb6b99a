"""Streamed (xrit_demod_prefetch_device) against plain calls on one stream cut in calls of random sizes: the symbols must be the same words.
python scripts/random_prefetch.py [calls] [seed]      (FRONT_EXACT=1: cfg.front_exact = 1)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import xritdemod_amd as xa
from xritdemod_amd import _capi
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
fs, D = 6.25e6, 5
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream(dev)
sizes = [int(rng.choice([rng.integers(1000, 60000), rng.integers(60000, 3000000), rng.integers(3000000, 40000000), 1 << 27])) for _ in range(calls)]
total = sum(sizes)
sp = _capi.synth_params(fs_in=fs)
x = torch.empty((total, 2), dtype=torch.float32, device=dev)
_capi.synth_generate_device(sp, 0, total, x.data_ptr(), device=0, stream=st.cuda_stream)
torch.cuda.synchronize()
offs = np.concatenate([[0], np.cumsum(sizes)])
cap = max(sizes) // (D * 4) + 4096
soft = torch.empty((cap,), dtype=torch.float32, device=dev)
def run(prefetch):
    dem = xa.Demodulator(xa.Demodulator.config("lrit", fs, D, front_exact=int(os.environ.get("FRONT_EXACT", "0"))))
    out, stats = [], []
    reg = 0          # inputs registered so far: as many ahead as the library takes (round 5: up to two behind the call in progress)
    def feed(c):
        nonlocal reg
        while prefetch and reg < calls and reg <= c + 2:
            try:
                dem.prefetch_device(x[offs[reg]:].data_ptr(), sizes[reg], stream=st.cuda_stream)
            except xa.XritError:
                break
            reg += 1
    for c in range(calls):
        feed(c)
        if prefetch and reg <= c:
            raise SystemExit("the call's own input could not be registered")
        k = dem.process_device(x[offs[c]:].data_ptr(), sizes[c], soft.data_ptr(), cap, stream=st.cuda_stream)
        out.append(soft[:k].cpu().numpy().copy())
        s = dem.stats(); stats.append((s.costas_passes, s.clock_passes, s.clock_relay_passes))
    return out, stats
a, sa = run(False)
b, sb = run(True)
diff = sum(int((u.view(np.uint32) != v.view(np.uint32)).sum()) if len(u) == len(v) else -10**9 for u, v in zip(a, b))
print("calls", calls, "samples", total, "symbols", sum(len(u) for u in a), "words differing", diff, "stats equal", sa == sb)
if sa != sb:
    for c, (p, q) in enumerate(zip(sa, sb)):
        if p != q: print("  call", c, "n", sizes[c], p, q)
