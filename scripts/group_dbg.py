import os, sys, threading
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import xritdemod_amd as xa
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.normpath(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), '../tests')))
import synth  # tests/synth.py: the NumPy specification of the synthetic burst (test infrastructure)
import oracle
n, D = 1200000, 5
dev = torch.device("cuda", 0)
for ph in (0.7, 2.3, 1.2, 3.0):
    x = synth.generate(synth.SynthParams(fs_in=6.25e6, phase0=ph), 2 * n)
    want = oracle.Demod(oracle.config("lrit", 6.25e6, D)).process(x)
    one = xa.Demodulator(xa.Demodulator.config("lrit", 6.25e6, D)).process(x)
    ser = xa.Demodulator(xa.Demodulator.config("lrit", 6.25e6, D, clock_serial=1)).process(x)
    fabric = xa.LocalFabric(2)
    xt = torch.from_numpy(x.view(np.float32).reshape(-1, 2)).to(dev)
    res = [None, None]
    def rank_main(r):
        g = xa.Group(xa.Demodulator.config("lrit", 6.25e6, D), r, fabric=fabric)
        cap = n // D + 1024
        soft = torch.empty(cap, dtype=torch.float32, device=dev)
        sl = xt[r * n:(r + 1) * n].contiguous()
        k, off, pol = g.process_slice_device(sl.data_ptr(), n, soft.data_ptr(), cap)
        res[r] = (soft[:k].cpu().numpy(), off, pol)
    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(2)]
    [t.start() for t in th]; [t.join() for t in th]
    (s0, o0, p0), (s1, o1, p1) = res
    got = np.concatenate([s0, s1])
    sg = 1.0 if np.dot(got[:50000], want[:50000]) > 0 else -1.0
    e = sg * got - want
    print(f"phase0 {ph}: pol1 {p1} global sign {sg}; rank0 rms {np.sqrt(np.mean(e[:len(s0)]**2)):.2e} rank1 rms {np.sqrt(np.mean(e[len(s0):]**2)):.2e}; single chain vs oracle {np.sqrt(np.mean((sg*one-want)**2)):.2e}; rank1 vs the single chain {np.sqrt(np.mean((got[len(s0):]-one[len(s0):])**2)):.2e}, vs the serial trajectory {np.sqrt(np.mean((got[len(s0):]-ser[len(s0):])**2)):.2e}; serial vs oracle on rank 1's part {np.sqrt(np.mean((sg*ser[len(s0):]-want[len(s0):])**2)):.2e}")
    e1 = e[len(s0):]
    for a in range(0, len(e1), 40000):
        print(f"   rank1 symbols {a:7d}..: rms {np.sqrt(np.mean(e1[a:a+40000]**2)):.2e}")
