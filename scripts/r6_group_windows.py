"""Round 6: WHERE rank 1 of a contiguous split differs from the uninterrupted CPU chain -- rms per window of 4096 symbols along
its part, the last differing word, per start phase (both Costas locks), for two halo lengths (XRIT_GROUP_WARM symbols).
   python scripts/r6_group_windows.py"""
import os, sys, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import oracle, synth
import xritdemod_amd as xa
def rms(a): return float(np.sqrt(np.mean(np.abs(a) ** 2))) if len(a) else 0.0
D = 5
dev = torch.device("cuda", 0)
for n in [int(v) for v in os.environ.get('GROUP_N', '3000000').split(',')]:
    for ph in [float(v) for v in os.environ.get('GROUP_PHASES', '0.7,1.5,2.3,3.1,3.9,4.7,5.4,6.1').split(',')]:
        x = synth.generate(synth.SynthParams(fs_in=6.25e6, phase0=ph, seed=4242), 2 * n)
        od = oracle.Demod(oracle.config("lrit", 6.25e6, D))
        want = od.process(x)
        fabric = xa.LocalFabric(2)
        xt = torch.from_numpy(x.view(np.float32).reshape(-1, 2)).to(dev)
        res, err = [None, None], []
        def rank_main(r):
            try:
                g = xa.Group(xa.Demodulator.config("lrit", 6.25e6, D, front_exact=int(os.environ.get("FRONT_EXACT", "0"))), r, fabric=fabric)
                cap = n // D + 1024
                soft = torch.empty(cap, dtype=torch.float32, device=dev)
                sl = xt[r * n:(r + 1) * n].contiguous()
                k, off, pol = g.process_slice_device(sl.data_ptr(), n, soft.data_ptr(), cap)
                res[r] = (soft[:k].cpu().numpy(), off, pol, g.halo_samples, g.counters())
            except Exception as e:
                err.append(e)
        th = [threading.Thread(target=rank_main, args=(r,)) for r in range(2)]
        [t.start() for t in th]; [t.join(timeout=120) for t in th]
        if err: print("error", err); continue
        (s0, off0, pol0, _, _), (s1, off1, pol1, halo, relocks) = res
        w1 = want[len(s0):len(s0) + len(s1)]
        sgn = 1.0 if np.dot(s0[:40000], want[:40000]) > 0 else -1.0
        d = s1 - sgn * w1
        neq = np.nonzero(s1.view(np.uint32) != (sgn * w1).astype(np.float32).view(np.uint32))[0]
        last = int(neq[-1]) if len(neq) else -1
        win = [rms(d[i:i + 4096]) for i in range(0, len(d), 4096)]
        print("n %d halo %d phase0 %.1f: rank 0 %.3e (%d symbols, capture sign %+d), rank 1 %.3e over %d symbols (polarity %+d), "
              "(second starts, clock hand-overs, joined) %s, differing words %d, last at %d" % (n, halo, ph, rms(s0 - sgn * want[:len(s0)]), len(s0), int(sgn), rms(d), len(s1), pol1, relocks, len(neq), last), flush=True)
        print("   per 4096 symbols: " + " ".join("%.1e" % v for v in win[:24]), flush=True)
