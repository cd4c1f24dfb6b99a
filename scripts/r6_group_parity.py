"""Round 6: what the ranks of a contiguous split (xrit_group_*, in-process fabric, two ranks on one GPU) are from the uninterrupted
CPU chain, per rank and lock polarity, with the default configuration (bit-exact front end on slices of this size) and with the
fast front end (cfg.front_exact = -1).   python scripts/r6_group_parity.py"""
import os, sys, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import oracle, synth
import xritdemod_amd as xa
def rms(a): return float(np.sqrt(np.mean(np.abs(a) ** 2))) if len(a) else 0.0
n, D = 900000, 5
dev = torch.device("cuda", 0)
for fe in (0, -1, 2):
    for ph in (0.7, 2.3, 3.9, 5.4):
        x = synth.generate(synth.SynthParams(fs_in=6.25e6, phase0=ph, seed=4242), 2 * n)
        want = oracle.Demod(oracle.config("lrit", 6.25e6, D)).process(x)
        fabric = xa.LocalFabric(2)
        xt = torch.from_numpy(x.view(np.float32).reshape(-1, 2)).to(dev)
        res, err = [None, None], []
        def rank_main(r):
            try:
                g = xa.Group(xa.Demodulator.config("lrit", 6.25e6, D, front_exact=fe), r, fabric=fabric)
                cap = n // D + 1024
                soft = torch.empty(cap, dtype=torch.float32, device=dev)
                sl = xt[r * n:(r + 1) * n].contiguous()
                k, off, pol = g.process_slice_device(sl.data_ptr(), n, soft.data_ptr(), cap)
                res[r] = (soft[:k].cpu().numpy(), off, pol)
            except Exception as e:
                err.append(e)
        th = [threading.Thread(target=rank_main, args=(r,)) for r in range(2)]
        [t.start() for t in th]; [t.join(timeout=120) for t in th]
        if err: print("error", err); continue
        (s0, off0, pol0), (s1, off1, pol1) = res
        print("front_exact %2d phase0 %.1f: rank 0 %.3e (%d symbols), rank 1 %.3e (polarity %+d)" % (fe, ph, rms(s0 - want[:len(s0)]), len(s0), rms(s1 - want[len(s0):len(s0) + len(s1)]), pol1), flush=True)
