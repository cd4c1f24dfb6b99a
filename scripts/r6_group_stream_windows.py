"""Round 6: the call-after-call capture of tests/test_gpu_parity.py::test_group_streams_a_capture_call_after_call, slice by slice:
rms against the uninterrupted CPU chain per window of 4096 symbols, differing words, second starts.
   python scripts/r6_group_stream_windows.py"""
import os, sys, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import oracle, synth
import xritdemod_amd as xa
def rms(a): return float(np.sqrt(np.mean(np.abs(a) ** 2))) if len(a) else 0.0
n, D, calls = int(os.environ.get("GROUP_N", "1300000")), 5, int(os.environ.get("GROUP_CALLS", "3"))
fe = int(os.environ.get("FRONT_EXACT", "0"))
x = synth.generate(synth.SynthParams(fs_in=6.25e6), 2 * calls * n)
want = oracle.Demod(oracle.config("lrit", 6.25e6, D)).process(x)
fabric = xa.LocalFabric(2)
dev = torch.device("cuda", 0)
xt = torch.from_numpy(x.view(np.float32).reshape(-1, 2)).to(dev)
parts, err = {}, []
def rank_main(r):
    try:
        g = xa.Group(xa.Demodulator.config("lrit", 6.25e6, D, front_exact=fe), r, fabric=fabric)
        cap = n // D + 1024
        soft = torch.empty(cap, dtype=torch.float32, device=dev)
        for c in range(calls):
            sl = xt[(2 * c + r) * n:(2 * c + r + 1) * n].contiguous()
            before = g.counters()
            k, off, pol = g.process_slice_device(sl.data_ptr(), n, soft.data_ptr(), cap)
            st = g.chain_stats() if hasattr(g, "chain_stats") else None
            parts[(c, r)] = (soft[:k].cpu().numpy().copy(), off, pol, tuple(a - b for a, b in zip(g.counters(), before)))
    except Exception as e:
        err.append(e)
th = [threading.Thread(target=rank_main, args=(r,)) for r in range(2)]
[t.start() for t in th]; [t.join(timeout=300) for t in th]
assert not err, err
pos = 0
for c in range(calls):
    for r in range(2):
        s, off, pol, rl = parts[(c, r)]
        w = want[pos:pos + len(s)]
        d = s - w
        neq = np.nonzero(s.view(np.uint32) != w.view(np.uint32))[0]
        print("call %d rank %d: %d symbols at %d (offset %d), first lock %+d, (second starts, clock hand-overs, joined) %s: rms %.3e, differing words %d, first %d last %d"
              % (c, r, len(s), pos, off, pol, rl, rms(d), len(neq), int(neq[0]) if len(neq) else -1, int(neq[-1]) if len(neq) else -1), flush=True)
        print("   per 4096 symbols: " + " ".join("%.1e" % rms(d[i:i + 4096]) for i in range(0, len(d), 4096)), flush=True)
        pos += len(s)
print("total %d symbols (CPU chain %d)" % (pos, len(want)))
