"""Streamed (prefetched) against plain calls on edge cases: empty and tiny calls inside a stream, s16 / s8 ingest, a 3 dB signal (closure on its own,
strict mode), the fast and quick configurations, HRIT without a decimator -- symbols word for word and per-call statistics equal."""
import os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
import xritdemod_amd as xa
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.normpath(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), '../tests')))
import synth  # tests/synth.py: the NumPy specification of the synthetic burst (test infrastructure)
dev = torch.device("cuda", 0)
def compare(name, x_np, sizes, typ, fs, D, mode="lrit", **cfg):
    per = 2
    if typ == 0:
        xt = torch.from_numpy(x_np.view(np.float32)).to(dev); esz = 2      # floats per sample
    elif typ == 1:
        xi = np.clip(np.round(x_np.view(np.float32) * 32768), -32768, 32767).astype(np.int16); xt = torch.from_numpy(xi).to(dev); esz = 2
    else:
        xi = np.clip(np.round(x_np.view(np.float32) * 128), -128, 127).astype(np.int8); xt = torch.from_numpy(xi).to(dev); esz = 2
    offs = np.concatenate([[0], np.cumsum(sizes)])
    cap = max(max(sizes) // D, 1) + 4096
    soft = torch.empty((cap,), dtype=torch.float32, device=dev)
    def ptr(c): return xt[offs[c] * esz:].data_ptr() if offs[c] * esz < xt.numel() else xt.data_ptr()
    def run(pf):
        dem = xa.Demodulator(xa.Demodulator.config(mode, fs, D, **cfg))
        out, st = [], []
        if pf: dem.prefetch_device(ptr(0), sizes[0], sample_type=typ)
        for c in range(len(sizes)):
            if pf and c + 1 < len(sizes): dem.prefetch_device(ptr(c + 1), sizes[c + 1], sample_type=typ)
            k = dem.process_device(ptr(c), sizes[c], soft.data_ptr(), cap, sample_type=typ)
            out.append(soft[:k].cpu().numpy().copy()); s = dem.stats(); st.append((s.costas_passes, s.clock_passes, s.clock_relay_passes, s.clock_relay_closed, s.costas_unconverged))
        return out, st
    a, sa = run(False); b, sb = run(True)
    same = all(len(u) == len(v) and np.array_equal(u.view(np.uint32), v.view(np.uint32)) for u, v in zip(a, b))
    print(name, "symbols", sum(len(u) for u in a), "streamed == plain:", same, "stats equal:", sa == sb, flush=True)
    if not same or sa != sb:
        for c in range(len(sizes)):
            if len(a[c]) != len(b[c]) or not np.array_equal(a[c].view(np.uint32), b[c].view(np.uint32)) or sa[c] != sb[c]:
                print("   call", c, "n", sizes[c], len(a[c]), len(b[c]), sa[c], sb[c])
fs, D = 6.25e6, 5
x = synth.generate(synth.SynthParams(fs_in=fs), 9_000_000)
compare("empty and tiny calls", x, [1500000, 0, 3, 1500000, 4, 0, 2000000, 7, 1500000, 2499986], 0, fs, D)
compare("s16 ingest", x, [1500000, 1500000, 3000000, 1000000, 2000000], 1, fs, D)
compare("s8 ingest", x, [1500000, 1500000, 3000000, 1000000, 2000000], 2, fs, D)
xl = synth.generate(synth.SynthParams(fs_in=fs, esn0_db=3.0, seed=9), 9_000_000)
compare("3 dB (closure on its own)", xl, [3000000, 3000000, 3000000], 0, fs, D)
compare("3 dB strict", xl, [3000000, 3000000, 3000000], 0, fs, D, strict=1)
compare("fast configuration", x, [3000000, 3000000, 3000000], 0, fs, D, clock_exact=-2)
compare("quick relay", x, [3000000, 3000000, 3000000], 0, fs, D, clock_exact=-3)
x1 = synth.generate(synth.SynthParams(fs_in=2.5e6, symbol_rate=927000.0, alpha=0.3), 6_000_000)
compare("hrit d=1", x1, [2000000, 1000000, 3000000], 0, 2.5e6, 1, mode="hrit")
