#!/usr/bin/env python3
"""Wall time of every streamed step of the default configuration, fed like bench.py feeds it (two inputs registered behind the call
in progress, nothing registered before the clock starts): the pipeline's fill and its steady state.
Usage: python scripts/r5_step_times.py [--steps 30] [--decimation 5] [--mode lrit]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import xritdemod_amd as xa
from xritdemod_amd import _capi

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--warmup", type=int, default=5)
ap.add_argument("--decimation", type=int, default=5)
ap.add_argument("--mode", default="lrit")
ap.add_argument("--log2", type=int, default=28)
a = ap.parse_args()
D, n = a.decimation, 1 << a.log2
fs = (1.25e6 if a.mode == "lrit" else 2.5e6) * D
kw = dict(fs_in=fs) if a.mode == "lrit" else dict(fs_in=fs, symbol_rate=927000.0, alpha=0.3)
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream(dev)
nb = a.steps + a.warmup
nbuf = min(nb, 24)
buf = torch.empty((nbuf, n, 2), dtype=torch.float32, device=dev)
sp = _capi.synth_params(**kw)
for b in range(nbuf):
    _capi.synth_generate_device(sp, b * n, n, buf[b].data_ptr(), device=0, stream=st.cuda_stream)
torch.cuda.synchronize(dev)
dem = xa.Demodulator(xa.Demodulator.config(a.mode, fs, D))
cap = int(n / (D * 2.6)) + 4096
soft = torch.empty((cap,), dtype=torch.float32, device=dev)
def reg(b): dem.prefetch_device(buf[b % nbuf].data_ptr(), n, stream=st.cuda_stream)
def go(b): return dem.process_device(buf[b % nbuf].data_ptr(), n, soft.data_ptr(), cap, stream=st.cuda_stream)
W = a.warmup
for q in range(min(2, W)): reg(q)
for b in range(W):
    if b + 2 < W: reg(b + 2)
    go(b)
torch.cuda.synchronize(dev)
t = [time.perf_counter()]
for q in range(2): reg(W + q)
for b in range(W, nb):
    if b + 2 < nb: reg(b + 2)
    go(b)
    t.append(time.perf_counter())
torch.cuda.synchronize(dev)
ms = [round((y - x) * 1e3, 3) for x, y in zip(t[:-1], t[1:])]
print(json.dumps({"ms_per_step": ms, "mean_all": round(sum(ms) / len(ms), 3), "mean_from_4": round(sum(ms[4:]) / len(ms[4:]), 3)}))
