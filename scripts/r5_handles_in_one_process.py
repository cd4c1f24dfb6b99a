"""Bursts per handle when several handles live in one process (C2, fed like bench.py): which hardware queues HIP deals a handle's
streams onto decides whether its pipeline overlaps (csrc/demod.cpp, create_own_queue_stream).  python scripts/r5_handles_in_one_process.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xritdemod_amd as xa
from xritdemod_amd import _capi
dev = torch.device("cuda", 0); st = torch.cuda.current_stream(dev)
n, D, fs = 1 << 28, 5, 6.25e6
nb = 12
sp = _capi.synth_params(fs_in=fs)
bursts = torch.empty((nb, n, 2), dtype=torch.float32, device=dev)
for b in range(nb): _capi.synth_generate_device(sp, b * n, n, bursts[b].data_ptr(), device=0, stream=st.cuda_stream)
torch.cuda.synchronize()
cap = n // (D * 4) + 4096
soft = torch.empty((cap,), dtype=torch.float32, device=dev)
def run(dem, K=20, W=4):
    for b in range(W): dem.process_device(bursts[b % nb].data_ptr(), n, soft.data_ptr(), cap, stream=st.cuda_stream)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for q in range(2): dem.prefetch_device(bursts[(W + q) % nb].data_ptr(), n, stream=st.cuda_stream)
    for b in range(W, W + K):
        if b + 2 < W + K: dem.prefetch_device(bursts[(b + 2) % nb].data_ptr(), n, stream=st.cuda_stream)
        dem.process_device(bursts[b % nb].data_ptr(), n, soft.data_ptr(), cap, stream=st.cuda_stream)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / K * 1e3
handles = []
for i in range(8):
    h = xa.Demodulator(xa.Demodulator.config("lrit", fs, D)); handles.append(h)
    print("handle", i, "(earlier handles alive): %.3f ms per burst" % run(h), flush=True)
print("handle 0 again: %.3f" % run(handles[0]))
del handles
h = xa.Demodulator(xa.Demodulator.config("lrit", fs, D))
print("a handle after the others were destroyed: %.3f" % run(h))
