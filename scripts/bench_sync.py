"""Frame-synchronisation front end (xrit_sync_correlate_device) on device-resident int8 soft symbols: GB/s against
the HBM peak.  (Equality with the CPU oracle is tests/test_gpu_parity.py::test_sync_correlator_bit_exact; the
oracle's own rate, 41 Msymbols/s on one host core, was taken with tests/experiments/sync_cpu_rate.py.)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import xritdemod_amd as xa

frame = 16384
nf = 1 << 16                                   # 65536 frames = 1 GiB of soft symbols
n = nf * frame
g = torch.Generator(device="cuda:0"); g.manual_seed(7)
sym = torch.randint(-128, 128, (n,), dtype=torch.int8, device="cuda:0", generator=g)
hits = torch.zeros((nf, 4), dtype=torch.int32, device="cuda:0")
st = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    xa.sync_correlate_device(sym.data_ptr(), n, hits.data_ptr(), stream=st)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 10
a.record()
for _ in range(reps):
    xa.sync_correlate_device(sym.data_ptr(), n, hits.data_ptr(), stream=st)
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b) / reps
print(json.dumps({"kernel": "sync_correlate", "frames": nf, "bytes": n, "ms": round(ms, 4),
                  "achieved_GBps": round(n / ms / 1e6, 1), "hbm_frac": round(n / ms / 1e6 / 8000.0, 4),
                  "Msymbols_per_s": round(n / ms / 1e3, 1)}))
