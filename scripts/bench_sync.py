"""Frame-synchronisation front end (xrit_sync_correlate_device) on device-resident int8 soft symbols: GB/s against
the HBM peak, with the oracle's literal loops timed beside it on one host core (bounded sample)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import xritdemod_amd as xa
import oracle

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
host = sym[:64 * frame].cpu().numpy()
t0 = time.perf_counter()
ref = oracle.sync_correlate(host)
cpu_s = time.perf_counter() - t0
assert np.array_equal(hits[:64, :3].cpu().numpy().astype(np.uint32), ref)
print(json.dumps({"kernel": "sync_correlate", "frames": nf, "bytes": n, "ms": round(ms, 4),
                  "achieved_GBps": round(n / ms / 1e6, 1), "hbm_frac": round(n / ms / 1e6 / 8000.0, 4),
                  "Msymbols_per_s": round(n / ms / 1e3, 1),
                  "cpu_oracle_Msymbols_per_s": round(len(host) / cpu_s / 1e6, 2), "cpu_sample_frames": 64}))
