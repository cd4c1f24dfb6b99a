"""Wall time of every streamed C2 step of one run (bench.py's feeding order): where a run's average comes from."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import xritdemod_amd as xa
from xritdemod_amd import _capi
n, D, fs = 1 << 28, 5, 6.25e6
K = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda", 0)
sp = _capi.synth_params(fs_in=fs)
nbuf = 30
bursts = torch.empty((nbuf, n, 2), dtype=torch.float32, device=dev)
st = torch.cuda.current_stream(dev)
for b in range(nbuf):
    _capi.synth_generate_device(sp, b * n, n, bursts[b].data_ptr(), device=0, stream=st.cuda_stream)
torch.cuda.synchronize()
dem = xa.Demodulator(xa.Demodulator.config("lrit", fs, D))
cap = int(n / (D * dem.sps * 0.99)) + 64
soft = torch.empty((cap,), dtype=torch.float32, device=dev)
WARM_STREAMED = os.environ.get("WARM_STREAMED", "1") == "1"       # warm-up steps fed like the timed ones (both buffer sets get allocated)
if WARM_STREAMED:
    dem.prefetch_device(bursts[0].data_ptr(), n, stream=st.cuda_stream)
for b in range(5):
    if WARM_STREAMED and b + 1 < 5:
        dem.prefetch_device(bursts[(b + 1) % nbuf].data_ptr(), n, stream=st.cuda_stream)
    dem.process_device(bursts[b % nbuf].data_ptr(), n, soft.data_ptr(), cap, stream=st.cuda_stream)
torch.cuda.synchronize()
ts = []
dem.prefetch_device(bursts[5 % nbuf].data_ptr(), n, stream=st.cuda_stream)
for b in range(5, 5 + K):
    t0 = time.perf_counter()
    if b + 1 < 5 + K:
        dem.prefetch_device(bursts[(b + 1) % nbuf].data_ptr(), n, stream=st.cuda_stream)
    dem.process_device(bursts[b % nbuf].data_ptr(), n, soft.data_ptr(), cap, stream=st.cuda_stream)
    ts.append((time.perf_counter() - t0) * 1e3)
torch.cuda.synchronize()
print("steps", K, "mean %.3f" % (sum(ts) / K), "first ten", " ".join("%.2f" % t for t in ts[:10]))
for i in range(10, K, 10):
    print("  steps %d..%d: mean %.3f" % (i, i + 9, sum(ts[i:i + 10]) / len(ts[i:i + 10])))
