"""Round 6: the reset scenario of tests/test_gpu_parity.py::test_overlap_pipeline_survives_resets_refusals_and_abandoned_inputs in a
loop, with the timing perturbed (host sleeps, a competing stream), to catch the rare difference seen once inside the whole suite.
python scripts/r6_reset_stress.py [iterations] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import xritdemod_amd as xa
from xritdemod_amd import _capi
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
n, fs, nb = 1 << 23, 1.25e6, 4
dev = torch.device("cuda", 0)
buf = torch.empty((nb, n, 2), dtype=torch.float32, device=dev)
sp = _capi.synth_params(fs_in=fs)
for b in range(nb):
    _capi.synth_generate_device(sp, b * n, n, buf[b].data_ptr(), device=0, stream=torch.cuda.current_stream(dev).cuda_stream)
torch.cuda.synchronize()
cap = int(n / 4.2) + 4096
soft = torch.empty(cap, dtype=torch.float32, device=dev)
cfg = xa.Demodulator.config("lrit", fs, 1, front_exact=int(os.environ.get("FRONT_EXACT", "0")))
dem = xa.Demodulator(cfg)
plain = []
for b in range(nb):
    k = dem.process_device(buf[b].data_ptr(), n, soft.data_ptr(), cap)
    plain.append(soft[:k].cpu().numpy())
del dem
side = torch.cuda.Stream()
junk = torch.empty(1 << 28, dtype=torch.float32, device=dev)
bad = 0
d = xa.Demodulator(cfg)
for it in range(iters):
    if rng.random() < 0.3:
        d = xa.Demodulator(cfg)            # (a new handle now and then: the pool hands the streams on)
    with torch.cuda.stream(side):
        if rng.random() < 0.5:
            junk.mul_(1.0001)              # a competing kernel on another stream
    for b in range(3):
        d.prefetch_device(buf[b].data_ptr(), n)
    k = d.process_device(buf[0].data_ptr(), n, soft.data_ptr(), cap)
    ok0 = np.array_equal(soft[:k].cpu().numpy().view(np.uint32), plain[0].view(np.uint32))
    d.prefetch_device(buf[3].data_ptr(), n)
    if rng.random() < 0.5:
        time.sleep(float(rng.uniform(0, 0.004)))
    k = d.process_device(buf[1].data_ptr(), n, soft.data_ptr(), cap)
    ok1 = np.array_equal(soft[:k].cpu().numpy().view(np.uint32), plain[1].view(np.uint32))
    if rng.random() < 0.5:
        time.sleep(float(rng.uniform(0, 0.004)))
    d.reset()
    for b in range(2):
        d.prefetch_device(buf[b].data_ptr(), n)
    res = []
    for b in range(2):
        k = d.process_device(buf[b].data_ptr(), n, soft.data_ptr(), cap)
        g = soft[:k].cpu().numpy()
        same = k == len(plain[b]) and np.array_equal(g.view(np.uint32), plain[b].view(np.uint32))
        res.append((same, k, int(np.sum(g.view(np.uint32) != plain[b].view(np.uint32))) if k == len(plain[b]) else -1))
    d.reset()
    if not (ok0 and ok1 and res[0][0] and res[1][0]):
        bad += 1
        print("iteration", it, "streamed", ok0, ok1, "after reset", res, flush=True)
print("iterations", iters, "with a difference:", bad)
