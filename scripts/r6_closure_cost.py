import os, sys, time
sys.path.insert(0, "/root/repo")
import torch, numpy as np
import xritdemod_amd as xa
from xritdemod_amd import _capi
dev = torch.device("cuda", 0); st = torch.cuda.current_stream(dev)
fs, D = 1.25e6, 1
for log2 in (19, 20, 21, 22):
    n = 1 << log2
    sp = _capi.synth_params(fs_in=fs)
    nb = 12
    buf = torch.empty((nb, n, 2), dtype=torch.float32, device=dev)
    for b in range(nb): _capi.synth_generate_device(sp, b * n, n, buf[b].data_ptr(), device=0, stream=st.cuda_stream)
    torch.cuda.synchronize()
    for ce in (0, 1):
        dem = xa.Demodulator(xa.Demodulator.config("lrit", fs, D, clock_exact=ce))
        cap = n // 4 + 4096
        soft = torch.empty(cap, dtype=torch.float32, device=dev)
        for b in range(3): dem.process_device(buf[b].data_ptr(), n, soft.data_ptr(), cap)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for b in range(3, nb): dem.process_device(buf[b].data_ptr(), n, soft.data_ptr(), cap)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / (nb - 3) * 1e3
        s = dem.stats()
        print("2^%d samples (%d k symbols) clock_exact %d: %.2f ms per call, relay passes %d closed %d" % (log2, n / 4.25 / 1000, ce, dt, s.clock_relay_passes, s.clock_relay_closed), flush=True)
