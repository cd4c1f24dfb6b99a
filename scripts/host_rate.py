"""PCIe-inclusive rate of the host entry point xrit_demod_process: host cf32 buffer in,
host soft symbols out, one call per chunk, consecutive chunks of ONE continuous stream (made on the device, copied
to pageable host memory before the clock starts)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import xritdemod_amd as xa
from xritdemod_amd import _capi
fs, D = 6.25e6, 5
sp = _capi.synth_params(fs_in=fs)
st = torch.cuda.current_stream().cuda_stream
for log2 in (19, 22, 24, 26):
    n = 1 << log2
    reps = max(3, min(24, (1 << 27) // n))
    dev = torch.empty((n, 2), dtype=torch.float32, device="cuda:0")
    chunks = []
    for c in range(reps + 2):
        _capi.synth_generate_device(sp, c * n, n, dev.data_ptr(), device=0, stream=st)
        torch.cuda.synchronize()
        chunks.append(dev.cpu().numpy().view(np.complex64).reshape(-1).copy())
    del dev
    dem = xa.Demodulator(xa.Demodulator.config("lrit", fs, D))
    dem.process(chunks[0])
    dem.process(chunks[1])
    t0 = time.perf_counter()
    for c in range(2, reps + 2):
        dem.process(chunks[c])
    dt = (time.perf_counter() - t0) / reps
    s = dem.stats()
    print(f"chunk 2^{log2} samples ({n*8/2**20:.0f} MiB): {dt*1e3:.2f} ms/call, {n/dt/1e6:.0f} Msamples/s host to host"
          f" (passes costas {s.costas_passes} clock {s.clock_passes})")
