"""Latency of one chain call on device-resident input at the reference's chunk sizes (32 Ki - 512 Ki samples per
processSamples call, demodulator.cpp:113): the fixed cost of a call, which dominates there."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import xritdemod_amd as xa
from xritdemod_amd import _capi
for fs, D in ((1.25e6, 1), (6.25e6, 5)):
    for log2 in (15, 17, 19, 21):
        n = 1 << log2
        sp = _capi.synth_params(fs_in=fs)
        buf = torch.empty((40, n, 2), dtype=torch.float32, device="cuda:0")
        st = torch.cuda.current_stream().cuda_stream
        for b in range(40):
            _capi.synth_generate_device(sp, b * n, n, buf[b].data_ptr(), device=0, stream=st)
        dem = xa.Demodulator(xa.Demodulator.config("lrit", fs, D, clock_chain_syms=int(__import__("os").environ.get("NS", "0")),
                                                          front_exact=int(__import__("os").environ.get("FRONT_EXACT", "0"))))
        soft = torch.empty((n,), dtype=torch.float32, device="cuda:0")
        for b in range(10):
            dem.process_device(buf[b].data_ptr(), n, soft.data_ptr(), n, stream=st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for b in range(10, 40):
            dem.process_device(buf[b].data_ptr(), n, soft.data_ptr(), n, stream=st)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 30
        s = dem.stats()
        print(f"D={D} chunk 2^{log2}: {dt*1e3:.3f} ms/call  {n/dt/1e6:.0f} Msamples/s  passes costas {s.costas_passes} clock {s.clock_passes}")
