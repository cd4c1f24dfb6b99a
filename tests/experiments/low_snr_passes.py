"""Dev aid: loop pass counts of consecutive 8 Mi-sample calls at low Es/N0 (does a locked but noisy stream ever take
the tight pull-in tolerances of CostasPolicy::scale?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import xritdemod_amd as xa
from xritdemod_amd import _capi
n = 1 << 23
for esn0 in (12.0, 6.0, 4.0, 2.0, 0.0):
    sp = _capi.synth_params(fs_in=6.25e6, esn0_db=esn0)
    buf = torch.empty((n, 2), dtype=torch.float32, device="cuda:0")
    soft = torch.empty((n // 20 + 64,), dtype=torch.float32, device="cuda:0")
    dem = xa.Demodulator(xa.Demodulator.config("lrit", 6.25e6, 5))
    st = torch.cuda.current_stream().cuda_stream
    out = []
    for b in range(5):
        _capi.synth_generate_device(sp, b * n, n, buf.data_ptr(), device=0, stream=st)
        torch.cuda.synchronize()
        dem.process_device(buf.data_ptr(), n, soft.data_ptr(), len(soft), stream=st)
        s = dem.stats()
        out.append((s.costas_passes, s.clock_passes, s.costas_unconverged))
    print("Es/N0 %4.1f dB: (costas, clock, costas open) per call" % esn0, out)
