import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, oracle
import xritdemod_amd as xa
from xritdemod_amd import _capi
n = 1 << 25
D, fs = 5, 6.25e6
sp = _capi.synth_params(fs_in=fs)
bufs = torch.empty((2, n, 2), dtype=torch.float32, device="cuda:0")
st = torch.cuda.current_stream().cuda_stream
for b in range(2):
    _capi.synth_generate_device(sp, b * n, n, bufs[b].data_ptr(), device=0, stream=st)
torch.cuda.synchronize()
host = [bufs[b].cpu().numpy().view(np.complex64).reshape(-1) for b in range(2)]
od = oracle.Demod(oracle.config("lrit", fs, D))
want = [od.process(h) for h in host]
cap = n // 20 + 64
soft = torch.empty((cap,), dtype=torch.float32, device="cuda:0")
for mp in (3, 4, 5, 6, 8, 12, 20):
    dem = xa.Demodulator(xa.Demodulator.config("lrit", fs, D, max_passes=mp, clock_min_passes=mp))
    res = []
    for b in range(2):
        t0 = time.perf_counter()
        ns = dem.process_device(bufs[b].data_ptr(), n, soft.data_ptr(), cap, stream=st)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        g = soft[:ns].cpu().numpy()
        e = np.abs(g - want[b][:ns])
        res.append((ns == len(want[b]), float(np.sqrt(np.mean(e**2))), float(e.max()), (t1 - t0) * 1e3))
    s = dem.stats()
    print(f"passes={mp} jac={os.environ.get('XRIT_UNUSED','2')}: burst0 ok={res[0][0]} rms={res[0][1]:.2e} | burst1 (steady) ok={res[1][0]} rms={res[1][1]:.2e} max={res[1][2]:.1e} ms={res[1][3]:.2f} costas={s.costas_passes}")
