import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import xritdemod_amd as xa
from xritdemod_amd import _capi
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream(dev)
def run(fs, D, log2, calls, prefetch):
    n = 1 << log2
    sp = _capi.synth_params(fs_in=fs)
    nbuf = 8
    buf = torch.empty((nbuf, n, 2), dtype=torch.float32, device=dev)
    for b in range(nbuf):
        _capi.synth_generate_device(sp, b * n, n, buf[b].data_ptr(), device=0, stream=st.cuda_stream)
    dem = xa.Demodulator(xa.Demodulator.config("lrit", fs, D, front_exact=int(os.environ.get("FRONT_EXACT", "0"))))
    cap = int(n / (D * dem.sps * 0.99)) + 64
    soft = torch.empty((cap,), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(dev)[0]
    t0 = time.perf_counter(); tot = 0; worst = (0, 0)
    depth = dem.prefetch_depth(n) if prefetch else 0
    for q in range(min(depth, calls)): dem.prefetch_device(buf[q % nbuf].data_ptr(), n, stream=st.cuda_stream)
    for c in range(calls):
        if prefetch and c + depth < calls: dem.prefetch_device(buf[(c + depth) % nbuf].data_ptr(), n, stream=st.cuda_stream)
        tot += dem.process_device(buf[c % nbuf].data_ptr(), n, soft.data_ptr(), cap, stream=st.cuda_stream)
        s = dem.stats()
        worst = (max(worst[0], s.costas_passes), max(worst[1], s.clock_relay_passes))
        if c == 20: free1 = torch.cuda.mem_get_info(dev)[0]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    free2 = torch.cuda.mem_get_info(dev)[0]
    print("fs %.3g D %d 2^%d x %d: %.3f ms/call, symbols %d, worst passes costas %d relay %d, device memory after 20 calls / at the end: %d / %d MB used since start" % (
        fs, D, log2, calls, dt / calls * 1e3, tot, worst[0], worst[1], (free0 - free1) >> 20, (free0 - free2) >> 20), flush=True)
run(6.25e6, 5, 28, int(os.environ.get("SOAK_BURSTS", "1500")), True)
run(1.25e6, 1, 17, 4000, False)
run(6.25e6, 5, 21, 3000, False)
