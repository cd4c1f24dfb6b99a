"""Do two chains on two streams overlap usefully?  (dev experiment)"""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import xritdemod_amd as xa
from xritdemod_amd import _capi

def run(nh, log2):
    n = 1 << log2
    D, fs = 5, 6.25e6
    dev = torch.device("cuda:0")
    nb = int(os.environ.get("STEPS", "12")) + 1
    bufs = torch.empty((nh, nb, n, 2), dtype=torch.float32, device=dev)
    st0 = torch.cuda.current_stream().cuda_stream
    for h in range(nh):
        sp = _capi.synth_params(fs_in=fs, seed=0x58524954 + 2 * h)
        for b in range(nb):
            _capi.synth_generate_device(sp, b * n, n, bufs[h, b].data_ptr(), device=0, stream=st0)
    torch.cuda.synchronize()
    dems = [xa.Demodulator(xa.Demodulator.config("lrit", fs, D)) for _ in range(nh)]
    cap = n // 20 + 64
    softs = [torch.empty((cap,), dtype=torch.float32, device=dev) for _ in range(nh)]
    def work(h, steps):
        for b in steps:
            dems[h].process_device(bufs[h, b].data_ptr(), n, softs[h].data_ptr(), cap)   # handle's own stream
    for h in range(nh):
        work(h, [0])
    torch.cuda.synchronize()
    steps = list(range(1, nb))          # consecutive bursts of one stream per handle (the loops stay locked)
    t0 = time.perf_counter()
    ths = [threading.Thread(target=work, args=(h, steps)) for h in range(nh)]
    for t in ths: t.start()
    for t in ths: t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{nh} handle(s) x {len(steps)} bursts of 2^{log2}: {dt*1e3:.2f} ms -> {nh*len(steps)*n/dt/1e6:.0f} Msamples/s", flush=True)

run(1, 28)
run(2, 28)
run(2, 27)
run(3, 27)
run(1, 28)
