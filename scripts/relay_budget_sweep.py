"""Partial closure: relay-pass budget x segment size (cfg.clock_exact = n, cfg.clock_exact_window) against the serial
device trajectory on steady-state C2 bursts -- which cut reaches a given parity in the least time.
    python scripts/relay_budget_sweep.py [--log2 28]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import xritdemod_amd as xa
from xritdemod_amd import _capi


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2", type=int, default=28)
    ap.add_argument("--bursts", type=int, default=4)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    n, D, fs_in = 1 << args.log2, 5, 6.25e6
    sp = _capi.synth_params(fs_in=fs_in, symbol_rate=293883.0, alpha=0.5)
    stream = torch.cuda.current_stream(dev)
    bursts = torch.empty((args.bursts, n, 2), dtype=torch.float32, device=dev)
    for b in range(args.bursts):
        _capi.synth_generate_device(sp, b * n, n, bursts[b].data_ptr(), device=0, stream=stream.cuda_stream)
    torch.cuda.synchronize(dev)

    def run(**kw):
        dem = xa.Demodulator(xa.Demodulator.config("lrit", fs_in, D, **kw))
        cap = int(n / (D * dem.sps * 0.99)) + 64
        soft = torch.empty((cap,), dtype=torch.float32, device=dev)
        outs, ms = [], []
        for b in range(args.bursts):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            ns = dem.process_device(bursts[b].data_ptr(), n, soft.data_ptr(), cap, stream=stream.cuda_stream)
            torch.cuda.synchronize(dev)
            ms.append((time.perf_counter() - t0) * 1e3)
            outs.append(soft[:ns].cpu().numpy().copy())
        return outs, ms, dem.stats()

    ser, _, _ = run(clock_serial=1)
    steady = list(range(1, args.bursts))
    s = np.concatenate([ser[b] for b in steady])
    print("budget window segments   ms(min)   rms vs serial")
    for window in (0, 222, 296, 444, 592):
        for budget in (-1, 2, 3, 4):
            if budget < 0 and window:
                continue
            o, ms, st = run(clock_exact=budget, clock_exact_window=window)
            g = np.concatenate([o[b] for b in steady])
            print(f"{budget:6d} {window:6d} {st.clock_relay_segments:8d} {min(ms[b] for b in steady):9.3f}   "
                  f"{np.sqrt(np.mean((g - s) ** 2)):.3e}", flush=True)


if __name__ == "__main__":
    main()
