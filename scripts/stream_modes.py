"""A handle in a given clock mode fed the way bench.py feeds it (the front end of burst b + 1 registered before burst b is
processed): for kernel traces of the exact / balanced legs.   python scripts/stream_modes.py --exact 3 --bursts 5"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import xritdemod_amd as xa
from xritdemod_amd import _capi


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2", type=int, default=28)
    ap.add_argument("--bursts", type=int, default=5)
    ap.add_argument("--exact", type=int, default=3)
    ap.add_argument("--no-prefetch", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    n, D, fs_in = 1 << args.log2, 5, 6.25e6
    sp = _capi.synth_params(fs_in=fs_in, symbol_rate=293883.0, alpha=0.5)
    stream = torch.cuda.current_stream(dev)
    bursts = torch.empty((args.bursts, n, 2), dtype=torch.float32, device=dev)
    for b in range(args.bursts):
        _capi.synth_generate_device(sp, b * n, n, bursts[b].data_ptr(), device=0, stream=stream.cuda_stream)
    dem = xa.Demodulator(xa.Demodulator.config("lrit", fs_in, D, clock_exact=args.exact))
    cap = int(n / (D * dem.sps * 0.99)) + 64
    soft = torch.empty((cap,), dtype=torch.float32, device=dev)
    torch.cuda.synchronize(dev)
    if not args.no_prefetch:
        dem.prefetch_device(bursts[0].data_ptr(), n, stream=stream.cuda_stream)
    for b in range(args.bursts):
        if not args.no_prefetch and b + 1 < args.bursts:
            dem.prefetch_device(bursts[b + 1].data_ptr(), n, stream=stream.cuda_stream)
        t0 = time.perf_counter()
        ns = dem.process_device(bursts[b].data_ptr(), n, soft.data_ptr(), cap, stream=stream.cuda_stream)
        torch.cuda.synchronize(dev)
        print(f"burst {b}: {ns} symbols, {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)


if __name__ == "__main__":
    main()
