"""Exact closure on device-resident bursts of one stream: time per burst, relay passes, bitwise comparison with the
serial device trajectory (optional: ~4 s per 2^28-sample burst).
    python scripts/relay_burst.py --log2 28 --bursts 3 [--serial] [--window W ...] [--exact N]"""
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
    ap.add_argument("--bursts", type=int, default=3)
    ap.add_argument("--serial", action="store_true")
    ap.add_argument("--window", type=int, nargs="*", default=[0])
    ap.add_argument("--exact", type=int, nargs="*", default=[1])
    ap.add_argument("--mode", default="lrit")
    ap.add_argument("--decimation", type=int, default=5)
    ap.add_argument("--prof", action="store_true")
    ap.add_argument("--esn0", type=float, default=None, help="Es/N0 of the synthetic stream (dB; default: the generator's 12)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    n = 1 << args.log2
    D = args.decimation
    fs_in = (1.25e6 if args.mode == "lrit" else 2.5e6) * D
    sym_rate, alpha = (293883.0, 0.5) if args.mode == "lrit" else (927000.0, 0.3)
    sp = _capi.synth_params(fs_in=fs_in, symbol_rate=sym_rate, alpha=alpha, **({"esn0_db": args.esn0} if args.esn0 is not None else {}))
    stream = torch.cuda.current_stream(dev)
    bursts = torch.empty((args.bursts, n, 2), dtype=torch.float32, device=dev)
    for b in range(args.bursts):
        _capi.synth_generate_device(sp, b * n, n, bursts[b].data_ptr(), device=0, stream=stream.cuda_stream)
    torch.cuda.synchronize(dev)

    def run(tag, **kw):
        dem = xa.Demodulator(xa.Demodulator.config(args.mode, fs_in, D, **kw))
        cap = int(n / (D * dem.sps * 0.99)) + 64
        soft = torch.empty((cap,), dtype=torch.float32, device=dev)
        outs = []
        for b in range(args.bursts):
            if args.prof and b == args.bursts - 1:
                dem.profile(1)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            ns = dem.process_device(bursts[b].data_ptr(), n, soft.data_ptr(), cap, stream=stream.cuda_stream)
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t0
            st = dem.stats()
            outs.append(soft[:ns].cpu().numpy().copy())
            print(f"{tag} burst {b}: {ns} symbols, {dt * 1e3:.2f} ms ({n / dt / 1e9:.2f} Gsamples/s), clock passes {st.clock_passes}, "
                  f"relay passes {st.clock_relay_passes} closed {st.clock_relay_closed} segments {st.clock_relay_segments}", flush=True)
        if args.prof:
            for name, ms, c in dem.profile_read():
                print(f"    {name:20s} {ms:9.3f} ms / {c}")
        return outs

    fast = run("fast ", clock_exact=-2)
    ser = run("serial", clock_serial=1) if args.serial else None
    for ex in args.exact:
        for w in args.window:
            got = run(f"exact={ex} window={w}", clock_exact=ex, clock_exact_window=w)
            for b in range(args.bursts):
                ref = ser[b] if ser is not None else None
                if ref is not None and len(ref) == len(got[b]):
                    nd = int(np.sum(ref.view(np.uint32) != got[b].view(np.uint32)))
                    print(f"   burst {b}: vs serial: differing words {nd}, rms {np.sqrt(np.mean((ref - got[b]) ** 2)):.3e}; "
                          f"fast vs serial rms {np.sqrt(np.mean((ref - fast[b]) ** 2)):.3e}")
                elif ref is not None:
                    print(f"   burst {b}: symbol count differs: {len(got[b])} vs serial {len(ref)}")
                elif len(fast[b]) == len(got[b]):
                    print(f"   burst {b}: vs fast rms {np.sqrt(np.mean((fast[b] - got[b]) ** 2)):.3e}")


if __name__ == "__main__":
    main()
