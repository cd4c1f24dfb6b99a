#!/usr/bin/env python3
"""Where the soft-symbol difference against the CPU chain comes from.

For one burst (default C2's shape at 2^25 samples; --burst-log2 28 = the bench burst) it runs
  * the CPU oracle,
  * the device chain with the clock recovery as ONE serial trajectory (cfg.clock_serial): no hand-offs, so the
    difference is what any float32 M&M fed by this chain's own Costas output shows -- the floor,
  * the device chain time-tiled, for a list of chain lengths,
and prints one JSON object: rms / max / sign mismatches of each against the oracle and against the serial device
run, with the time per call.  Test infrastructure: it imports the oracle."""
import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, HERE)


def cmp(a, b):
    n = min(len(a), len(b))
    e = np.abs(a[:n].astype(np.float64) - b[:n].astype(np.float64))
    big = np.abs(b[:n]) > 1e-3
    return {"symbols": int(n), "count_equal": bool(len(a) == len(b)), "rms": float(np.sqrt(np.mean(e ** 2))),
            "max": float(e.max()), "sign_mismatches": int((np.sign(a[:n])[big] != np.sign(b[:n])[big]).sum())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--burst-log2", type=int, default=25)
    ap.add_argument("--decimation", type=int, default=5)
    ap.add_argument("--mode", default="lrit")
    ap.add_argument("--chains", default="0,64,112,192,256,512")
    ap.add_argument("--bursts", type=int, default=2, help="consecutive bursts; the last one is compared (steady state)")
    ap.add_argument("--esn0", type=float, default=12.0)
    ap.add_argument("--passes", default="", help="also: force the clock recovery to exactly these pass counts (e.g. 3,4,5,6,8)")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import torch
    import xritdemod_amd as xa
    from xritdemod_amd import _capi
    import oracle

    n = 1 << args.burst_log2
    D = args.decimation
    fs_in = (1.25e6 if args.mode == "lrit" else 2.5e6) * D
    sym_rate, alpha = (293883.0, 0.5) if args.mode == "lrit" else (927000.0, 0.3)
    dev = torch.device("cuda", 0)
    sp = _capi.synth_params(fs_in=fs_in, symbol_rate=sym_rate, alpha=alpha, esn0_db=args.esn0)
    stream = torch.cuda.current_stream(dev)
    bursts = torch.empty((args.bursts, n, 2), dtype=torch.float32, device=dev)
    for b in range(args.bursts):
        _capi.synth_generate_device(sp, b * n, n, bursts[b].data_ptr(), device=0, stream=stream.cuda_stream)
    torch.cuda.synchronize(dev)
    host = [bursts[b].cpu().numpy().view(np.complex64).reshape(-1) for b in range(args.bursts)]

    od = oracle.Demod(oracle.config(args.mode, fs_in, D))
    t0 = time.perf_counter()
    for b in range(args.bursts):
        want = od.process(host[b])
    t_cpu = (time.perf_counter() - t0) / args.bursts

    def run(**over):
        dem = xa.Demodulator(xa.Demodulator.config(args.mode, fs_in, D, device=0, **over))
        cap = int(n / (D * dem.sps * 0.99)) + 64
        soft = torch.empty((cap,), dtype=torch.float32, device=dev)
        ms = 0.0
        for b in range(args.bursts):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            ns = dem.process_device(bursts[b].data_ptr(), n, soft.data_ptr(), cap, stream=stream.cuda_stream)
            torch.cuda.synchronize(dev)
            ms = (time.perf_counter() - t0) * 1e3
        st = dem.stats()
        return soft[:ns].cpu().numpy(), ms, st

    out = {"workload": f"{args.mode} d={D} 2^{args.burst_log2} samples, burst {args.bursts} of one stream, Es/N0 {args.esn0} dB",
           "oracle_s_per_burst": round(t_cpu, 3)}
    ser, ms, st = run(clock_serial=1)
    out["serial_device"] = {"vs_oracle": cmp(ser, want), "ms_per_call": round(ms, 2)}
    out["tiled"] = {}
    for ns_ in [int(v) for v in args.chains.split(",")]:
        g, ms, st = run(clock_chain_syms=ns_, clock_exact=-2)       # (hand-off passes only: the fast configuration)
        out["tiled"][str(ns_)] = {"vs_oracle": cmp(g, want), "vs_serial_device": cmp(g, ser), "ms_per_call": round(ms, 3),
                                  "clock_passes": st.clock_passes, "Msamples_per_s": round(n / ms / 1e3, 1)}
    if args.passes:
        out["forced_passes"] = {}
        for p_ in [int(v) for v in args.passes.split(",")]:
            # max_passes caps both loops (the Costas loop needs 2 here), clock_min_passes keeps the stop rule from ending earlier
            g, ms, st = run(max_passes=max(p_, 3), clock_min_passes=p_, clock_exact=-2) if p_ >= 3 else run(max_passes=p_, clock_exact=-2)
            # (the cap also applies to the cold-started first burst, which may then lock with the other BPSK polarity)
            if len(g) == len(ser) and float(np.dot(g, ser)) < 0:
                g = -g
            out["forced_passes"][str(p_)] = {"vs_oracle": cmp(g, want), "vs_serial_device": cmp(g, ser), "ms_per_call": round(ms, 3),
                                             "clock_passes": st.clock_passes, "costas_passes": st.costas_passes}
    s = json.dumps(out, indent=1)
    print(s)
    if args.out:
        with open(args.out, "w") as f:
            f.write(s + "\n")


if __name__ == "__main__":
    main()
