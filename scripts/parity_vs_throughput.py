"""Parity against throughput of the clock recovery's modes on the bench workload (C2: LRIT, decimation 5, 2^28-sample
bursts of one stream): the tiled evaluation alone, the exact closure with n relay passes, the closure run to the
end for several segment lengths, and the serial wave.  Every row: time per steady-state burst, Gsamples/s, rms of
the soft symbols against the oracle and against the serial device trajectory, words that differ from it.
    python scripts/parity_vs_throughput.py [--log2 28] [--out profiles/r3_parity_vs_throughput.json]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import xritdemod_amd as xa
from xritdemod_amd import _capi
import oracle


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2", type=int, default=28)
    ap.add_argument("--bursts", type=int, default=3)
    ap.add_argument("--out", default="gpurun_out/r3_parity_vs_throughput.json")
    ap.add_argument("--no-oracle", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    n, D, fs_in = 1 << args.log2, 5, 6.25e6
    sp = _capi.synth_params(fs_in=fs_in)
    stream = torch.cuda.current_stream(dev)
    bursts = torch.empty((args.bursts, n, 2), dtype=torch.float32, device=dev)
    for b in range(args.bursts):
        _capi.synth_generate_device(sp, b * n, n, bursts[b].data_ptr(), device=0, stream=stream.cuda_stream)
    torch.cuda.synchronize(dev)
    want = None
    if not args.no_oracle:
        od = oracle.Demod(oracle.config("lrit", fs_in, D))
        t0 = time.perf_counter()
        want = [od.process(bursts[b].cpu().numpy().view(np.complex64).reshape(-1)) for b in range(args.bursts)]
        print(f"oracle: {time.perf_counter() - t0:.1f} s for {args.bursts} bursts", flush=True)

    def run(**kw):
        dem = xa.Demodulator(xa.Demodulator.config("lrit", fs_in, D, **kw))
        cap = int(n / (D * dem.sps * 0.99)) + 64
        soft = torch.empty((cap,), dtype=torch.float32, device=dev)
        outs, ms, stats = [], [], []
        for b in range(args.bursts):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            ns = dem.process_device(bursts[b].data_ptr(), n, soft.data_ptr(), cap, stream=stream.cuda_stream)
            torch.cuda.synchronize(dev)
            ms.append((time.perf_counter() - t0) * 1e3)
            stats.append(dem.stats())
            outs.append(soft[:ns].cpu().numpy().copy())
        return outs, ms, stats

    ser, ms_ser, _ = run(clock_serial=1)
    rows = []

    def row(tag, outs, ms, stats, **kw):
        steady = range(1, args.bursts)             # burst 0 is the cold start
        g = np.concatenate([outs[b] for b in steady])
        s = np.concatenate([ser[b] for b in steady])
        r = {"mode": tag, **kw, "ms_per_burst": round(float(np.mean([ms[b] for b in steady])), 3),
             "clock_passes": [int(stats[b].clock_passes) for b in steady],
             "relay_passes": [int(stats[b].clock_relay_passes) for b in steady],
             "relay_closed": [int(stats[b].clock_relay_closed) for b in steady],
             "relay_segments": int(stats[-1].clock_relay_segments)}
        r["Gsamples_per_s"] = round(n / r["ms_per_burst"] / 1e6, 2)
        if len(g) == len(s):
            r["rms_vs_serial_device"] = float(np.sqrt(np.mean((g - s) ** 2)))
            r["words_differing_from_serial_device"] = int((g.view(np.uint32) != s.view(np.uint32)).sum())
        else:
            r["symbol_count_differs"] = [len(g), len(s)]
        if want is not None:
            w = np.concatenate([want[b] for b in steady])
            if len(w) == len(g):
                big = np.abs(w) > 1e-3
                r["rms_vs_oracle"] = float(np.sqrt(np.mean((g - w) ** 2)))
                r["sign_mismatches_vs_oracle"] = int((np.sign(g[big]) != np.sign(w[big])).sum())
        r["symbols"] = int(len(g))
        rows.append(r)
        print(json.dumps(r), flush=True)

    row("serial wave (cfg.clock_serial)", ser, ms_ser, run(clock_serial=1)[2])
    o, m, st = run(clock_exact=-1)
    row("tiled evaluation only (cfg.clock_exact = -1 / -2: the fast configuration)", o, m, st)
    o, m, st = run()
    row("default configuration (cfg.clock_exact = 0)", o, m, st)
    for passes in (2, 3, 4, 6, 8, 12, 16, 24):
        o, m, st = run(clock_exact=passes)
        row("exact closure stopped after n relay passes", o, m, st, relay_pass_budget=passes)
    for window in (0, 37, 74, 296, 592, 1184, 4736):
        o, m, st = run(clock_exact=1, clock_exact_window=window)
        row("exact closure (cfg.clock_exact = 1)", o, m, st, window_chains=window or "auto (3 segments per CU)")
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    json.dump({"workload": f"C2: LRIT d=5, {args.bursts} consecutive bursts of 2^{args.log2} cf32 samples, Es/N0 12 dB; steady-state "
                           "bursts (all but the cold-started first) are timed and compared",
               "note": "ms_per_burst is the wall time of xrit_demod_process_device, one burst at a time (no front-end prefetch)",
               "rows": rows}, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
