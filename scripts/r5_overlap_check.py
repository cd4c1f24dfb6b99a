#!/usr/bin/env python3
"""Round 5: the clock recovery as overlapping exactly walked blocks (csrc/clock_overlap.h) against the serial device trajectory
and the relay of round 4, on consecutive bursts of one stream.  Checks, per configuration:
  * symbol count and hard decisions equal to the serial trajectory's (cfg.clock_serial) on every burst;
  * rms against the serial trajectory (and against the round-4 relay: XRIT_NO_OVERLAP=1 in a second process is not possible inside
    one process -- the switch is read at create -- so the relay's figure comes from cfg.clock_exact = 3 where the plan allows);
  * the streamed run (two inputs registered ahead) gives the words of plain consecutive calls;
  * two runs give the same words.
Usage: python scripts/r5_overlap_check.py [--case C2] [--log2 26] [--bursts 4] [--oracle]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import xritdemod_amd as xa  # noqa: E402
from xritdemod_amd import _capi  # noqa: E402

CASES = {"C1": ("lrit", 1.25e6, 1, dict(fs_in=1.25e6)), "C2": ("lrit", 6.25e6, 5, dict(fs_in=6.25e6)),
         "C3": ("hrit", 2.5e6, 1, dict(fs_in=2.5e6, symbol_rate=927000.0, alpha=0.3)), "C5": ("lrit", 40e6, 32, dict(fs_in=40e6))}


def rms(a):
    a = np.asarray(a, np.float64)
    return float(np.sqrt(np.mean(a * a))) if a.size else 0.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="C2")
    ap.add_argument("--log2", type=int, default=26)
    ap.add_argument("--bursts", type=int, default=4)
    ap.add_argument("--oracle", action="store_true")
    ap.add_argument("--no-serial", action="store_true")
    args = ap.parse_args()
    mode, fs, D, kw = CASES[args.case]
    n = 1 << args.log2
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream(dev)
    sp = _capi.synth_params(**kw)
    bufs = torch.empty((args.bursts, n, 2), dtype=torch.float32, device=dev)
    for b in range(args.bursts):
        _capi.synth_generate_device(sp, b * n, n, bufs[b].data_ptr(), device=0, stream=st.cuda_stream)
    torch.cuda.synchronize(dev)
    cfg = lambda **k: xa.Demodulator.config(mode, fs, D, **k)  # noqa: E731
    cap = int(n / (D * 2.6)) + 4096
    soft = torch.empty((cap,), dtype=torch.float32, device=dev)

    def run(dem, streamed):
        out, stats, t = [], [], []
        if streamed:
            for q in range(min(2, args.bursts)):
                dem.prefetch_device(bufs[q].data_ptr(), n, stream=st.cuda_stream)
        for b in range(args.bursts):
            if streamed and b + 2 < args.bursts:
                dem.prefetch_device(bufs[b + 2].data_ptr(), n, stream=st.cuda_stream)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            k = dem.process_device(bufs[b].data_ptr(), n, soft.data_ptr(), cap, stream=st.cuda_stream)
            torch.cuda.synchronize(dev)
            t.append((time.perf_counter() - t0) * 1e3)
            out.append(soft[:k].cpu().numpy().copy())
            s = dem.stats()
            stats.append((int(s.clock_passes), int(s.clock_relay_passes), int(s.clock_relay_segments), int(s.clock_relay_closed)))
        return out, stats, t

    res = {"case": args.case, "samples_per_burst": n, "bursts": args.bursts}
    plain, sp_, tp = run(xa.Demodulator(cfg()), False)
    again, _, _ = run(xa.Demodulator(cfg()), False)
    strm, ss_, ts = run(xa.Demodulator(cfg()), True)
    res["plan"] = sp_
    res["ms_plain"] = [round(v, 3) for v in tp]
    res["ms_streamed_calls"] = [round(v, 3) for v in ts]
    res["run_to_run_identical"] = all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(plain, again))
    res["streamed_equals_plain"] = [bool(len(a) == len(b) and np.array_equal(a.view(np.uint32), b.view(np.uint32))) for a, b in zip(plain, strm)]
    res["counts_plain_streamed"] = [(len(a), len(b)) for a, b in zip(plain, strm)]
    res["streamed_vs_plain_rms"] = [rms(a - b) if len(a) == len(b) else None for a, b in zip(plain, strm)]
    if not args.no_serial:
        ser, _, _ = run(xa.Demodulator(cfg(clock_serial=1)), False)
        res["count_equal_serial"] = [len(a) == len(b) for a, b in zip(plain, ser)]
        res["vs_serial_rms"] = [rms(a - b) if len(a) == len(b) else None for a, b in zip(plain, ser)]
        res["sign_mismatch_vs_serial"] = [int(np.sum(np.sign(a[np.abs(b) > 1e-3]) != np.sign(b[np.abs(b) > 1e-3]))) if len(a) == len(b) else None
                                          for a, b in zip(plain, ser)]
        res["streamed_vs_serial_rms"] = [rms(a - b) if len(a) == len(b) else None for a, b in zip(strm, ser)]
        rel, sr_, _ = run(xa.Demodulator(cfg(clock_exact=3)), False)
        res["relay3_plan"] = sr_
        res["relay3_vs_serial_rms"] = [rms(a - b) if len(a) == len(b) else None for a, b in zip(rel, ser)]
        if args.oracle:
            import oracle
            od = oracle.Demod(oracle.config(mode, fs, D))
            res["vs_oracle_rms"], res["serial_vs_oracle_rms"] = [], []
            for b in range(args.bursts):
                w = od.process(bufs[b].cpu().numpy().view(np.complex64).reshape(-1))
                res["vs_oracle_rms"].append(rms(plain[b] - w) if len(w) == len(plain[b]) else None)
                res["serial_vs_oracle_rms"].append(rms(ser[b] - w) if len(w) == len(ser[b]) else None)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
