"""Clock-recovery configurations side by side on steady-state bursts of one stream: ms per burst (one burst at a time), rms of the
soft symbols against the serial device trajectory (cfg.clock_serial) and, with --oracle, against the CPU oracle.
Every row is "ENV=VALUE,... key=value ..." -- environment switches of the library (read when the handle is created) and
fields of xrit_demod_config.
    python scripts/r4_modes.py [--log2 28] [--mode lrit] [--decimation 5] [--bursts 3] --row "clock_exact=3" --row "XRIT_NO_HANDOFF=1 clock_exact=4" ..."""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2", type=int, default=28)
    ap.add_argument("--bursts", type=int, default=3)
    ap.add_argument("--mode", default="lrit")
    ap.add_argument("--decimation", type=int, default=5)
    ap.add_argument("--esn0", type=float, default=None)
    ap.add_argument("--oracle", action="store_true")
    ap.add_argument("--row", action="append", default=[])
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    n, D = 1 << args.log2, args.decimation
    fs_in = (1.25e6 if args.mode == "lrit" else 2.5e6) * D
    sym_rate, alpha = (293883.0, 0.5) if args.mode == "lrit" else (927000.0, 0.3)
    sp = _capi.synth_params(fs_in=fs_in, symbol_rate=sym_rate, alpha=alpha, **({"esn0_db": args.esn0} if args.esn0 is not None else {}))
    stream = torch.cuda.current_stream(dev)
    bursts = torch.empty((args.bursts, n, 2), dtype=torch.float32, device=dev)
    for b in range(args.bursts):
        _capi.synth_generate_device(sp, b * n, n, bursts[b].data_ptr(), device=0, stream=stream.cuda_stream)
    torch.cuda.synchronize(dev)
    want = None
    if args.oracle:
        import oracle
        od = oracle.Demod(oracle.config(args.mode, fs_in, D))
        want = [od.process(bursts[b].cpu().numpy().view(np.complex64).reshape(-1)) for b in range(args.bursts)]

    def run(spec):
        env, kw = {}, {}
        for tok in spec.split():
            k, v = tok.split("=", 1)
            if k.isupper():
                env[k] = v
            else:
                kw[k] = int(v)
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            dem = xa.Demodulator(xa.Demodulator.config(args.mode, fs_in, D, **kw))
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
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

    ser, _, _ = run("clock_serial=1")
    rows = []
    steady = range(1, args.bursts)
    s = np.concatenate([ser[b] for b in steady])
    if want is not None:
        w = np.concatenate([want[b] for b in steady])
        print(f"serial device vs oracle (steady-state bursts): rms {np.sqrt(np.mean((s - w) ** 2)):.3e}", flush=True)
    for spec in args.row:
        outs, ms, stats = run(spec)
        g = np.concatenate([outs[b] for b in steady])
        r = {"row": spec, "ms_per_burst": round(float(np.mean([ms[b] for b in steady])), 3),
             "clock_passes": [int(stats[b].clock_passes) for b in steady],
             "relay_passes": [int(stats[b].clock_relay_passes) for b in steady],
             "relay_segments": int(stats[-1].clock_relay_segments)}
        if len(g) == len(s):
            r["rms_vs_serial"] = float(np.sqrt(np.mean((g - s) ** 2)))
            r["words_differing"] = int((g.view(np.uint32) != s.view(np.uint32)).sum())
            big = np.abs(s) > 1e-3
            r["sign_mismatches_vs_serial"] = int((np.sign(g[big]) != np.sign(s[big])).sum())
        else:
            r["symbol_count_differs"] = [len(g), len(s)]
        if want is not None and len(w) == len(g):
            r["rms_vs_oracle"] = float(np.sqrt(np.mean((g - w) ** 2)))
        rows.append(r)
        print(json.dumps(r), flush=True)
    if args.out:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        json.dump({"workload": f"{args.mode} d={D}, {args.bursts} bursts of 2^{args.log2} samples; steady-state bursts compared", "rows": rows},
                  open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
