"""How the parity floor of the clock recovery moves with the distance of the device chain's Costas output from the oracle's
(VERDICT round 3, item 3).  The clock recovery runs on the device as ONE exact trajectory (cfg.clock_exact = 1 on the stage
object: bit-identical to the serial float32 recurrence) on inputs of controlled quality,
    y_a = y_oracle + a (y_device - y_oracle),  a = 0 .. 2,
i.e. the device front end's own error pattern scaled -- a = 0 is the oracle's Costas output itself (floor 0: the stage is
the oracle's recurrence bit for bit), a = 1 the shipped chain -- for a steady-state burst (the second of two consecutive
bursts) of C2 and of C3.  With a library built with -DXRIT_EXPERIMENTS the knobs that move the device's Costas output are
measured too (spare Costas pass kept, stop rule x 0.1; -DXRIT_ACCURATE_SINCOS: the math library's sincosf in the loop).
    python scripts/r4_floor_vs_frontend.py [--out gpurun_out/r4_floor_vs_frontend.json]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import xritdemod_amd as xa
from xritdemod_amd import _capi
import oracle


def rms(a):
    return float(np.sqrt(np.mean(np.abs(a) ** 2)))


def bursts_of(mode, fs_in, sym_rate, alpha, n, count):
    dev = torch.device("cuda", 0)
    sp = _capi.synth_params(fs_in=fs_in, symbol_rate=sym_rate, alpha=alpha)
    buf = torch.empty((n, 2), dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    out = []
    for b in range(count):
        _capi.synth_generate_device(sp, b * n, n, buf.data_ptr(), device=0, stream=st)
        torch.cuda.synchronize(dev)
        out.append(buf.cpu().numpy().view(np.complex64).reshape(-1).copy())
    return out


def chain_stages(mode, fs_in, D, xs, env=None):
    old = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        dem = xa.Demodulator(xa.Demodulator.config(mode, fs_in, D, clock_exact=1))
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    dem.keep_stages(True)
    res = []
    for x in xs:
        soft = dem.process(x)
        res.append((dem.stage("costas").copy(), soft.copy()))
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/r4_floor_vs_frontend.json")
    ap.add_argument("--log2-c2", type=int, default=27)
    ap.add_argument("--log2-c3", type=int, default=25)
    args = ap.parse_args()
    report = {"what": __doc__.split("\n    python")[0], "experiments_build": xa.build_experiments(), "configs": {}}
    for name, mode, fs_in, D, sym_rate, alpha, log2 in (("C2", "lrit", 6.25e6, 5, 293883.0, 0.5, args.log2_c2),
                                                        ("C3", "hrit", 2.5e6, 1, 927000.0, 0.3, args.log2_c3)):
        n = 1 << log2
        n -= n % D
        xs = bursts_of(mode, fs_in, sym_rate, alpha, n, 2)
        od = oracle.Demod(oracle.config(mode, fs_in, D))
        ref = []
        for x in xs:
            soft = od.process(x)
            ref.append((od.stage("costas").copy(), soft.copy()))
        dev = chain_stages(mode, fs_in, D, xs)
        yo, so = ref[1]
        yg, sg = dev[1]
        cfg = {"samples_per_burst": n, "symbols_compared": int(len(so)), "costas_stage_rms_device_vs_oracle": rms(yg - yo),
               "chain_exact_closure_soft_rms_vs_oracle": rms(sg - so) if len(sg) == len(so) else None, "curve": []}
        args_clk = (od.sps, 0.0037 ** 2 / 4, 0.5, 0.0037, 0.005)
        for a in (0.0, 0.05, 0.1, 0.2, 0.35, 0.5, 0.75, 1.0, 1.5, 2.0):
            clk = xa.ClockRecovery(*args_clk, exact=1)
            y0 = (ref[0][0] + np.float32(a) * (dev[0][0] - ref[0][0])).astype(np.complex64)
            y1 = (yo + np.float32(a) * (yg - yo)).astype(np.complex64)
            clk.Work(y0)
            s1 = clk.Work(y1).real
            row = {"a": a, "costas_stage_rms": rms(y1 - yo)}
            if len(s1) == len(so):
                row["soft_rms_vs_oracle"] = rms(s1 - so)
                big = np.abs(so) > 1e-3
                row["sign_mismatches"] = int((np.sign(s1[big]) != np.sign(so[big])).sum())
            else:
                row["symbol_count"] = [len(s1), len(so)]
            cfg["curve"].append(row)
            print(name, json.dumps(row), flush=True)
        knobs = []
        if xa.build_experiments():
            for label, env in (("shipped", {}), ("spare Costas pass kept", {"XRIT_KEEP_SPARE": "1"}),
                               ("stop rule x 0.1 (1e-6 rad)", {"XRIT_COSTAS_TOL": "0.1"}),
                               ("stop rule x 0.1, spare pass kept", {"XRIT_COSTAS_TOL": "0.1", "XRIT_KEEP_SPARE": "1"})):
                st = chain_stages(mode, fs_in, D, xs, env)
                yk, sk = st[1]
                row = {"knob": label, "costas_stage_rms": rms(yk - yo),
                       "soft_rms_vs_oracle": rms(sk - so) if len(sk) == len(so) else None}
                knobs.append(row)
                print(name, json.dumps(row), flush=True)
        cfg["knobs"] = knobs
        report["configs"][name] = cfg
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    json.dump(report, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
