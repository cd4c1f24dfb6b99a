"""What an exact front end would buy (VERDICT round 4, item 3d -- measured with the stage objects instead of built into the chain).
The chain's stages run on the device one by one (xrit_fir_*, xrit_costas_*, xrit_clock_* with clock_exact = 1: the serial float32
trajectory) and the oracle's own stage outputs are substituted for the device's up to a chosen point:
    shipped              device decimator -> AGC -> matched filter -> Costas -> clock            (the chain as it ships)
    exact AGC            oracle's AGC output -> device matched filter -> Costas -> clock          (AGC walked literally, bit-exact)
    exact AGC, tight     ... and the Costas hand-off's stop rule x 0.1 (a second pass over the samples; -DXRIT_EXPERIMENTS)
    exact RRC            oracle's matched-filter output -> device Costas -> clock                  (FIRs in the oracle's order too)
    exact RRC, tight     ... and the tight stop rule: what is left is the Costas loop's own arithmetic
for the second of two consecutive bursts of C2 and C3: distance of the Costas stage's output and of the soft symbols from the oracle's.
    python scripts/r5_front_exact_estimate.py [--out gpurun_out/r5_front_exact_estimate.json]"""
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


def bursts_of(fs_in, sym_rate, alpha, n, count):
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


class env_set:
    def __init__(self, env):
        self.env = env or {}

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.env}
        os.environ.update(self.env)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/r5_front_exact_estimate.json")
    ap.add_argument("--log2-c2", type=int, default=27)
    ap.add_argument("--log2-c3", type=int, default=25)
    args = ap.parse_args()
    report = {"what": __doc__.split("\n    python")[0], "experiments_build": xa.build_experiments(), "configs": {}}
    tight = {"XRIT_COSTAS_TOL": "0.1"}
    for name, mode, fs_in, D, sym_rate, alpha, log2 in (("C2", "lrit", 6.25e6, 5, 293883.0, 0.5, args.log2_c2),
                                                        ("C3", "hrit", 2.5e6, 1, 927000.0, 0.3, args.log2_c3)):
        n = 1 << log2
        n -= n % D
        xs = bursts_of(fs_in, sym_rate, alpha, n, 2)
        ocfg = oracle.config(mode, fs_in, D)
        od = oracle.Demod(ocfg)
        ref = []
        for x in xs:
            soft = od.process(x)
            ref.append({k: od.stage(k).copy() for k in ("agc", "rrc", "costas")} | {"soft": soft.copy()})
        cfg = xa.Demodulator.config(mode, fs_in, D)
        rrc_taps = od.rrc_taps()
        so, yo, ro = ref[1]["soft"], ref[1]["costas"], ref[1]["rrc"]
        rows = []

        def clock():
            return xa.ClockRecovery(od.sps, cfg.clock_gain_omega, cfg.clock_mu, cfg.clock_alpha, cfg.clock_omega_limit, exact=1)

        def finish(label, rrc_out, env):
            # rrc_out: the matched filter's output of the two bursts (device or oracle)
            with env_set(env):
                cl = xa.CostasLoop(cfg.pll_alpha)
            clk = clock()
            y = s = None
            for r in rrc_out:
                y = cl.Work(r)
                s = clk.Work(y).real
            row = {"variant": label, "rrc_stage_rms": rms(rrc_out[1] - ro), "costas_stage_rms": rms(y - yo)}
            if len(s) == len(so):
                row["soft_rms_vs_oracle"] = rms(s - so)
                big = np.abs(so) > 1e-3
                row["sign_mismatches"] = int((np.sign(s[big]) != np.sign(so[big])).sum())
            else:
                row["symbol_count"] = [len(s), len(so)]
            rows.append(row)
            print(name, json.dumps(row), flush=True)

        # the chain as it ships, to closure
        dem = xa.Demodulator(xa.Demodulator.config(mode, fs_in, D, clock_exact=1))
        dem.keep_stages(True)
        for x in xs:
            sg = dem.process(x)
        row = {"variant": "shipped chain (clock to closure)", "rrc_stage_rms": rms(dem.stage("rrc") - ro),
               "costas_stage_rms": rms(dem.stage("costas") - yo), "soft_rms_vs_oracle": rms(sg - so) if len(sg) == len(so) else None}
        rows.append(row)
        print(name, json.dumps(row), flush=True)
        # exact AGC: the oracle's AGC output through the device's matched filter
        f = xa.FirFilter(1, rrc_taps)
        dev_rrc = [f.Work(r["agc"], len(r["agc"])) for r in ref]
        finish("exact AGC", dev_rrc, {})
        if xa.build_experiments():
            finish("exact AGC, tight Costas stop rule", dev_rrc, tight)
        finish("exact matched-filter output", [r["rrc"] for r in ref], {})
        if xa.build_experiments():
            finish("exact matched-filter output, tight Costas stop rule", [r["rrc"] for r in ref], tight)
        report["configs"][name] = {"samples_per_burst": n, "symbols_compared": int(len(so)), "rows": rows}
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    json.dump(report, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
