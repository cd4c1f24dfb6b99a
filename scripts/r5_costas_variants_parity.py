"""Steady-state parity of the chain against the oracle for the Costas stage's late round-5 variants, on consecutive bursts of one
stream (bursts 1.. of N; C2, C5: 2^28 samples, C1, C3: 2^26): the default, two chains of warm-up in the final pass (XRIT_COSTAS_FINAL_WARM,
library built with -DXRIT_EXPERIMENTS) and cfg.front_exact = 1 (four).
    python scripts/r5_costas_variants_parity.py [--bursts 6] [--out gpurun_out/r5_costas_variants_parity.json]"""
import argparse, json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import xritdemod_amd as xa
from xritdemod_amd import _capi
import oracle

def rms(a): return float(np.sqrt(np.mean(np.abs(a) ** 2)))

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bursts", type=int, default=6)
    ap.add_argument("--out", default="gpurun_out/r5_costas_variants_parity.json")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream(dev)
    rep = {"what": __doc__.split("\n    python")[0], "configs": {}}
    for name, mode, fs_in, D, sym, alpha, log2 in (("C2", "lrit", 6.25e6, 5, 293883.0, 0.5, 28), ("C3", "hrit", 2.5e6, 1, 927000.0, 0.3, 26),
                                                    ("C1", "lrit", 1.25e6, 1, 293883.0, 0.5, 26), ("C5", "lrit", 40e6, 32, 293883.0, 0.5, 28)):
        n = 1 << log2; n -= n % D
        sp = _capi.synth_params(fs_in=fs_in, symbol_rate=sym, alpha=alpha)
        buf = torch.empty((n, 2), dtype=torch.float32, device=dev)
        od = oracle.Demod(oracle.config(mode, fs_in, D))
        xs, want = [], []
        for b in range(args.bursts):
            _capi.synth_generate_device(sp, b * n, n, buf.data_ptr(), device=0, stream=st.cuda_stream)
            torch.cuda.synchronize(dev)
            x = buf.cpu().numpy().view(np.complex64).reshape(-1).copy()
            xs.append(x); want.append(od.process(x))
        rows = []
        for label, env, fe in (("default", {}, 0), ("final pass: 2 chains of warm-up (XRIT_COSTAS_FINAL_WARM=2, -DXRIT_EXPERIMENTS)", {"XRIT_COSTAS_FINAL_WARM": "2"}, 0),
                               ("cfg.front_exact = 1 (4 chains of warm-up)", {}, 1)):
            if env and not xa.build_experiments(): continue
            os.environ.pop("XRIT_COSTAS_FINAL_WARM", None)
            os.environ.update(env)
            dem = xa.Demodulator(xa.Demodulator.config(mode, fs_in, D, front_exact=fe))
            per, ok = [], True
            for b in range(args.bursts):
                g = dem.process(xs[b])
                if len(g) != len(want[b]): ok = False; per.append(None); continue
                per.append(rms(g - want[b]))
            tail = [p for p in per[1:] if p is not None]
            row = {"variant": label, "burst0": per[0], "steady_bursts": per[1:], "steady_rms": float(np.sqrt(np.mean(np.square(tail)))) if tail else None, "counts_equal": ok}
            rows.append(row); print(name, json.dumps(row), flush=True)
        rep["configs"][name] = {"samples_per_burst": n, "rows": rows}
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    json.dump(rep, open(args.out, "w"), indent=1)

if __name__ == "__main__":
    main()
