"""Segments per CU x relay passes on the C2 burst (XRIT_RELAY_PER_CU, read when a handle is created): time per steady-state burst,
one burst at a time, and distance from the serial device trajectory.    python scripts/relay_per_cu.py [--log2 28]"""
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
    ap.add_argument("--bursts", type=int, default=4)
    ap.add_argument("--grid", default="3x3,2x2,2x3,1x2,4x4,4x3")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    n, D, fs_in = 1 << args.log2, 5, 6.25e6
    sp = _capi.synth_params(fs_in=fs_in)
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
    s = np.concatenate(ser[2:])
    for cell in args.grid.split(","):
        per_cu, passes = (int(v) for v in cell.split("x"))
        os.environ["XRIT_RELAY_PER_CU"] = str(per_cu)
        o, ms, st = run(clock_exact=passes)
        g = np.concatenate(o[2:])
        r = {"per_cu": per_cu, "passes": passes, "segments": int(st.clock_relay_segments), "ms_per_burst": round(float(np.mean(ms[2:])), 3)}
        if len(g) == len(s):
            r["rms_vs_serial_device"] = float(np.sqrt(np.mean((g - s) ** 2)))
            r["words_differing"] = int((g.view(np.uint32) != s.view(np.uint32)).sum())
        print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
