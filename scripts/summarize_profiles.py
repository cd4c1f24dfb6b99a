"""Turns gpurun_out/<tag>/ (scripts/collect_profiles.sh) into the committed summaries under profiles/."""
import collections
import csv
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

shutil.copy(os.path.join(src, "bench_c2.json"), os.path.join(dst, f"{tag}_bench_c2.json"))
shutil.copy(os.path.join(src, "stats", "c2_kernel_stats.csv"), os.path.join(dst, f"{tag}_c2_kernel_stats.csv"))


def short(name):
    return re.sub(r"\(.*", "", name).replace("void ", "").replace("xrit::", "")


def load(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return acc


f = load(os.path.join(src, "pmc_fetch", "c2_counter_collection.csv"), "FETCH_SIZE")
w = load(os.path.join(src, "pmc_write", "c2_counter_collection.csv"), "WRITE_SIZE")
rows = []
for k in sorted(set(f) | set(w)):
    fv, wv = f.get(k, []), w.get(k, [])
    rows.append((k, len(fv), sum(fv) / max(1, len(fv)), sum(wv) / max(1, len(wv))))
with open(os.path.join(dst, f"{tag}_c2_pmc_hbm.csv"), "w") as fo:
    fo.write("# rocprofv3 --kernel-trace --pmc FETCH_SIZE  and  --pmc WRITE_SIZE (separate passes), "
             "bench.py --steps 1 --warmup 1, C2 burst (256 Mi samples)\n")
    fo.write("# FETCH_SIZE/WRITE_SIZE are in KiB per dispatch (averaged over the dispatches of the run).\n")
    fo.write("# gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports 1/2 of the bytes of a coalesced "
             "streaming read;\n# calibrated here on costas_pass/fir kernels of known volume and synth_kernel "
             "(writes 2 GiB, reports 2 GiB).\n")
    fo.write("kernel,dispatches,FETCH_SIZE_KiB_raw,WRITE_SIZE_KiB_raw,hbm_bytes_corrected\n")
    for k, n, fv, wv in rows:
        fo.write(f"{k},{n},{fv:.1f},{wv:.1f},{(2 * fv + wv) * 1024:.0f}\n")
names = {"fir_decim": "fir_decim_kernel<3, false, 0, 0>", "clock_pass": "clock_pass_kernel<1, 32, 20>",
         "clock_pass_jac": "clock_pass_kernel<3, 32, 20>", "costas_pass": "costas_pass_kernel<false>",
         "costas_final": "costas_pass_kernel<true>", "fir_rrc": "fir_decim_kernel<5, false, 0, 3>",
         "agc_apply": "agc_apply_runs_kernel<3>", "clock_output": "clock_output_kernel<32, 20>"}
d = {r[0]: r for r in rows}
out = {}
for s, full in names.items():
    if full in d:
        _, n, fv, wv = d[full]
        out[s] = {"burst_log2": 28, "hbm_bytes_per_launch": round((2 * fv + wv) * 1024), "fetch_kib_raw": round(fv, 1),
                  "write_kib_raw": round(wv, 1),
                  "source": f"profiles/{tag}_c2_pmc_hbm.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, FETCH_SIZE doubled per "
                            "the gfx950 correction)"}
json.dump(out, open(os.path.join(dst, "hbm_traffic.json"), "w"), indent=1)
b = json.load(open(os.path.join(dst, f"{tag}_bench_c2.json")))
print("bench:", b["value"], b["unit"], b["ms_per_step"], "ms/step", "roofline", b["roofline"]["frac"], "chain", b["roofline"]["chain_frac"])
for k, v in b["kernels"].items():
    t = out.get(k, {}).get("hbm_bytes_per_launch")
    print(f"  {k:16s} {v['launches']/b['steps']:4.1f}/step avg {v['avg_launch_ms']:.4f} ms  own-bytes GB/s {v.get('achieved_gbs')} "
          f"frac {v.get('hbm_frac')}  pmc bytes {t}")
