"""Turns gpurun_out/<tag>/ (scripts/collect_profiles.sh) into the committed summaries under profiles/."""
import collections
import csv
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r2"
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def last_json(path):
    lines = [l for l in open(path).read().splitlines() if l.startswith("{")]
    return json.loads(lines[-1])


def copy(rel, name):
    p = os.path.join(src, rel)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, name))
        return True
    return False


json.dump(last_json(os.path.join(src, "bench_c2.json")), open(os.path.join(dst, f"{tag}_bench_c2.json"), "w"), indent=1)
if os.path.exists(os.path.join(src, "bench_c2_noprefetch.json")):
    json.dump(last_json(os.path.join(src, "bench_c2_noprefetch.json")),
              open(os.path.join(dst, f"{tag}_bench_c2_noprefetch.json"), "w"), indent=1)
copy(os.path.join("stats", "c2_kernel_stats.csv"), f"{tag}_c2_kernel_stats.csv")
copy(os.path.join("stats_np", "c2_kernel_stats.csv"), f"{tag}_c2_kernel_stats_noprefetch.csv")
copy("timeline.txt", f"{tag}_c2_timeline.txt")
copy("timeline_streamed.txt", f"{tag}_c2_timeline_streamed.txt")
copy("small_calls.txt", f"{tag}_small_calls.txt")
copy("step_times_c2.json", f"{tag}_step_times_c2.json")
copy("parity_floor_c2.json", f"{tag}_parity_floor_c2.json")
copy("parity_floor_chain_sweep.json", f"{tag}_parity_floor_chain_sweep.json")
copy("parity_floor_passes_sweep.json", f"{tag}_parity_floor_passes_sweep.json")
for name in ("c1", "c3", "c5"):
    if os.path.exists(os.path.join(src, f"bench_{name}.json")):
        json.dump(last_json(os.path.join(src, f"bench_{name}.json")), open(os.path.join(dst, f"{tag}_bench_{name}.json"), "w"), indent=1)
    copy(os.path.join(f"stats_{name}", f"{name}_kernel_stats.csv"), f"{tag}_{name}_kernel_stats.csv")


def short(name):
    return re.sub(r"\(.*", "", name).replace("void ", "").replace("xrit::", "")


def load(path, counter):
    """per kernel: list of per-dispatch values; also the dispatch order of the run"""
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[short(r["Kernel_Name"])].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    return acc


fpath = os.path.join(src, "pmc_fetch", "c2_counter_collection.csv")
wpath = os.path.join(src, "pmc_write", "c2_counter_collection.csv")
if os.path.exists(fpath) and os.path.exists(wpath):
    f, w = load(fpath, "FETCH_SIZE"), load(wpath, "WRITE_SIZE")

    def real(vals):
        """dispatches that did something: the passes a batch enqueues beyond the last needed one return at once"""
        if not vals:
            return []
        top = max(v for _, v in vals)
        return [v for _, v in vals if v > 0.05 * top] if top > 0 else []

    STREAMED = not tag.startswith(("r1", "r2", "r3"))      # round 4 on: the counter passes run the headline's command
    WARM, STEPS = 3, 3

    def last_burst(acc):
        """rounds 1-3 (one burst at a time): sum over the dispatches of the run's LAST burst (steady state; the first one,
        cold-started, runs more passes).  Round 4 on (bursts streamed, three timed steps behind three warm-up steps): everything
        dispatched from the first timed step's decimator on -- bench.py registers it when the clock starts --, per step."""
        starts = sorted(d for k, vals in acc.items() if k.startswith("fir_decim_kernel<3, false, 0, 0") for d, _ in vals)
        if STREAMED:
            lo = starts[WARM] if len(starts) > WARM else 0
            return sum(v for k, vals in acc.items() if "synth_kernel" not in k and "read_bw" not in k for d, v in vals if d >= lo) / STEPS
        lo = starts[-1] if starts else 0
        return sum(v for k, vals in acc.items() if "synth_kernel" not in k and "read_bw" not in k for d, v in vals if d >= lo)

    rows = []
    for k in sorted(set(f) | set(w)):
        fv, wv = real(f.get(k, [])), real(w.get(k, []))
        rows.append((k, max(len(fv), len(wv)), sum(fv) / max(1, len(fv)), sum(wv) / max(1, len(wv)), 0, 0))
    tot_f, tot_w = last_burst(f), last_burst(w)
    with open(os.path.join(dst, f"{tag}_c2_pmc_hbm.csv"), "w") as fo:
        fo.write("# rocprofv3 --kernel-trace --pmc FETCH_SIZE  and  --pmc WRITE_SIZE (separate passes), "
                 + ("bench.py --steps 3 --warmup 3 (the headline's command: bursts streamed), C2 burst (256 Mi samples)\n" if STREAMED else
                    "bench.py --steps 1 --warmup 3 --no-prefetch, C2 burst (256 Mi samples): four bursts per run\n"))
        fo.write("# FETCH_SIZE/WRITE_SIZE are in KiB per dispatch, averaged over the dispatches of the run that did work (the "
                 "passes a batch enqueues beyond the last needed one return at once and are left out).\n")
        fo.write("# gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports 1/2 of the bytes of a coalesced "
                 "streaming read;\n# calibrated on kernels of known volume (costas_pass, the FIRs) and on synth_kernel "
                 "(writes 2 GiB, reports 2 GiB).\n")
        fo.write("kernel,dispatches,FETCH_SIZE_KiB_raw,WRITE_SIZE_KiB_raw,hbm_bytes_corrected_per_dispatch\n")
        for k, n, fv, wv, _, _ in rows:
            fo.write(f"{k},{n},{fv:.1f},{wv:.1f},{(2 * fv + wv) * 1024:.0f}\n")
    names = {"fir_decim": "fir_decim_kernel<3, false, 0, 0", "clock_pass": "clock_pass_kernel<1, 32, 20, false>", "clock_pass_writing": "clock_pass_kernel<1, 32, 20, true>",
             "clock_pass_jac": "clock_pass_kernel<3, 32, 20>", "costas_pass": "costas_pass_kernel<false>",
             "costas_final": "costas_pass_kernel<true>", "fir_rrc": "fir_decim_kernel<5, false, 0, 3",
             "clock_output": "clock_output_kernel<32, 20, false>", "clock_relay_pass": "clock_relay_kernel<false, true>",
             "clock_overlap": "clock_overlap_kernel<2048>"}
    d = {r[0]: r for r in rows}
    out = {}
    src_note = (f"profiles/{tag}_c2_pmc_hbm.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH_SIZE "
                "doubled per the gfx950 correction)")
    for s, full in names.items():
        full = next((k for k in d if k.startswith(full)), full)      # (template arguments behind the listed ones vary)
        if full in d:
            _, n, fv, wv, _, _ = d[full]
            out[s] = {"burst_log2": 28, "hbm_bytes_per_launch": round((2 * fv + wv) * 1024), "fetch_kib_raw": round(fv, 1),
                      "write_kib_raw": round(wv, 1), "source": src_note}
    # the whole step: every kernel of the chain (the synthetic generator is not part of it) over the LAST burst of the
    # run, i.e. the steady state (the first burst is cold-started and runs more hand-off passes)
    out["_step"] = {"burst_log2": 28, "hbm_bytes_per_step": round((2 * tot_f + tot_w) * 1024),
                    "source": src_note + (", all kernels of the chain over the run's three timed (steady-state, streamed) steps, per step"
                                          if STREAMED else ", all kernels of the chain over the run's last (steady-state) burst")}
    json.dump(out, open(os.path.join(dst, "hbm_traffic.json"), "w"), indent=1)

b = json.load(open(os.path.join(dst, f"{tag}_bench_c2.json")))
print("bench:", b["value"], b["unit"], b["ms_per_step"], "ms/step, chain roofline", b["roofline"]["frac"],
      "input kernel", b["roofline"]["dominant_kernel"])
traffic = json.load(open(os.path.join(dst, "hbm_traffic.json"))) if os.path.exists(os.path.join(dst, "hbm_traffic.json")) else {}
for k, v in b["kernels"].items():
    t = traffic.get(k, {}).get("hbm_bytes_per_launch")
    print(f"  {k:16s} {v['launches'] / b['steps']:4.1f}/step avg {v['avg_launch_ms']:.4f} ms  own-bytes GB/s {v.get('achieved_gbs')} "
          f"frac {v.get('hbm_frac')}  pmc bytes {t}")
print("step traffic:", traffic.get("_step"))
print("parity:", b.get("parity_vs_oracle"))
print("cpu:", b.get("cpu_baseline"))
