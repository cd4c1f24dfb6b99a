#!/bin/bash
# round 6: bench.py with cfg.front_exact = $1 on C2, C5, C1, C3; one JSON line per configuration under gpurun_out/$2/
# usage: scripts/r6_bench_modes.sh <front_exact> <outdir> [extra bench.py flags]
FE=${1:-2}; OUT=gpurun_out/${2:-r6_fe}; shift; shift
mkdir -p $OUT
run() { name=$1; shift; timeout 900 python bench.py --front-exact $FE --no-exact "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err || tail -5 $OUT/bench_$name.err; }
run c2 --decimation 5 --steps 10 --warmup 4 "$@"
run c5 --decimation 32 --steps 10 --warmup 4 "$@"
run c1 --decimation 1 --steps 6 --warmup 3 "$@"
run c3 --mode hrit --decimation 1 --steps 6 --warmup 3 "$@"
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/bench_c*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "no line", e); continue
    p = d.get("parity_vs_oracle") or {}
    print(f.split("/")[-1], "ms/step", d["ms_per_step"], "Gsps", round(d["value"] / 1e3, 1), "roofline", d["roofline"]["frac"])
    print("   parity:", {k: v for k, v in p.items() if not isinstance(v, dict)})
    for k, v in p.items():
        if isinstance(v, dict): print("   ", k, v)
    for key in ("kernels", "kernel_ms", "per_kernel"):
        if key in d: print("   ", key, json.dumps(d[key])[:2500])
PY
