#!/bin/bash
# round 6: bench.py --front-exact 2 (C2 unless other flags are given) under a list of environment settings, 10 steps each;
# prints ms per step and the exact kernels' times.   usage: scripts/r6_ab_env.sh "A=1" "B=2 C=3" ...  ("-" = no setting)
mkdir -p gpurun_out/r6_ab
i=0
for e in "$@"; do
  i=$((i+1))
  envs=""; [ "$e" != "-" ] && envs="$e"
  env $envs timeout 600 python bench.py --front-exact 2 --no-exact --no-cpu --no-serial-floor --steps 10 --warmup 4 $BENCH_FLAGS > gpurun_out/r6_ab/b$i.json 2> gpurun_out/r6_ab/b$i.err || tail -3 gpurun_out/r6_ab/b$i.err
  python - "$e" gpurun_out/r6_ab/b$i.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    k = d.get("kernels", {})
    pick = {n: round(v["avg_launch_ms"], 3) for n, v in k.items() if n in ("fir_decim", "agc_exact", "fir_rrc", "costas_exact", "costas_exact_fix", "clock_overlap")}
    print("%-40s ms/step %.3f  %s" % (sys.argv[1], d["ms_per_step"], pick))
except Exception as ex:
    print(sys.argv[1], "failed", ex)
PY
done
