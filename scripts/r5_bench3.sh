#!/bin/bash
# bench.py at the BASELINE configurations (driver's command), short summaries.  Usage: scripts/r5_bench3.sh TAG [cases...] [-- extra bench args]
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
mkdir -p $R/gpurun_out/$TAG
CASES=""
while [ $# -gt 0 ] && [ "$1" != "--" ]; do CASES="$CASES $1"; shift; done
[ "$1" == "--" ] && shift
[ -z "$CASES" ] && CASES="c2"
for c in $CASES; do
  case $c in
    c2) A="";;
    c5) A="--decimation 32";;
    c1) A="--decimation 1";;
    c3) A="--decimation 1 --mode hrit";;
  esac
  python $R/bench.py --steps 20 --warmup 5 --no-exact --no-cpu --no-serial-floor $A "$@" > $R/gpurun_out/$TAG/bench_$c.json 2> $R/gpurun_out/$TAG/bench_$c.err
  python - <<PY
import json
try:
    d=json.loads(open("$R/gpurun_out/$TAG/bench_$c.json").read().strip().splitlines()[-1])
    print("$c", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["loop_passes"]["clock_relay_segments"], {k:v.get("avg_launch_ms") for k,v in d.get("kernels",{}).items() if k in ("fir_decim","fir_rrc","clock_overlap","clock_relay","costas_final","costas_pass")})
except Exception as e:
    print("$c failed", e); print(open("$R/gpurun_out/$TAG/bench_$c.err").read()[-1500:])
PY
done
