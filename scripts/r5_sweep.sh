#!/bin/bash
# A/B of the overlap plan's knobs on one box: scripts/r5_sweep.sh TAG "ENV1" "ENV2" ... (each an env assignment string, "" = default)
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
mkdir -p $R/gpurun_out/$TAG
CASE_ARGS=${CASE_ARGS:-}
i=0
for e in "$@"; do
  i=$((i+1))
  env $e python $R/bench.py --steps ${STEPS:-20} --warmup 5 --no-exact --no-cpu --no-serial-floor --no-profile $CASE_ARGS > $R/gpurun_out/$TAG/s$i.json 2> $R/gpurun_out/$TAG/s$i.err
  python - <<PY
import json
try:
    d=json.loads(open("$R/gpurun_out/$TAG/s$i.json").read().strip().splitlines()[-1])
    print("[$e]", d["value"], d["ms_per_step"], d["roofline"]["frac"] if "roofline" in d else "", d["loop_passes"]["clock_relay_segments"])
except Exception as ex:
    print("[$e] failed", ex); print(open("$R/gpurun_out/$TAG/s$i.err").read()[-800:])
PY
done
