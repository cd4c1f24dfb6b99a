#!/bin/bash
# Runs on the GPU box: the bench under each environment setting given as an argument ("-" = none), interleaved
# twice on the same box (boxes differ by several per cent).  Usage: scripts/ab_env.sh - XRIT_NO_STATIC_MF=1 [-- bench args]
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
vars=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do vars+=("$1"); shift; done
[ "$1" = "--" ] && shift
for rep in 1 2; do
  for v in "${vars[@]}"; do
    if [ "$v" = "-" ]; then e=(); else e=($v); fi
    env "${e[@]}" python bench.py --steps 20 --warmup 4 --no-cpu --no-serial-floor "$@" 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=j['kernels']
print('$v', j['value'], j['ms_per_step'], j['loop_passes']['clock'], j['parity_vs_oracle']['rms'] if j.get('parity_vs_oracle') else None, {n:round(k[n]['avg_launch_ms'],4) for n in k})
"
  done
done
