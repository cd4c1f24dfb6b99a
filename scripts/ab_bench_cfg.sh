#!/bin/bash
# A/B on one box: bench.py of another configuration under environment switches; usage: ab_bench_cfg.sh "<bench args>" "ENV=.." ...
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
ARGS="$1"; shift
for cfg in "$@"; do
  env $cfg python bench.py $ARGS --steps 8 --warmup 3 --no-cpu --no-exact 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernels']; p = d.get('parity_vs_oracle') or {}
print('$cfg', 'ms_per_step', d['ms_per_step'], 'Gs/s', round(d['value']/1e3,1), 'relay', k['clock_relay']['avg_launch_ms'], d['loop_passes']['clock_relay'], d['config'].get('clock_recovery','')[:90], 'parity', p.get('rms'), (p.get('steady_state') or {}).get('rms'), (p.get('steady_state') or {}).get('vs_serial_gpu_rms'))
"
done
