#!/bin/bash
# A/B on one box: the streamed C2 bench under environment switches (experiments build); one line per configuration
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for cfg in "$@"; do
  env $cfg python bench.py $XARGS --steps 20 --warmup 5 --no-cpu --no-exact --no-serial-floor 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernels']
print('$cfg', 'ms_per_step', d['ms_per_step'], 'relay', k['clock_relay']['avg_launch_ms'], 'dec', k['fir_decim']['avg_launch_ms'], 'costas_guess', k['costas_guess']['avg_launch_ms'], 'clock_guess', k['clock_guess']['avg_launch_ms'], 'parity', (d.get('parity_vs_oracle') or {}).get('rms'), (d.get('parity_vs_oracle') or {}).get('steady_state'))
"
done
