#!/bin/bash
# Runs on the GPU box: the bench with each library variant under xritdemod_amd/lib/ab/ in turn (same box, same
# session -- boxes differ by several per cent).  Prints value / ms_per_step and the loop kernels' average times.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
cp xritdemod_amd/lib/libxritdemod_amd.so /tmp/lib_orig.so
for v in xritdemod_amd/lib/ab/*.so; do
  cp $v xritdemod_amd/lib/libxritdemod_amd.so
  for rep in 1 2; do
  python bench.py --steps 20 --warmup 4 --no-cpu --no-serial-floor "$@" 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=j['kernels']
print('$(basename $v)', j['value'], j['ms_per_step'], j['loop_passes']['clock'], {n:round(k[n]['avg_launch_ms'],4) for n in k})
"
  done
done
cp /tmp/lib_orig.so xritdemod_amd/lib/libxritdemod_amd.so
