#!/bin/bash
# Runs on the GPU box: bench.py with each library variant under xritdemod_amd/lib/ab/ in turn, interleaved twice (same box).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
cp xritdemod_amd/lib/libxritdemod_amd.so /tmp/lib_orig.so
for rep in 1 2; do
for v in xritdemod_amd/lib/ab/*.so; do
  cp $v xritdemod_amd/lib/libxritdemod_amd.so
  python bench.py --steps ${STEPS:-40} --warmup 5 --no-cpu --no-serial-floor --no-exact --no-profile "$@" 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$(basename $v)', j['value'], j['ms_per_step'])
"
done
done
cp /tmp/lib_orig.so xritdemod_amd/lib/libxritdemod_amd.so
