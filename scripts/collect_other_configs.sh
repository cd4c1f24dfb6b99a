#!/bin/bash
# Runs on the GPU box: bench JSON and rocprofv3 kernel statistics for C5 (40 Msps, d=32), C3 (HRIT) and C1's chain.
set -u
TAG=${1:-r2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in "c5:--decimation 32" "c3:--mode hrit --decimation 1" "c1:--decimation 1"; do
  name=${cfg%%:*}; args=${cfg#*:}
  python $R/bench.py $args > $OUT/bench_$name.json 2> /dev/null
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$name -o $name -- python $R/bench.py --steps 3 --warmup 2 --no-cpu --no-profile --no-exact $args > /dev/null 2>&1
done
ls $OUT | head -30
