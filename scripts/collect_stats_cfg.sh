#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel statistics of the other configurations' streamed bench (C1, C3, C5), 8 steps each.
# Outputs under gpurun_out/<tag>/stats_<cfg>/; scripts/summarize_profiles.py <tag> copies them to profiles/<tag>_<cfg>_kernel_stats.csv
set -u
TAG=${1:-r4}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in "c5:--decimation 32" "c3:--mode hrit --decimation 1" "c1:--decimation 1"; do
  name=${cfg%%:*}; args=${cfg#*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$name -o $name -- python $R/bench.py $args --steps 8 --warmup 3 --no-cpu --no-profile --no-exact > $OUT/stats_$name.log 2>&1
done
find $OUT -name "*_kernel_trace.csv" -delete
find $OUT -name "*_agent_info.csv" -delete
ls $OUT/stats_c1 $OUT/stats_c3 $OUT/stats_c5
