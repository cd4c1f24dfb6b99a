#!/bin/bash
# Runs on the GPU box (via gpurun): bench JSON, rocprofv3 kernel statistics and the two PMC passes for HBM traffic.
# Usage: scripts/collect_profiles.sh <round-tag>     (outputs under gpurun_out/<tag>/)
set -u
TAG=${1:-r1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python bench.py > $OUT/bench_c2.json 2> $OUT/bench_c2.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o c2 -- python $R/bench.py --steps 3 --warmup 2 --no-cpu --no-profile > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o c2 -- python $R/bench.py --steps 1 --warmup 1 --no-cpu --no-profile > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o c2 -- python $R/bench.py --steps 1 --warmup 1 --no-cpu --no-profile > $OUT/pmc_write.log 2>&1
ls -R $OUT | head -30
