#!/bin/bash
# Runs on the GPU box: only the two PMC passes of collect_profiles.sh (HBM bytes per kernel), into gpurun_out/<tag>/
TAG=${1:-r3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/pmc_fetch $OUT/pmc_write
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o c2 -- python $R/bench.py --steps 1 --warmup 3 --no-cpu --no-profile --no-exact --no-prefetch > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o c2 -- python $R/bench.py --steps 1 --warmup 3 --no-cpu --no-profile --no-exact --no-prefetch > $OUT/pmc_write.log 2>&1
ls $OUT/pmc_fetch $OUT/pmc_write
