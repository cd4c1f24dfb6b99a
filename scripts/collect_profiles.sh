#!/bin/bash
# Runs on the GPU box (via gpurun): bench JSON, rocprofv3 kernel statistics and the two PMC passes for HBM traffic.
# Usage: scripts/collect_profiles.sh <round-tag>     (outputs under gpurun_out/<tag>/)
set -u
TAG=${1:-r2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python bench.py > $OUT/bench_c2.json 2> $OUT/bench_c2.err
python bench.py --no-prefetch --no-cpu --no-exact > $OUT/bench_c2_noprefetch.json 2> /dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o c2 -- python $R/bench.py --steps 3 --warmup 2 --no-cpu --no-profile --no-exact > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_np -o c2 -- python $R/bench.py --steps 3 --warmup 2 --no-cpu --no-profile --no-exact --no-prefetch > $OUT/stats_np.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o c2 -- python $R/bench.py --steps 1 --warmup 3 --no-cpu --no-profile --no-exact --no-prefetch > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o c2 -- python $R/bench.py --steps 1 --warmup 3 --no-cpu --no-profile --no-exact --no-prefetch > $OUT/pmc_write.log 2>&1
cd $R
scripts/timeline.sh $TAG --no-prefetch > /dev/null 2>&1
python scripts/timeline_print.py gpurun_out/tl_$TAG/t_kernel_trace.csv > $OUT/timeline.txt 2>&1
python scripts/small_call_latency.py 2>&1 | grep "^D=" > $OUT/small_calls.txt
python scripts/parity_floor.py --burst-log2 28 --chains 0,256 --out $OUT/parity_floor_c2.json > /dev/null 2>&1
python scripts/parity_floor.py --burst-log2 25 --chains 0,64,112,192,256,512 --out $OUT/parity_floor_chain_sweep.json > /dev/null 2>&1
python scripts/parity_floor.py --burst-log2 27 --chains 0 --passes 2,3,4,5,6,7,8,10,14 --out $OUT/parity_floor_passes_sweep.json > /dev/null 2>&1
ls -R $OUT | head -40
