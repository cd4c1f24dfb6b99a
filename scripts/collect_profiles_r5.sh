#!/bin/bash
# Runs on the GPU box (via gpurun): round 5's profile set, every pass with the headline's command (bursts streamed: front end,
# Costas loop and clock-recovery walkers of the next two bursts run ahead of their calls).  Outputs under gpurun_out/<tag>/;
# scripts/summarize_profiles.py <tag> turns them into profiles/<tag>_*.
set -u
TAG=${1:-r5}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd $R
python bench.py --steps 20 --warmup 5 > $OUT/bench_c2.json 2> $OUT/bench_c2.err      # (the driver's command)
python bench.py --no-prefetch --no-cpu --no-exact > $OUT/bench_c2_noprefetch.json 2> /dev/null
cd /tmp && export TMPDIR=/tmp
# kernel statistics over 20 streamed steps (beside), and one burst at a time (alone)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o c2 -- python $R/bench.py --steps 20 --warmup 5 --no-cpu --no-profile --no-exact > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_np -o c2 -- python $R/bench.py --steps 20 --warmup 5 --no-cpu --no-profile --no-exact --no-prefetch > $OUT/stats_np.log 2>&1
# HBM traffic: separate counter passes, the headline's command, three timed steps behind three warm-up steps
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o c2 -- python $R/bench.py --steps 3 --warmup 3 --no-cpu --no-profile --no-exact > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o c2 -- python $R/bench.py --steps 3 --warmup 3 --no-cpu --no-profile --no-exact > $OUT/pmc_write.log 2>&1
cd $R
bash scripts/r5_timeline.sh > /dev/null 2>&1
cp gpurun_out/r5_timeline.txt $OUT/timeline_streamed.txt
python scripts/r5_step_times.py --steps 19 > $OUT/step_times_c2.json 2> /dev/null
python scripts/small_call_latency.py 2>&1 | grep "^D=" > $OUT/small_calls.txt
for cfg in "c5:--decimation 32" "c3:--mode hrit --decimation 1" "c1:--decimation 1"; do
  name=${cfg%%:*}; args=${cfg#*:}
  python $R/bench.py --steps 20 --warmup 5 $args > $OUT/bench_$name.json 2> /dev/null
  ( cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$name -o $name -- python $R/bench.py --steps 20 --warmup 5 --no-cpu --no-profile --no-exact --no-serial-floor $args > $OUT/stats_$name.log 2>&1 )
done
# keep what the summariser reads, drop the raw traces (tens of MB)
find $OUT -name "*_kernel_trace.csv" -delete
find $OUT -name "*_agent_info.csv" -delete
ls -R $OUT | head -80
