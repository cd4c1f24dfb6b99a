#!/bin/bash
# Runs on the GPU box (via gpurun): round 6's profile set.  The default configuration with the headline's command (as round 5),
# and the parity mode (cfg.front_exact = 2) beside it.  Outputs under gpurun_out/<tag>/; scripts/summarize_profiles.py <tag>
# turns them into profiles/<tag>_*; the parity mode's files are copied by this script's caller (see the end).
set -u
TAG=${1:-r6}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd $R
python bench.py --steps 20 --warmup 5 > $OUT/bench_c2.json 2> $OUT/bench_c2.err      # (the driver's command)
python bench.py --no-prefetch --no-cpu --no-exact --no-other-configs > $OUT/bench_c2_noprefetch.json 2> /dev/null
Q="--no-cpu --no-profile --no-exact --no-other-configs"
cd /tmp && export TMPDIR=/tmp
# kernel statistics over 20 streamed steps (beside), and one burst at a time (alone)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o c2 -- python $R/bench.py --steps 20 --warmup 5 $Q > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_np -o c2 -- python $R/bench.py --steps 20 --warmup 5 $Q --no-prefetch > $OUT/stats_np.log 2>&1
# the same for the parity mode
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_fe2 -o c2 -- python $R/bench.py --steps 20 --warmup 5 $Q --front-exact 2 > $OUT/stats_fe2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_fe2_np -o c2 -- python $R/bench.py --steps 10 --warmup 3 $Q --front-exact 2 --no-prefetch > $OUT/stats_fe2_np.log 2>&1
# HBM traffic: separate counter passes, the headline's command, three timed steps behind three warm-up steps
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o c2 -- python $R/bench.py --steps 3 --warmup 3 $Q > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o c2 -- python $R/bench.py --steps 3 --warmup 3 $Q > $OUT/pmc_write.log 2>&1
cd $R
bash scripts/r6_timeline.sh --no-other-configs > /dev/null 2>&1
cp gpurun_out/r6_timeline.txt $OUT/timeline_streamed.txt
bash scripts/r6_timeline.sh --no-other-configs --front-exact 2 > /dev/null 2>&1
cp gpurun_out/r6_timeline.txt $OUT/timeline_streamed_fe2.txt
python scripts/r5_step_times.py --steps 19 > $OUT/step_times_c2.json 2> /dev/null
python scripts/small_call_latency.py 2>&1 | grep "^D=" > $OUT/small_calls.txt
for cfg in "c5:--decimation 32" "c3:--mode hrit --decimation 1" "c1:--decimation 1"; do
  name=${cfg%%:*}; args=${cfg#*:}
  python $R/bench.py --steps 20 --warmup 5 --no-other-configs $args > $OUT/bench_$name.json 2> /dev/null
  python $R/bench.py --steps 10 --warmup 4 --no-other-configs --no-exact --front-exact 2 $args > $OUT/bench_${name}_fe2.json 2> /dev/null
  ( cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$name -o $name -- python $R/bench.py --steps 20 --warmup 5 $Q --no-serial-floor $args > $OUT/stats_$name.log 2>&1 )
done
python $R/bench.py --steps 20 --warmup 5 --no-other-configs --no-exact --front-exact 2 > $OUT/bench_c2_fe2.json 2> /dev/null
# keep what the summariser reads, drop the raw traces (tens of MB)
find $OUT -name "*_kernel_trace.csv" -delete
find $OUT -name "*_agent_info.csv" -delete
ls -R $OUT | head -100
