#!/bin/bash
# session script: GPU suite + streamed bench with the Costas trace
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s4; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "pytest rc $?" >> $O/gputest.log
XRIT_TRACE=1 timeout 600 python bench.py --steps 6 --warmup 3 --no-cpu --no-exact --no-serial-floor > $O/bench_trace.json 2> $O/bench_trace.err
timeout 600 python bench.py --no-cpu --no-exact > $O/bench.json 2> $O/bench.err
tail -3 $O/gputest.log; grep -c "costas pass" $O/bench_trace.err
