#!/bin/bash
# Runs on the GPU box: kernels of one steady-state burst of the DEFAULT configuration fed like bench.py feeds it (the front end
# of burst b + 1 registered before burst b is processed), both hardware queues, from rocprofv3 --kernel-trace: shows the
# next burst's front end running next to the relay kernels.  Output: gpurun_out/streamed_timeline.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/tr_streamed
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tr_streamed -o t -- python $R/scripts/stream_modes.py --exact ${1:-0} --bursts 5 > $R/gpurun_out/tr_streamed.log 2>&1
python - "$(find $R/gpurun_out/tr_streamed -name 't_kernel_trace.csv' | head -1)" > $R/gpurun_out/streamed_timeline.txt <<'PY'
import csv, sys, re
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'clock_relay_init' in r['Kernel_Name']]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]['Start_Timestamp'])
print("one burst, relay start to relay start: %.3f ms" % ((int(rows[b]['Start_Timestamp']) - t0) / 1e6))
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '').replace('xrit::', '')[:52]
    print("%9.1f .. %9.1f us  (%7.1f)  queue %s  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, r.get('Queue_Id', '?'), name))
PY
rm -rf $R/gpurun_out/tr_streamed
cat $R/gpurun_out/streamed_timeline.txt
