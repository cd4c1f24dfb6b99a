#!/bin/bash
# Runs on the GPU box (round 6: any configuration, e.g. --front-exact 2): every kernel of two steady-state bursts fed like bench.py feeds it (two inputs
# registered behind the call in progress), all hardware queues, from rocprofv3 --kernel-trace.
# Usage: scripts/r5_timeline.sh [bench.py arguments]   ->  gpurun_out/r6_timeline.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/tr_r6
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tr_r6 -o t -- python $R/bench.py --steps 20 --warmup 4 --no-cpu --no-exact --no-serial-floor --no-profile "$@" > $R/gpurun_out/tr_r6.log 2>&1
python - "$(find $R/gpurun_out/tr_r6 -name 't_kernel_trace.csv' | head -1)" > $R/gpurun_out/r6_timeline.txt <<'PY'
import csv, sys, re
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
rows = [r for r in rows if 'synth' not in r['Kernel_Name']]
idx = [i for i, r in enumerate(rows) if 'clock_overlap_scan' in r['Kernel_Name'] or 'clock_relay_finalize' in r['Kernel_Name']]
m = len(idx) // 2          # the middle of the run: the pipeline is full, nothing drains yet
a, b = idx[m], idx[m + 2]
t0 = int(rows[a]['Start_Timestamp'])
print("two bursts, joints to joints: %.3f ms" % ((int(rows[b]['Start_Timestamp']) - t0) / 1e6))
for r in rows[a:b + 3]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '').replace('xrit::', '')[:52]
    print("%9.1f .. %9.1f us  (%7.1f)  queue %s  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, r.get('Queue_Id', '?'), name))
t1 = int(rows[b]['Start_Timestamp'])
busy = {}
for r in rows:
    s, e = max(int(r['Start_Timestamp']), t0), min(int(r['End_Timestamp']), t1)
    if e > s:
        busy[r.get('Queue_Id', '?')] = busy.get(r.get('Queue_Id', '?'), 0) + (e - s)
print("busy per queue over the two bursts: " + ", ".join("queue %s %.0f %%" % (q, 100.0 * v / (t1 - t0)) for q, v in sorted(busy.items())))
PY
rm -rf $R/gpurun_out/tr_r6
tail -2 $R/gpurun_out/tr_r6.log | cut -c1-300
cat $R/gpurun_out/r6_timeline.txt
