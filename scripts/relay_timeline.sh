#!/bin/bash
# Runs on the GPU box: duration of every relay pass (clock_relay kernels) of the last burst of scripts/relay_burst.py,
# from rocprofv3 --kernel-trace.  Output: gpurun_out/relay_timeline.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/tl_relay
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_relay -o t -- python $R/scripts/relay_burst.py --log2 ${1:-28} --bursts 2 > $R/gpurun_out/tl_relay.log 2>&1
F=$(find $R/gpurun_out/tl_relay -name 't_kernel_trace.csv' | head -1)
python - "$F" > $R/gpurun_out/relay_timeline.txt <<'PY'
import csv, sys, re
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'clock_relay_init' in r['Kernel_Name']]
a = idx[-1]
t0 = int(rows[a]['Start_Timestamp']); prev = t0
for r in rows[a:a + 60]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '').replace('xrit::', '')[:50]
    print("%9.1f us  dur %8.1f  gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, name))
    prev = e
PY
rm -rf $R/gpurun_out/tl_relay
cat $R/gpurun_out/relay_timeline.txt
