#!/bin/bash
# Like r5_timeline.sh, but the FIRST timed bursts of the run: how the pipeline fills.  -> gpurun_out/r5_timeline_fill.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/tr_r5f
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tr_r5f -o t -- python $R/bench.py --steps 8 --warmup 4 --no-cpu --no-exact --no-serial-floor --no-profile "$@" > $R/gpurun_out/tr_r5f.log 2>&1
python - "$(find $R/gpurun_out/tr_r5f -name 't_kernel_trace.csv' | head -1)" > $R/gpurun_out/r5_timeline_fill.txt <<'PY'
import csv, sys, re
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
rows = [r for r in rows if 'synth' not in r['Kernel_Name']]
dec = [i for i, r in enumerate(rows) if 'fir_decim_kernel<3, false, 0, 0, 151' in r['Kernel_Name']]
a = dec[4]          # the first timed burst's decimator (four warm-up bursts in front)
t0 = int(rows[a]['Start_Timestamp'])
for r in rows[a:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if (s - t0) / 1e3 > 9000: break
    if (e - s) < 25000 and 'overlap' not in r['Kernel_Name']: continue
    name = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '').replace('xrit::', '')[:52]
    print("%9.1f .. %9.1f us  (%7.1f)  queue %s  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, r.get('Queue_Id', '?'), name))
PY
rm -rf $R/gpurun_out/tr_r5f
cat $R/gpurun_out/r5_timeline_fill.txt
