#!/bin/bash
# Runs on the GPU box: every kernel of the last timed burst in launch order -- start offset, duration, gap before --
# from rocprofv3 --kernel-trace of an un-bracketed bench run.  Output: gpurun_out/timeline_<tag>.txt
TAG=${1:-c2}
shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/tl_$TAG
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_$TAG -o t -- python $R/bench.py --steps 3 --warmup 2 --no-cpu --no-profile --no-exact "$@" > $R/gpurun_out/tl_$TAG.log 2>&1
F=$(find $R/gpurun_out/tl_$TAG -name 't_kernel_trace.csv' | head -1)
python - "$F" > $R/gpurun_out/timeline_$TAG.txt <<'PY'
import csv, sys, re
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
rows = [r for r in rows if 'synth_kernel' not in r['Kernel_Name']]
def first_of_burst(r):
    n = r['Kernel_Name']
    return 'fir_decim_kernel' in n or 'fir_poly_kernel' in n or 'fir_hist' in n
# a burst starts at the first front-end kernel after a clock kernel
starts = [i for i, r in enumerate(rows) if first_of_burst(r) and (i == 0 or 'clock' in rows[i - 1]['Kernel_Name'] or 'copyBuffer' in rows[i-1]['Kernel_Name'])]
a, b = starts[-2], starts[-1]
seg = rows[a:b]
t0 = int(seg[0]['Start_Timestamp'])
prev_end = t0
tot_k = tot_g = 0
print("burst: %.3f ms, %d launches" % ((int(rows[b]['Start_Timestamp']) - t0) / 1e6, len(seg)))
for r in seg:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '').replace('xrit::', '')[:58]
    gap = (s - prev_end) / 1e3
    print("%9.1f us  dur %8.1f  gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, name))
    tot_k += e - s
    tot_g += max(0, s - prev_end)
    prev_end = max(prev_end, e)
print("kernel time %.3f ms, idle %.3f ms" % (tot_k / 1e6, tot_g / 1e6))
PY
tail -3 $R/gpurun_out/timeline_$TAG.txt
