#!/bin/bash
# Runs on the GPU box: timeline of the last timed C2 burst -- kernel time, gaps between kernels, small kernels.
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tr -o c2 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu --no-profile > /dev/null 2>&1
python - <<'PY'
import csv, os
f = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/tr/c2_kernel_trace.csv'
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
starts = [i for i, r in enumerate(rows) if 'fir_decim_kernel<3' in r['Kernel_Name']]
a, b = starts[-2], starts[-1]          # one whole burst: from a decimator launch to the next
seg = rows[a:b]
t0, t1 = int(seg[0]['Start_Timestamp']), int(rows[b]['Start_Timestamp'])
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg)
small = [r for r in seg if int(r['End_Timestamp']) - int(r['Start_Timestamp']) < 20000]
gaps = [int(seg[i + 1]['Start_Timestamp']) - int(seg[i]['End_Timestamp']) for i in range(len(seg) - 1)]
gaps.append(t1 - int(seg[-1]['End_Timestamp']))
print("burst %.3f ms: %d launches, kernel time %.3f ms, idle between kernels %.3f ms (largest %.1f us, after %s)" % (
    (t1 - t0) / 1e6, len(seg), busy / 1e6, sum(g for g in gaps if g > 0) / 1e6, max(gaps) / 1e3,
    seg[gaps.index(max(gaps))]['Kernel_Name'][:40]))
print("launches under 20 us: %d, together %.3f ms" % (len(small), sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in small) / 1e6))
big = sorted(gaps, reverse=True)[:6]
print("largest gaps (us):", [round(g / 1e3, 1) for g in big])
PY
