import csv, sys, re
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
rows = [r for r in rows if 'synth_kernel' not in r['Kernel_Name']]
key = sys.argv[2] if len(sys.argv)>2 else 'fir_decim_kernel<3'
starts = [i for i, r in enumerate(rows) if key in r['Kernel_Name']]
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
