cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tr -o c2 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu --no-profile > /dev/null 2>&1
python - <<'PY'
import csv,os,collections
f=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/tr/c2_kernel_trace.csv'
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
acc=collections.OrderedDict()
prev_end=None
gaps=collections.defaultdict(list)
for r in rows:
    k=r['Kernel_Name'].split('(')[0].replace('void xrit::','')[:50]
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    acc.setdefault(k,[]).append(d)
    if prev_end is not None: gaps[k].append((int(r['Start_Timestamp'])-prev_end)/1e3)
    prev_end=int(r['End_Timestamp'])
for k,v in acc.items():
    g=gaps[k]
    print(f"{k:52s} n={len(v):4d} avg={sum(v)/len(v):8.1f} us  gap_before avg={sum(g)/max(len(g),1):6.1f}")
PY
