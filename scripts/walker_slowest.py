import sys, collections
rows=[]
for line in sys.stdin:
    if line.startswith("[xrit] walker "):
        f=line.split(); rows.append(tuple(int(v) for v in f[2:10]))
# take the last burst's rows: find last index where p==0 and g==0
start=max(i for i,r in enumerate(rows) if r[0]==0 and r[1]==0)
rows=rows[start:]
for p in (0,1,2):
    rs=[r for r in rows if r[0]==p]
    rs.sort(key=lambda r:-r[2])
    mean=sum(r[2] for r in rs)/len(rs)
    print("pass",p,"mean cycles %.0f wait %.0f steps %.0f rounds %.0f"%(mean, sum(r[3] for r in rs)/len(rs), sum(r[4] for r in rs)/len(rs), sum(r[5] for r in rs)/len(rs)))
    for r in rs[:8]:
        hw=r[6]; print("   seg %4d cycles %8d wait %7d steps %4d rounds %5d simd %d cu %3d xcc %d"%(r[1],r[2],r[3],r[4],r[5],(hw>>4)&3,(hw>>8)&255,r[7]&7))
    # histogram
    import math
    qs=sorted(r[2] for r in rs)
    print("   quantiles 50/90/99/max:", qs[len(qs)//2], qs[int(len(qs)*0.9)], qs[int(len(qs)*0.99)], qs[-1])
