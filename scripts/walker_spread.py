"""What a relay without pass boundaries could save (library built with make EXTRA="-DXRIT_EXPERIMENTS -DXRIT_RELAY_TIMING", XRIT_TRACE=1):
per-walker cycles of every pass from the "[xrit] walker p g cycles ..." lines on stdin; compares the sum over passes of the slowest
walker with the longest dependency chain (segment s of pass p + 1 waits for segments s and s - 1 of pass p only)."""
import sys, collections
t = collections.defaultdict(dict)
bursts = []
for line in sys.stdin:
    if line.startswith("[xrit] walker "):
        f = line.split()
        p, g, cyc = int(f[2]), int(f[3]), int(f[4])
        if p == 0 and g == 0 and t:
            bursts.append(t); t = collections.defaultdict(dict)
        t[p][g] = cyc
if t: bursts.append(t)
for b, t in enumerate(bursts):
    P = sorted(t)
    G = max(len(t[p]) for p in P)
    if G < 100: continue
    per_pass = [max(t[p].values()) for p in P]
    mean = [sum(t[p].values()) / len(t[p]) for p in P]
    fin = {g: 0 for g in range(G)}
    for p in P:
        new = {}
        for g in range(G):
            ready = max(fin.get(g, 0), fin.get(g - 1, 0) if g > 0 else 0)
            new[g] = ready + t[p].get(g, 0)
        fin = new
    print("burst %d: %d passes, %d segments; slowest walker per pass %s (mean %s) cycles; sum of maxima %.0f, longest chain %.0f: -%.1f %%" % (
        b, len(P), G, [int(v) for v in per_pass], [int(v) for v in mean], sum(per_pass), max(fin.values()), 100 * (1 - max(fin.values()) / sum(per_pass))))
