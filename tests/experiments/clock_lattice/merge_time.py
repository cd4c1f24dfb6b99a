"""Distribution of the bitwise merge time of an exact M&M trajectory started from a floor-quality pseudo state
(tiled hand-off result) with the true serial trajectory."""
import sys
import numpy as np
from clock_emulator import *  # noqa

z = costas()
NS = 112
NCH = int(sys.argv[1]) if len(sys.argv) > 1 else 30000     # chains of truth
e = Emu(z, NS, NS * NCH)
for p in range(6):
    e.run(jac=(p == 0)); e.solve()
e.run(); e.report("floor:")
# pick origins every STEP chains; walk exactly chain by chain (all origins in parallel), compare with truth St
STEP = int(sys.argv[2]) if len(sys.argv) > 2 else 50
MAXC = int(sys.argv[3]) if len(sys.argv) > 3 else 6000      # chains to walk at most
orig = np.arange(1, NCH - MAXC - 1, STEP)
cur = e.S[orig].copy()
merged_at = np.full(len(orig), -1)
for c in range(MAXC):
    alive = merged_at < 0
    if not alive.any():
        break
    tr = e.St[orig + c]
    same = (cur['ii'] == tr['ii']) & (cur['mu'] == tr['mu']) & (cur['omega'] == tr['omega']) & \
        (cur['p0'] == tr['p0']).all(1) & (cur['p1'] == tr['p1']).all(1)
    merged_at[alive & same] = c
    nxt, *_ = run_chains(z, cur, NS, e.par)
    cur = nxt
m = merged_at[merged_at >= 0] * NS
print(f"{len(orig)} origins, merged {len(m)}; symbols to merge: median {np.median(m):.0f}, mean {m.mean():.0f}, 90% {np.percentile(m, 90):.0f}, "
      f"99% {np.percentile(m, 99):.0f}, max {m.max()}; not merged within {MAXC * NS}: {(merged_at < 0).sum()}")
np.save("/tmp/xrit_clock_lab/merge.npy", merged_at)
