"""Exact relay: G segments walked exactly from their start states, ends handed to the successors, until nothing
changes.  How many passes until every segment start is the serial trajectory's?"""
import sys
import numpy as np
from clock_emulator import *  # noqa

z = costas()
NS = 112
CPS = int(sys.argv[1]) if len(sys.argv) > 1 else 110      # chains per segment
G = int(sys.argv[2]) if len(sys.argv) > 2 else 128
e = Emu(z, NS, NS * CPS * G)
for p in range(6):
    e.run(jac=(p == 0)); e.solve()
e.run(); e.report("floor:")
idx = np.arange(G) * CPS
start = e.S[idx].copy()
truth = e.St[idx]


def same(a, b):
    return (a['ii'] == b['ii']) & (a['mu'] == b['mu']) & (a['omega'] == b['omega']) & \
        (a['p0'] == b['p0']).all(1) & (a['p1'] == b['p1']).all(1)


for p in range(1, 400):
    end, *_ = run_chains(z, start, NS * CPS, e.par)
    new = start.copy(); new[1:] = end[:-1]
    changed = (~same(new, start)).sum()
    start = new
    ok = same(start, truth)
    if p % 5 == 0 or changed == 0:
        print(f"pass {p}: changed {changed}, starts equal to the serial trajectory {ok.sum()} / {G}")
    if changed == 0:
        break
