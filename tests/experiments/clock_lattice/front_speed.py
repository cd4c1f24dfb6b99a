"""How fast does the bitwise-exact front advance?  Window of W chains, chain 0 starts from the exact state, the others
from floor-quality guesses; Newton + freeze passes as on the device (tol 2e-6 / 2e-7)."""
import sys
import numpy as np
from clock_emulator import *  # noqa

z = costas()
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 112
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
NP = int(sys.argv[3]) if len(sys.argv) > 3 else 40
e = Emu(z, NS, NS * W)
# first reach the floor with plain Newton passes
for p in range(6):
    e.run(jac=(p == 0)); e.solve()
e.run()
e.report("floor:")
TOL_T, TOL_W = f32(2e-6), f32(2e-7)


def same(a, b):
    return (a['ii'] == b['ii']) & (a['mu'] == b['mu']) & (a['omega'] == b['omega']) & \
        (a['p0'] == b['p0']).all(1) & (a['p1'] == b['p1']).all(1) & (a['c0'] == b['c0']).all(1) & (a['c1'] == b['c1']).all(1)


for p in range(NP):
    # solve with freeze
    r1, r2, m = e.residuals()
    d = np.zeros(2, f32)
    S = e.S.copy(); changed = 0
    for k in range(e.K - 1):
        j = (e.J[k] @ d).astype(f32)
        n = (np.array([r1[k], r2[k]], f32) + j).astype(f32)
        hist_same = (e.E['p0'][k] == e.S['p0'][k + 1]).all() and (e.E['p1'][k] == e.S['p1'][k + 1]).all()
        frozen = abs(n[0]) <= TOL_T and abs(n[1]) <= TOL_W and hist_same
        if not frozen:
            nw = shift(e.E[k:k + 1], j[:1]); nw['omega'] = (e.E['omega'][k] + j[1]).astype(f32)
            if not same(nw, e.S[k + 1:k + 2])[0]:
                S[k + 1] = nw[0]; changed += 1
        d = n
    e.S = S
    ex = same(e.S, e.St[:e.K])
    front = int(np.argmin(ex)) if not ex.all() else e.K
    e.run()
    d_ = e.out.real - e.true_out.real
    print(f"pass {p}: changed {changed}, exact starts {ex.sum()} / {e.K}, exact prefix {front}, symbols rms {np.sqrt(np.mean(d_.astype(np.float64)**2)):.3e}")
    if changed == 0:
        break
