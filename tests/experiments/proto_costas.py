"""Development prototype (numpy): multiple-shooting Costas loop.

Not product code and not the oracle: a CPU sketch used to choose chain length,
guess construction and the Newton hand-off before writing csrc/costas.hip.
"""
import sys
import numpy as np

sys.path.insert(0, "/root/repo")
import oracle
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.normpath(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), '..')))
import synth  # tests/synth.py: the NumPy specification of the synthetic burst (test infrastructure)

f32 = np.float32
TWOPI = f32(2 * np.pi)
TRUST = 1.0
TOL_P = 2e-6
TOL_F = 1e-8
HEAD = 64


def gains(bw):
    d = f32(np.sqrt(f32(2.0)) / f32(2.0))
    den = f32(1.0) + f32(2.0) * d * f32(bw) + f32(bw) * f32(bw)
    return f32(4 * d * f32(bw) / den), f32(4 * f32(bw) * f32(bw) / den)


def run_chains(z, L, S, alpha, beta, want_out=False):
    K = len(S)
    N = len(z)
    phi = S[:, 0].astype(f32).copy()
    fr = S[:, 1].astype(f32).copy()
    # tangent columns: d/dphi0, d/df0
    Tp = np.stack([np.ones(K, f32), np.zeros(K, f32)], 1)
    Tf = np.stack([np.zeros(K, f32), np.ones(K, f32)], 1)
    zr = np.zeros((K, L), f32)
    zi = np.zeros((K, L), f32)
    valid = np.zeros((K, L), bool)
    flat_r = np.zeros(K * L, f32); flat_r[:N] = z.real
    flat_i = np.zeros(K * L, f32); flat_i[:N] = z.imag
    fv = np.zeros(K * L, bool); fv[:N] = True
    zr = flat_r.reshape(K, L); zi = flat_i.reshape(K, L); valid = fv.reshape(K, L)
    out = np.zeros((K, L), np.complex64) if want_out else None
    for n in range(L):
        v = valid[:, n]
        s = np.sin(phi).astype(f32); c = np.cos(phi).astype(f32)
        yr = zr[:, n] * c + zi[:, n] * s
        yi = zi[:, n] * c - zr[:, n] * s
        if want_out:
            out[:, n] = yr + 1j * yi
        e = yr * yi
        ed = (yi * yi - yr * yr) * (np.abs(e) < 1)
        e = np.clip(e, -1, 1).astype(f32)
        fn = fr + beta * e
        pn = phi + fn + alpha * e
        Tfn = Tf + (beta * ed)[:, None] * Tp
        Tpn = Tp + Tfn + (alpha * ed)[:, None] * Tp
        pn = np.where(pn > TWOPI, pn - TWOPI, pn)
        pn = np.where(pn < -TWOPI, pn + TWOPI, pn)
        fn = np.clip(fn, -1, 1)
        phi = np.where(v, pn, phi).astype(f32); fr = np.where(v, fn, fr).astype(f32)
        Tp = np.where(v[:, None], Tpn, Tp).astype(f32); Tf = np.where(v[:, None], Tfn, Tf).astype(f32)
    E = np.stack([phi, fr], 1)
    J = np.stack([Tp, Tf], 1)  # J[k] = [[dphi/dphi0, dphi/df0],[df/dphi0, df/df0]]
    return E, J, out


def guess_vv(z, L, K, S0):
    N = len(z)
    pad = np.zeros(K * L, np.complex64); pad[:N] = z
    c = (pad.reshape(K, L).astype(np.complex128) ** 2).sum(1)
    th2 = np.angle(c)                      # 2*theta mod 2pi at chain centres
    d = np.diff(th2); d = (d + np.pi) % (2 * np.pi) - np.pi
    th = np.concatenate([[th2[0]], th2[0] + np.cumsum(d)]) / 2   # unwrapped theta at centres
    # boundary k (start of chain k): between centre k-1 and centre k
    S = np.zeros((K, 2), f32)
    thb = np.empty(K); thb[1:] = 0.5 * (th[:-1] + th[1:]); thb[0] = th[0]
    fb = np.empty(K)
    w = 2
    for k in range(K):
        a = max(0, k - w); b = min(K - 1, k + w - 1)
        fb[k] = (th[b] - th[a]) / ((b - a) * L) if b > a else 0.0
    S[:, 0] = np.remainder(thb + np.pi, 2 * np.pi) - np.pi
    S[:, 1] = fb
    S[0] = S0
    # coarse chain-level loop model over the head
    alpha, beta = gains(0.0037)
    phi, fr = float(S0[0]), float(S0[1])
    for k in range(min(HEAD, K - 1)):
        pm = phi + fr * L / 2
        eb = (np.exp(-2j * pm) * c[k]).imag / 2
        phi = phi + L * fr + (alpha + beta * L / 2) * eb
        fr = fr + beta * eb
        S[k + 1] = (np.remainder(phi + np.pi, 2 * np.pi) - np.pi, fr)
    return S


def newton_update(S, E, J):
    K = len(S)
    Sn = S.copy()
    delta = np.zeros(2, np.float64)
    M = 0
    nchg = 0
    for k in range(K - 1):
        R = E[k].astype(np.float64) - S[k + 1].astype(np.float64)
        m = np.rint(R[0] / np.pi)
        r = R.copy(); r[0] -= m * np.pi
        Jd = J[k].astype(np.float64) @ delta
        if abs(delta[0]) > TRUST or abs(delta[1]) > TRUST / 256 or not np.isfinite(Jd).all():
            Jd = np.zeros(2)
        new = E[k].astype(np.float64) + np.array([np.pi * M, 0.0]) + Jd
        delta = r + Jd
        M = (M + int(m)) & 1
        new = new.astype(f32)
        if abs(delta[0]) <= TOL_P and abs(delta[1]) <= TOL_F and M == 0 and int(m) % 2 == 0:
            delta = np.zeros(2)
            new = S[k + 1]
        if not np.array_equal(new, S[k + 1]):
            nchg += 1
        Sn[k + 1] = new
    return Sn, nchg


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 600000
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    mode = sys.argv[3] if len(sys.argv) > 3 else "vv"
    esn0 = float(sys.argv[4]) if len(sys.argv) > 4 else 12.0
    p = synth.SynthParams(esn0_db=esn0)
    x = synth.generate(p, N)
    if mode == "noise":
        x = (x - synth.generate(synth.SynthParams(esn0_db=None), N)).astype(np.complex64)
    d = oracle.Demod(oracle.config("lrit", 1.25e6, 1))
    d.process(x)
    z = d.stage("rrc"); truth = d.stage("costas")
    alpha, beta = gains(0.0037)
    K = (N + L - 1) // L
    S0 = np.zeros(2, f32)
    if mode == "cold":
        S = np.zeros((K, 2), f32)
    else:
        S = guess_vv(z, L, K, S0)
    # true boundary states for diagnostics
    co = oracle.CostasLoop(0.0037)
    tb = np.zeros((K, 2))
    for k in range(K):
        tb[k] = (co.s.phase, co.s.freq)
        co.Work(z[k * L:(k + 1) * L])
    for it in range(40):
        E, J, _ = run_chains(z, L, S, alpha, beta)
        Sn, nchg = newton_update(S, E, J)
        dphi = (S[:, 0] - tb[:, 0] + np.pi / 2) % np.pi - np.pi / 2
        bad = np.abs(dphi) > 1e-5
        first_bad = np.argmax(bad) if bad.any() else K
        print(f"pass {it}: changed={nchg} max|dphi mod pi|={np.abs(dphi).max():.3e} "
              f"median={np.median(np.abs(dphi)):.2e} first_bad={first_bad}/{K} nbad={bad.sum()}")
        S = Sn
        if nchg == 0:
            break
    E, J, out = run_chains(z, L, S, alpha, beta, want_out=True)
    y = out.reshape(-1)[:N]
    err = y - truth
    print("final rms err vs oracle:", np.sqrt(np.mean(np.abs(err) ** 2)), "max", np.abs(err).max())


if __name__ == "__main__":
    main()


def serial_self(z, alpha, beta):
    """Serial run with the prototype's own arithmetic (math.sin on float32 values)."""
    import math
    N = len(z)
    out = np.zeros(N, np.complex64)
    phi = f32(0); fr = f32(0)
    zr = z.real.astype(f32); zi = z.imag.astype(f32)
    for n in range(N):
        s = f32(np.sin(phi)); c = f32(np.cos(phi))
        yr = f32(zr[n] * c) + f32(zi[n] * s)
        yi = f32(zi[n] * c) - f32(zr[n] * s)
        out[n] = complex(yr, yi)
        e = f32(yr * yi)
        e = min(max(e, f32(-1)), f32(1))
        fr = f32(fr + f32(beta * e))
        phi = f32(f32(phi + fr) + f32(alpha * e))
        if phi > TWOPI: phi = f32(phi - TWOPI)
        if phi < -TWOPI: phi = f32(phi + TWOPI)
        fr = min(max(fr, f32(-1)), f32(1))
    return out


def newton_update_par(S, E, J):
    """Scan-friendly variant: linear scan, cut flags from |delta_lin|, rescan, then element-wise freeze."""
    K = len(S)
    R = E[:-1].astype(np.float64) - S[1:].astype(np.float64)
    m = np.rint(R[:, 0] / np.pi)
    r = R.copy(); r[:, 0] -= m * np.pi
    mi = m.astype(np.int64) & 1
    Jm = J[:-1].astype(np.float64)

    def scan(cut):
        d = np.zeros((K, 2))     # d[k] = delta at boundary k (d[0]=0)
        for k in range(K - 1):
            Jd = np.zeros(2) if cut[k] else Jm[k] @ d[k]
            d[k + 1] = r[k] + Jd
        return d
    d1 = scan(np.zeros(K, bool))
    cut = (np.abs(d1[:, 0]) > TRUST) | (np.abs(d1[:, 1]) > TRUST / 256) | ~np.isfinite(d1).all(1)
    d = scan(cut)
    Mpre = np.concatenate([[0], np.cumsum(mi)[:-1]]) & 1   # parity accumulated before boundary k+1 (exclusive)
    Sn = S.copy(); nchg = 0
    for k in range(K - 1):
        Jd = np.zeros(2) if cut[k] else Jm[k] @ d[k]
        frozen = (abs(d[k + 1, 0]) <= TOL_P and abs(d[k + 1, 1]) <= TOL_F and Mpre[k] == 0 and mi[k] == 0)
        if not frozen:
            new = (E[k].astype(np.float64) + np.array([np.pi * Mpre[k], 0.0]) + Jd).astype(f32)
            if not np.array_equal(new, S[k + 1]):
                nchg += 1
            Sn[k + 1] = new
    return Sn, nchg
