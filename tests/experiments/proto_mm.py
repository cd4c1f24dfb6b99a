"""Development prototype (numpy): multiple-shooting Mueller & Mueller clock recovery.

Not product code and not the oracle: a CPU sketch used to choose chain length,
guess construction and the secant/Newton hand-off before writing csrc/clock.hip.
"""
import sys
import numpy as np

sys.path.insert(0, "/root/repo")
import oracle
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.normpath(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), '..')))
import synth  # tests/synth.py: the NumPy specification of the synthetic burst (test infrastructure)

f32 = np.float32
NT, FUDGE = 8, 16
H_T = 0.0625      # finite-difference step in t (samples)
H_W = 1e-3        # finite-difference step in omega
TRUST_T = 0.75
TOL_T = 2e-6
TOL_W = 2e-7


def clip(x, c):
    return np.clip(x, -c, c)


def bclip(x, c):
    """branchless_clip of the oracle: 0.5*(|x+c| - |x-c|), float32 roundings included."""
    x = x.astype(f32); c = f32(c)
    return (f32(0.5) * (np.abs(x + c).astype(f32) - np.abs(x - c).astype(f32)).astype(f32)).astype(f32)


def run_chains(x, NS, st, par, table, want_out=False):
    """st: dict of arrays (ii int64, mu, omega f32, p1,p2,c1,c2 complex64).  Advance NS symbols."""
    ii = st["ii"].copy(); mu = st["mu"].copy(); om = st["omega"].copy()
    p1 = st["p1"].copy(); p2 = st["p2"].copy(); c1 = st["c1"].copy(); c2 = st["c2"].copy()
    K = len(ii)
    ni = len(x) - NT - FUDGE
    omega_mid, omega_lim, g_om, g_mu = par
    out = np.zeros((K, NS), np.complex64) if want_out else None
    cnt = np.zeros(K, np.int64)
    xr = np.concatenate([x, np.zeros(64, np.complex64)])
    for n in range(NS):
        act = (ii < ni) & (ii >= 0)
        iic = np.clip(ii, 0, len(x))
        imu = np.rint(mu * f32(128)).astype(np.int64)
        rows = table[imu]                       # (K,8)
        win = xr[iic[:, None] + np.arange(8)[None, :]]   # (K,8)
        accr = np.zeros(K, f32); acci = np.zeros(K, f32)
        for k in range(8):
            accr = accr + rows[:, 7 - k] * win[:, k].real
            acci = acci + rows[:, 7 - k] * win[:, k].imag
        p0 = (accr + 1j * acci).astype(np.complex64)
        c0 = ((p0.real > 0).astype(f32) + 1j * (p0.imag > 0).astype(f32)).astype(np.complex64)
        dc = c0 - c2
        xre = dc.real * p1.real + dc.imag * p1.imag
        dp = p0 - p2
        yre = dp.real * c1.real + dp.imag * c1.imag
        mm = bclip((yre - xre).astype(f32), f32(1))
        omn = (om + g_om * mm).astype(f32)
        omn = (omega_mid + bclip((omn - omega_mid).astype(f32), omega_lim)).astype(f32)
        mun = (mu + omn + g_mu * mm).astype(f32)
        fl = np.floor(mun)
        iin = ii + fl.astype(np.int64)
        mun = (mun - fl).astype(f32)
        if want_out:
            out[act, n] = p0[act]
        cnt += act
        ii = np.where(act, iin, ii); mu = np.where(act, mun, mu); om = np.where(act, omn, om)
        p2 = np.where(act, p1, p2); p1 = np.where(act, p0, p1)
        c2 = np.where(act, c1, c2); c1 = np.where(act, c0, c1)
    return dict(ii=ii, mu=mu, omega=om, p1=p1, p2=p2, c1=c1, c2=c2), out, cnt


def shift_t(st, dt):
    s = {k: v.copy() for k, v in st.items()}
    m = (s["mu"].astype(np.float64) + dt)
    fl = np.floor(m)
    s["ii"] = s["ii"] + fl.astype(np.int64)
    s["mu"] = (m - fl).astype(f32)
    return s


def tdiff(a, b):
    """(t_a - t_b) in float64."""
    return (a["ii"] - b["ii"]).astype(np.float64) + (a["mu"].astype(np.float64) - b["mu"].astype(np.float64))


def guess_om(x, K, NS, sps, st0, blk=256):
    """Oerder & Meyr timing-phase estimate per block, unwrapped -> start position of each chain."""
    N = len(x)
    nb = N // blk
    n = np.arange(nb * blk)
    e = np.abs(x[:nb * blk]) ** 2 * np.exp(-2j * np.pi * n / sps)
    X = e.reshape(nb, blk).sum(1)
    # smooth over a few blocks
    Xs = np.convolve(X, np.ones(5), mode="same")
    ph = np.angle(Xs)
    d = np.diff(ph); d = (d + np.pi) % (2 * np.pi) - np.pi
    phu = np.concatenate([[ph[0]], ph[0] + np.cumsum(d)])
    # symbol instants: t = m*sps_true + tau; phase of line: -2 pi tau/sps relative to nominal grid
    # position (in samples) of symbol peaks near block centre c_b:  t = c_b + offset, offset = -phu/(2pi)*sps mod sps
    cb = (np.arange(nb) + 0.5) * blk
    # continuous "symbol count" function: count(t) = (t + phu(t)/(2pi)*sps)/sps
    cnt_b = (cb + phu / (2 * np.pi) * sps) / sps
    # anchor: chain 0 starts at st0 (ii0+mu0) as symbol #0
    t0 = float(st0["ii"][0]) + float(st0["mu"][0])
    c0 = np.interp(t0, cb, cnt_b)
    # want t_k with count(t_k) - c0 = round(...)?  symbol j of the run sits at count = c_anchor + j
    # lattice: counts at which M&M symbols sit = integer + frac_lock; estimate frac_lock as fractional part at lock.
    # M&M interpolation instant is ii+3+mu, i.e. its "t" is 3 samples before the eye centre.
    frac = 0.0  # eye centres are at integer counts by construction of O&M (|x|^2 peaks)
    ca = c0 + 3.0 / sps
    j0 = np.round(ca - frac)       # nearest lattice point to chain 0 start
    tk = np.empty(K)
    for k in range(K):
        target = j0 + k * NS + frac
        tk[k] = np.interp(target, cnt_b, cb) - 3.0
    return tk


def make_states(K, tk, sps, st0):
    st = dict(ii=np.floor(tk).astype(np.int64), mu=(tk - np.floor(tk)).astype(f32),
              omega=np.full(K, sps, f32),
              p1=np.zeros(K, np.complex64), p2=np.zeros(K, np.complex64),
              c1=np.zeros(K, np.complex64), c2=np.zeros(K, np.complex64))
    for k in st:
        st[k][0] = st0[k][0]
    return st


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 600000
    NS = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    mode = sys.argv[3] if len(sys.argv) > 3 else "lrit"
    esn0 = float(sys.argv[4]) if len(sys.argv) > 4 else 12.0
    if mode == "lrit":
        p = synth.SynthParams(esn0_db=esn0)
        cfg = oracle.config("lrit", 1.25e6, 1)
    else:
        p = synth.SynthParams(fs_in=2.5e6, symbol_rate=927000.0, alpha=0.3, esn0_db=esn0)
        cfg = oracle.config("hrit", 2.5e6, 1)
    xin = synth.generate(p, N)
    d = oracle.Demod(cfg)
    soft = d.process(xin)
    x = d.stage("costas")
    truth = d.stage("clock")
    sps = f32(d.sps)
    table = oracle.mmse_table()
    par = (sps, f32(sps * f32(0.005)), f32(cfg.clock_gain_omega), f32(cfg.clock_alpha))
    nsym_true = len(truth)
    K = int(np.ceil(N / (float(sps) * 0.995) / NS)) + 1
    st0 = dict(ii=np.array([0], np.int64), mu=np.array([0.5], f32), omega=np.array([sps], f32),
               p1=np.zeros(1, np.complex64), p2=np.zeros(1, np.complex64),
               c1=np.zeros(1, np.complex64), c2=np.zeros(1, np.complex64))
    tk = guess_om(x, K, NS, float(sps), st0)
    S = make_states(K, tk, sps, st0)
    ni = len(x) - NT - FUDGE
    for it in range(int(__import__('os').environ.get('PASSES','8'))):
        E, _, cnt = run_chains(x, NS, S, par, table)
        Et, _, _ = run_chains(x, NS, shift_t(S, H_T), par, table)
        Sw = {k: v.copy() for k, v in S.items()}; Sw["omega"] = (Sw["omega"] + f32(H_W)).astype(f32)
        Ew, _, _ = run_chains(x, NS, Sw, par, table)
        # Jacobian columns
        J = np.zeros((K, 2, 2))
        J[:, 0, 0] = tdiff(Et, E) / H_T
        J[:, 1, 0] = (Et["omega"].astype(np.float64) - E["omega"]) / H_T
        J[:, 0, 1] = tdiff(Ew, E) / H_W
        J[:, 1, 1] = (Ew["omega"].astype(np.float64) - E["omega"]) / H_W
        # newton scan
        Sn = {k: v.copy() for k, v in S.items()}
        delta = np.zeros(2); M = 0.0; nchg = 0; maxr = 0.0; rl = []
        for k in range(K - 1):
            if S["ii"][k] >= ni:   # chain k inactive => everything after is too
                for key in Sn:
                    Sn[key][k + 1] = E[key][k]
                continue
            rt = float(E["ii"][k] - S["ii"][k + 1]) + (float(E["mu"][k]) - float(S["mu"][k + 1]))
            rw = float(E["omega"][k]) - float(S["omega"][k + 1])
            om = float(E["omega"][k])
            m = np.rint(rt / om)
            r = np.array([rt - m * om, rw])
            Jd = J[k] @ delta
            if abs(delta[0]) > TRUST_T or abs(delta[1]) > 0.01:
                Jd = np.zeros(2)
            newt = Jd[0] + M * om
            delta = r + Jd
            hist_ok = (E["p1"][k] == S["p1"][k + 1] and E["p2"][k] == S["p2"][k + 1]
                       and E["c1"][k] == S["c1"][k + 1] and E["c2"][k] == S["c2"][k + 1])
            if abs(delta[0]) <= TOL_T and abs(delta[1]) <= TOL_W and M == 0 and m == 0 and hist_ok:
                delta = np.zeros(2)
            else:
                nchg += 1
                mm_ = float(E["mu"][k]) + newt
                fl = np.floor(mm_)
                Sn["ii"][k + 1] = E["ii"][k] + int(fl)
                Sn["mu"][k + 1] = f32(mm_ - fl)
                Sn["omega"][k + 1] = f32(float(E["omega"][k]) + Jd[1])
                for key in ("p1", "p2", "c1", "c2"):
                    Sn[key][k + 1] = E[key][k]
            M += m
            maxr = max(maxr, abs(r[0])); rl.append(abs(r[0]))
        S = Sn
        rl = np.array(rl)
        print(f"pass {it}: changed={nchg} max|r_t|={maxr:.3e} at {rl.argmax()} med={np.median(rl):.2e} p90={np.percentile(rl,90):.2e} p99={np.percentile(rl,99):.2e}")
        if nchg == 0:
            break
    import os
    JTOL = float(os.environ.get("JTOL", "0"))
    if JTOL > 0:
        for jt in range(60):
            E, _, cnt = run_chains(x, NS, S, par, table)
            nch = 0
            Sn = {k: v.copy() for k, v in S.items()}
            rts = []
            for k in range(K - 1):
                if S["ii"][k] >= ni:
                    continue
                rt = float(E["ii"][k] - S["ii"][k + 1]) + (float(E["mu"][k]) - float(S["mu"][k + 1]))
                rw = float(E["omega"][k]) - float(S["omega"][k + 1])
                rts.append(abs(rt))
                if abs(rt) > JTOL or abs(rw) > JTOL * 0.01:
                    nch += 1
                    for key in Sn:
                        Sn[key][k + 1] = E[key][k]
            S = Sn
            rts = np.array(rts)
            print(f"jacobi pass {jt}: moved={nch} rms_r={np.sqrt((rts**2).mean()):.2e} max={rts.max():.2e}")
            if nch == 0:
                break
    NSL = int(os.environ.get("NSL", "512"))
    step = NSL // NS
    SL = {k: v[::step].copy() for k, v in S.items()}
    E, out, cnt = run_chains(x, NSL, SL, par, table, want_out=True)
    KL = len(SL["ii"])
    # residuals at long boundaries
    rt = (E["ii"][:-1] - SL["ii"][1:]).astype(np.float64) + (E["mu"][:-1].astype(np.float64) - SL["mu"][1:])
    act = SL["ii"][1:] < ni
    print("long-chain boundary residual rms", np.sqrt(np.mean(rt[act] ** 2)), "max", np.abs(rt[act]).max())
    sym = np.concatenate([out[k, :cnt[k]] for k in range(KL)])
    n = min(len(sym), nsym_true)
    print("symbols", len(sym), "oracle", nsym_true)
    err = sym[:n] - truth[:n]
    a = np.abs(err)
    q = n // 8
    print("per-eighth rms", [f"{np.sqrt(np.mean(a[i*q:(i+1)*q]**2)):.2e}" for i in range(8)])
    print("rms err", np.sqrt(np.mean(a ** 2)), "max", a.max(), "n>1e-3:", (a > 1e-3).sum(),
          "sign mismatches", (np.sign(sym[:n].real) != np.sign(truth[:n].real)).sum())


if __name__ == "__main__":
    main()
