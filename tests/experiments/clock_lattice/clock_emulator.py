"""CPU emulator of the device's time-tiled Mueller & Mueller evaluation (csrc/clock.hip, csrc/newton.h), used in
round 2 to find out where the 2.2e-4 rms against the serial loop comes from and what could be done about it.
Test infrastructure: it uses the oracle (through mm_lab.c, the same float32 statements as oracle/xrit_oracle.c, with
start states and per-symbol traces exposed).  Not collected by pytest.

    python tests/experiments/clock_lattice/clock_emulator.py gen            # 16 Mi samples -> oracle Costas output (/tmp)
    python tests/experiments/clock_lattice/clock_emulator.py sensitivity    # serial loop: input perturbed by 1e-7 .. 2e-6
    python tests/experiments/clock_lattice/clock_emulator.py tiled [NS]     # the multiple-shooting passes, chain length NS
    python tests/experiments/clock_lattice/clock_emulator.py jacobian       # effective Jacobian at the floor
    python tests/experiments/clock_lattice/clock_emulator.py meanj          # one mean Jacobian for every chain

What it showed:
  * mu and omega live on a lattice: omega ~ 4.25 is a float32 with ulp 2^-21 = 4.8e-7 and is updated by
    round(7.18 mm) ulps per symbol; mu = frac(mu + omega + 0.0037 mm) is rounded onto the same lattice.  A trajectory
    that is one omega-ulp off sits ~5e-4 sample away in mu until a rounding goes the other way, thousands of symbols
    later; two trajectories merge bit for bit only after ~1e5 symbols.
  * sensitivity: the serial loop fed with its own input perturbed by 1e-7 / 5e-7 / 2e-6 relative differs by
    5.4e-5 / 7.3e-5 / 9.5e-5 rms (omega one ulp apart for 3 .. 16 % of the symbols): that is the floor of ANY
    implementation that is not bit-identical up to the Costas output.
  * tiled: residuals fall 3.7e-2 -> 6.6e-4 -> 2.4e-4 -> 1.4e-4 -> 1.1e-4 -> 7.5e-5 and stay; the starts are then
    5e-4 sample / one omega-ulp (half of the boundaries) away from the serial trajectory, every pass moves them by as
    much again: 2.1..2.3e-4 rms in the symbols, the same for chains of 112, 256, 512 and 1024 symbols.
  * the chain Jacobians scatter by 2 % around (0.926, 108, -6.5e-5, 0.996); with the mean for every chain the passes
    converge as with each chain's own -> the finite-difference pass runs once per stream (ClockStage::begin)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

TMP = os.environ.get("XRIT_LAB_TMP", "/tmp/xrit_clock_lab")
os.makedirs(TMP, exist_ok=True)
SO = os.path.join(TMP, "libmmlab.so")
if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(os.path.join(HERE, "mm_lab.c")):
    subprocess.check_call(["gcc", "-O2", "-mavx2", "-ffp-contract=off", "-fopenmp", "-shared", "-fPIC", "-o", SO,
                           os.path.join(HERE, "mm_lab.c"), "-lm"])
L = C.CDLL(SO)
st_dt = np.dtype([('ii', '<i8'), ('mu', '<f4'), ('omega', '<f4'), ('p0', '<f4', 2), ('p1', '<f4', 2), ('c0', '<f4', 2), ('c1', '<f4', 2)])
par_dt = np.dtype([('omega_mid', '<f4'), ('omega_lim', '<f4'), ('gain_omega', '<f4'), ('gain_mu', '<f4')])
table = oracle.mmse_table().astype(np.float32).ravel()
f32 = np.float32
H_T, H_W = f32(0.0625), f32(1e-3)
ULP = f32(4.76837158203125e-07)
SPS = f32(1.25e6 / f32(293883))


def P(a):
    return a.ctypes.data_as(C.c_void_p)


def make_par(sps=SPS):
    p = np.zeros(1, par_dt)
    p['omega_mid'] = f32(sps); p['omega_lim'] = f32(sps) * f32(0.005)
    p['gain_omega'] = f32(0.0037) * f32(0.0037) / f32(4.0); p['gain_mu'] = f32(0.0037)
    return p


def run_chains(x, S, ns, par, out=False, trace=False):
    K = len(S); E = np.zeros(K, st_dt); ni = len(x) - 8 - 16
    o = np.zeros(K * ns, np.complex64) if out else None
    mu = np.zeros(K * ns, np.float32) if trace else None
    om = np.zeros(K * ns, np.float32) if trace else None
    arm = np.zeros(K * ns, np.int32) if trace else None
    prod = np.zeros(K, np.int32)
    L.run_chains(P(x), C.c_int64(ni), P(table), P(par), P(S), P(E), K, ns, P(o) if out else None,
                 P(mu) if trace else None, P(om) if trace else None, P(arm) if trace else None, P(prod))
    return E, o, mu, om, arm, prod


def start_state(sps=SPS):
    S = np.zeros(1, st_dt); S['mu'] = 0.5; S['omega'] = f32(sps)
    return S


def shift(S, dt):
    S = S.copy(); m = (S['mu'] + dt.astype(f32)).astype(f32); fl = np.floor(m)
    S['ii'] += fl.astype(np.int64); S['mu'] = (m - fl).astype(f32)
    return S


def tdiff(a, b):
    return ((a['ii'] - b['ii']).astype(f32) + (a['mu'] - b['mu']).astype(f32)).astype(f32)


class Emu:
    """The device's passes: chains of NS symbols from guessed starts, finite-difference Jacobians in the first
    pass, one Newton step on the multiple-shooting system per pass (sequential here, a scan on the device)."""

    def __init__(self, z, NS, nsym, seed=0, guess_sigma=2.6e-2):
        self.z, self.NS, self.par = z, NS, make_par()
        self.K = nsym // NS
        E, o, mu, om, arm, _ = run_chains(z, start_state(), self.K * NS, self.par, out=True, trace=True)
        self.true_out, self.true_arm = o, arm
        St = np.zeros(self.K + 1, st_dt); cur = start_state(); St[0] = cur[0]
        for k in range(self.K):
            cur, *_ = run_chains(z, cur, NS, self.par)
            St[k + 1] = cur[0]
        self.St = St
        rng = np.random.default_rng(seed)
        g = shift(St[:self.K].copy(), (guess_sigma * rng.standard_normal(self.K)).astype(f32)); g['omega'] = SPS
        g[0] = St[0]
        self.S, self.J = g, None

    def run(self, jac=False):
        self.E, self.out, _, _, self.arm, _ = run_chains(self.z, self.S, self.NS, self.par, out=True, trace=True)
        if jac:
            Et, *_ = run_chains(self.z, shift(self.S, np.full(self.K, H_T)), self.NS, self.par)
            Sw = self.S.copy(); Sw['omega'] = (Sw['omega'] + H_W).astype(f32)
            Ew, *_ = run_chains(self.z, Sw, self.NS, self.par)
            J = np.zeros((self.K, 2, 2), f32)
            J[:, 0, 0] = tdiff(Et, self.E) / H_T; J[:, 0, 1] = tdiff(Ew, self.E) / H_W
            J[:, 1, 0] = (Et['omega'] - self.E['omega']) / H_T; J[:, 1, 1] = (Ew['omega'] - self.E['omega']) / H_W
            self.J = J

    def residuals(self):
        r1 = tdiff(self.E[:-1], self.S[1:]); m = np.rint(r1 / self.E['omega'][:-1])
        return (r1 - m * self.E['omega'][:-1]).astype(f32), (self.E['omega'][:-1] - self.S['omega'][1:]).astype(f32), m

    def solve(self):
        r1, r2, _ = self.residuals()
        d = np.zeros(2, f32); j1s = np.zeros(self.K - 1, f32); j2s = np.zeros(self.K - 1, f32)
        for k in range(self.K - 1):
            j = (self.J[k] @ d).astype(f32); j1s[k], j2s[k] = j
            d = (np.array([r1[k], r2[k]], f32) + j).astype(f32)
        nw = shift(self.E[:-1], j1s); nw['omega'] = (self.E['omega'][:-1] + j2s).astype(f32)
        self.S = self.S.copy(); self.S[1:] = nw

    def report(self, tag=''):
        r1, r2, m = self.residuals()
        et = tdiff(self.S, self.St[:self.K]); eu = np.rint((self.S['omega'] - self.St['omega'][:self.K]) / ULP)
        d = self.out.real - self.true_out.real
        print(f"{tag} residual t rms {np.sqrt(np.mean(r1.astype(np.float64) ** 2)):.2e}, omega != 0 at {np.mean(r2 != 0):.3f} | "
              f"starts vs serial: t rms {np.sqrt(np.mean(et.astype(np.float64) ** 2)):.2e}, omega off at {np.mean(eu != 0):.3f} "
              f"(rms {np.sqrt(np.mean(eu ** 2)):.2f} ulp) | symbols rms {np.sqrt(np.mean(d.astype(np.float64) ** 2)):.3e}, "
              f"arms differ {np.mean(self.arm != self.true_arm):.4f}")


def costas():
    f = os.path.join(TMP, "costas.npy")
    if not os.path.exists(f):
        raise SystemExit("run `clock_emulator.py gen` first")
    return np.load(f)


def main():
    cmd = sys.argv[1] if len(sys.argv) > 1 else "tiled"
    if cmd == "gen":
        import os as _os, sys as _sys
        _sys.path.insert(0, _os.path.normpath(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), '../..')))
        import synth  # tests/synth.py: the NumPy specification of the synthetic burst (test infrastructure)
        N, p = 16 << 20, synth.SynthParams()
        x = np.concatenate([synth.generate(p, N // 8, start=i * (N // 8)) for i in range(8)])
        d = oracle.Demod(oracle.config("lrit", 1.25e6, 1)); d.process(x)
        np.save(os.path.join(TMP, "costas.npy"), d.stage("costas"))
        return
    z = costas()
    if cmd == "sensitivity":
        nsym = 3_900_000
        _, o0, mu0, om0, arm0, _ = run_chains(z, start_state(), nsym, make_par(), out=True, trace=True)
        rng = np.random.default_rng(1)
        for eps in (1e-7, 5e-7, 2e-6):
            zp = (z * (1 + eps * (rng.standard_normal(len(z)) + 1j * rng.standard_normal(len(z))) / np.sqrt(2))).astype(np.complex64)
            _, o1, mu1, om1, arm1, _ = run_chains(zp, start_state(), nsym, make_par(), out=True, trace=True)
            dmu = mu1 - mu0; dmu -= np.rint(dmu)
            print(f"input perturbed by {eps:g}: symbols rms {np.sqrt(np.mean((o1.real - o0.real) ** 2)):.3e}, arms differ "
                  f"{np.mean(arm0 != arm1):.4f}, mu rms {np.sqrt(np.mean(dmu ** 2)):.2e}, omega a ulp or more apart at "
                  f"{np.mean(np.rint((om1 - om0) / ULP) != 0):.3f} of the symbols")
    elif cmd == "tiled":
        e = Emu(z, int(sys.argv[2]) if len(sys.argv) > 2 else 112, 2_000_000)
        for p in range(10):
            e.run(jac=(p == 0)); e.report(f"pass {p}:"); e.solve()
    elif cmd == "jacobian":
        e = Emu(z, 112, 2_000_000)
        for p in range(9):
            e.run(jac=(p == 0))
            if p >= 1:
                dSt = tdiff(e.S, Sp); dSu = np.rint((e.S['omega'] - Sp['omega']) / ULP)
                dEt = tdiff(e.E, Ep)
                A = np.stack([dSt, dSu * ULP], 1).astype(np.float64)
                ct, *_ = np.linalg.lstsq(A, dEt.astype(np.float64), rcond=None)
                print(f"pass {p}: starts moved by {dSt.std():.2e} (omega at {np.mean(dSu != 0):.3f}); dE_t = {ct[0]:.3f} dS_t + "
                      f"{ct[1]:.1f} dS_w, unexplained {np.std(dEt - A @ ct):.2e}; finite-difference mean {e.J[:, 0, 0].mean():.3f}, {e.J[:, 0, 1].mean():.1f}")
            Sp, Ep = e.S.copy(), e.E.copy()
            e.solve()
    elif cmd == "meanj":
        for mode in ("per chain", "mean"):
            e = Emu(z, 112, 2_000_000); print("Jacobian:", mode)
            for p in range(7):
                e.run(jac=(p == 0))
                if p == 0 and mode == "mean":
                    e.J = np.broadcast_to(e.J.mean(0), e.J.shape).copy()
                e.report(f"  pass {p}:"); e.solve()


if __name__ == "__main__":
    main()
