"""Round 6: fuzz of the contiguous split (xrit_group_*, in-process fabric) in the regime where a single chain's symbols are the CPU
chain's word for word -- the bit-exact front end, slices of one exact walk.  Random mode (LRIT / HRIT), decimation, world (2 or 3
ranks), calls (1 .. 3, the later ones a ring), slice length, carrier offset and start phase, clock offset, Es/N0, front_exact (0 or
2): the joined symbols must be np.array_equal to the oracle's.   python scripts/r6_group_fuzz.py [cases] [seed]"""
import os, sys, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import oracle, synth
import xritdemod_amd as xa
def rms(a): return float(np.sqrt(np.mean(np.abs(a) ** 2))) if len(a) else 0.0
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 606)
dev = torch.device("cuda", 0)
bad = 0
for case in range(cases):
    mode, fs, D, sr, al = [("lrit", 1.25e6, 1, 293883.0, 0.5), ("lrit", 6.25e6, 5, 293883.0, 0.5), ("hrit", 2.5e6, 1, 927000.0, 0.3),
                           ("lrit", 2.5e6, 2, 293883.0, 0.5)][int(rng.integers(4))]
    world, calls = int(rng.integers(2, 4)), int(rng.integers(1, 4))
    fe = int(rng.choice([0, 2]))
    cfg = lambda: xa.Demodulator.config(mode, fs, D, front_exact=fe)
    probe = xa.Group(cfg(), 0, fabric=xa.LocalFabric(2))
    halo = probe.halo_samples
    del probe
    sps = fs / D / sr
    hi = int(195000 * sps * D)      # (the single-walk limit, 200 k symbols, in both modes)
    relay = os.environ.get("GROUP_FUZZ_REGIME", "walk") == "relay"     # slices of 1 .. 4 times the single-walk limit: relayed, no hand-over
    n = int(rng.integers(hi + 6000 * D, 4 * hi)) if relay else int(rng.integers(halo + 1000 * D, max(halo + 2000 * D, min(hi, 4 * halo))))
    n -= n % D
    p = synth.SynthParams(fs_in=fs, symbol_rate=sr, alpha=al, carrier_hz=float(rng.uniform(-800, 800)), phase0=float(rng.uniform(0, 6.28)),
                          timing_offset=float(rng.uniform(0, 1)), clock_ppm=float(rng.uniform(-40, 40)), esn0_db=float(rng.uniform(float(os.environ.get("ESN0_LO", "7")), float(os.environ.get("ESN0_HI", "20")))),
                          seed=int(rng.integers(1, 1 << 30)))
    x = synth.generate(p, world * calls * n)
    want = oracle.Demod(oracle.config(mode, fs, D)).process(x)
    fabric = xa.LocalFabric(world)
    xt = torch.from_numpy(x.view(np.float32).reshape(-1, 2)).to(dev)
    parts, cnts, err = {}, {}, []
    def rank_main(r):
        try:
            g = xa.Group(cfg(), r, fabric=fabric)
            cap = int(n / (D * sps * 0.98)) + 1024
            soft = torch.empty(cap, dtype=torch.float32, device=dev)
            for c in range(calls):
                sl = xt[(world * c + r) * n:(world * c + r + 1) * n].contiguous()
                k, off, pol = g.process_slice_device(sl.data_ptr(), n, soft.data_ptr(), cap)
                parts[(c, r)] = (soft[:k].cpu().numpy().copy(), off, pol)
            cnts[r] = g.counters()
        except Exception as e:
            err.append(e)
    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    [t.start() for t in th]; [t.join(timeout=300) for t in th]
    tag = f"case {case}: {mode} fs {fs:g} D {D} world {world} calls {calls} n {n} ({n / D / sps / 1e3:.0f} k symbols, halo {halo}) front_exact {fe} " \
          f"carrier {p.carrier_hz:+.0f} Hz phase0 {p.phase0:.2f} ppm {p.clock_ppm:+.0f} Es/N0 {p.esn0_db:.1f} dB"
    if err or any(t.is_alive() for t in th):
        print(tag, "-> ERROR", err, flush=True); bad += 1; continue
    got = np.concatenate([parts[(c, r)][0] for c in range(calls) for r in range(world)])
    same = len(got) == len(want) and np.array_equal(got.view(np.uint32), want.view(np.uint32))
    if relay and len(got) == len(want):
        big = np.abs(want) > 1e-3
        pos, each = 0, []
        for c in range(calls):
            for r in range(world):
                k = len(parts[(c, r)][0]); each.append(rms(parts[(c, r)][0] - want[pos:pos + k])); pos += k
        ok = np.array_equal(np.sign(got[big]), np.sign(want[big])) and max(each) < 2e-4
        bad += 0 if ok else 1
        print(tag, "-> relayed slices:", "decisions equal," if np.array_equal(np.sign(got[big]), np.sign(want[big])) else "DECISIONS DIFFER,",
              "rms per slice", ["%.1e" % e for e in each], "first locks", [parts[(c, r)][2] for c in range(calls) for r in range(world)], flush=True)
        continue
    pols = [parts[(c, r)][2] for c in range(calls) for r in range(world)]
    if not same:
        bad += 1
        pos, each = 0, []
        for c in range(calls):
            for r in range(world):
                k = len(parts[(c, r)][0]); w = want[pos:pos + k]
                each.append("%.1e" % rms(parts[(c, r)][0] - w) if len(w) == k else "len"); pos += k
        print(tag, f"-> DIFFERS: symbols {len(got)} / {len(want)}, per slice {each}, first locks {pols}, counters {cnts}", flush=True)
    else:
        print(tag, f"-> word for word; first locks {pols}, (second starts, hand-overs, joined) per rank {[cnts[r] for r in range(world)]}", flush=True)
print(f"{cases - bad} of {cases} cases " + ("within 2e-4 per slice with equal decisions" if os.environ.get("GROUP_FUZZ_REGIME", "walk") == "relay" else "word for word"))
