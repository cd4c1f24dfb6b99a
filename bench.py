#!/usr/bin/env python3
"""bench.py -- throughput of the xRIT BPSK demodulation chain on MI355X.

Workload (BASELINE.json configs[1], "C2"): LRIT BPSK at 293 883 sym/s, synthetic
complex-float IQ, 256 Mi samples per burst at an input rate of 6.25 Msps,
decimation 5 (151-tap Hamming low-pass) -> AGC -> 63-tap RRC (alpha 0.5) -> Costas
-> Mueller & Mueller.  One "step" = one pass of the chain over one burst that is
already resident in HBM; consecutive steps are consecutive bursts of one
continuous stream (loop and filter state carried from step to step, exactly as
the reference's processSamples() carries it from chunk to chunk,
/root/reference/demodulator/src/demodulator.cpp:100-168).

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL); every rank
demodulates its own independent capture segment (distinct seed) -- the path
shards by time slice / segment with no data-path collective, so scaling is weak
and the only collectives are the timing barrier and the max-over-ranks.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def run_contiguous(torch, dist, xa, _capi, world, rank, local_rank, dev, n_burst, D, fs_in, K, W, slices=None):
    """ONE stream per step, cut in `world` time slices of n_burst samples (BASELINE config 4): the C++ group API
    (xrit_group_*: ncclSend / ncclRecv of the halo, of 256 boundary symbols and of the clock recovery's carried state, two
    ncclAllGather of (polarity, status) and (count, status)).  The handles persist across steps, the slices of every step are generated and resident before the
    clock starts; torch.distributed only hands out the ncclUniqueId and brackets the timing.  `slices`: an existing
    (nbuf, n_burst, 2) float32 device tensor to generate into (the default leg reuses the headline's bursts).
    Returns the measurement as a dict (every rank; the reductions are collective)."""
    sp = _capi.synth_params(fs_in=fs_in)
    stream = torch.cuda.current_stream(dev)
    n_burst -= n_burst % D      # (slices are whole decimation periods: the next slice's decimator phase must not shift, group.hip)
    uid = [xa.group_unique_id() if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(uid, src=0)
    grp = xa.Group(xa.Demodulator.config("lrit", fs_in, D, device=local_rank), rank, world, uid[0])
    halo = grp.halo_samples
    rccl_ranks = grp.rccl_ranks
    if slices is None:
        free_b, _ = torch.cuda.mem_get_info(dev)
        nbuf = max(1, min(W + K, int(free_b * 0.6) // (n_burst * 8)))
        slices = torch.empty((nbuf, n_burst, 2), dtype=torch.float32, device=dev)
    nbuf = slices.shape[0]
    slices = slices[:, :n_burst]
    for t in range(min(W + K, nbuf)):          # step t, rank r: samples [(t * world + r) * n_burst, ...) of the stream
        _capi.synth_generate_device(sp, (t * world + rank) * n_burst, n_burst, slices[t].data_ptr(), device=local_rank,
                                    stream=stream.cuda_stream)
    cap = int(n_burst / (D * 4.2)) + 4096
    soft = torch.empty((cap,), dtype=torch.float32, device=dev)
    torch.cuda.synchronize(dev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    nsym, flips = 0, 0
    for t in range(W):
        grp.process_slice_device(slices[t % nbuf].data_ptr(), n_burst, soft.data_ptr(), cap, stream=stream.cuda_stream)
    barrier()
    t0 = time.perf_counter()
    for t in range(W, W + K):
        k, _off, pol = grp.process_slice_device(slices[t % nbuf].data_ptr(), n_burst, soft.data_ptr(), cap,
                                                stream=stream.cuda_stream)
        nsym += k
        flips += 1 if pol < 0 else 0
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed, float(nsym), float(flips)], dtype=torch.float64, device=dev)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed, nsym_all, flips_all = float(tmax[0].item()), float(t[1].item()), int(t[2].item())
    else:
        nsym_all, flips_all = float(nsym), flips
    relocks, handovers, joined = grp.counters()      # this rank's (rank 0 prints: its own boundaries)
    del grp
    return {"value": round(n_burst * world * K / elapsed / 1e6, 2), "unit": "Msamples/s", "steps": K, "warmup": W,
            "ms_per_step": round(elapsed / K * 1e3, 3), "symbols_per_s": round(nsym_all / elapsed, 1),
            "samples_per_step_per_gpu": n_burst, "halo_samples": int(halo), "halo_bytes_per_boundary": int(halo) * 8,
            "rccl_ranks": int(rccl_ranks), "polarity_flips": flips_all, "slices_reused": bool(W + K > nbuf),
            "rank0_boundaries": {"second_starts": relocks, "clock_handovers": handovers, "joined": joined},
            "what": "one LRIT stream per step cut in n_gpus time slices: ncclSend / ncclRecv of the halo (49 152 symbols), of 256 "
                    "boundary symbols and of the clock recovery's carried state (8 KB), two ncclAllGather of two words per rank "
                    "(xrit_group_process_slice_device)"}


def bench_contiguous(args, torch, dist, xa, _capi, world, rank, local_rank, dev, n_burst, D, fs_in):
    """--contiguous: the edge-exchange mode as the whole run (its own JSON line)."""
    K, W = args.steps, args.warmup
    r = run_contiguous(torch, dist, xa, _capi, world, rank, local_rank, dev, n_burst, D, fs_in, K, W)
    if rank == 0:
        print(json.dumps({
            "metric": "Msamples/s in -> soft-symbols/s out (LRIT 293 ksym/s chain); % HBM roofline",
            "value": r["value"], "unit": "Msamples/s", "n_gpus": world, "steps": K,
            "warmup": W, "ms_per_step": r["ms_per_step"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "symbols_per_s": r["symbols_per_s"],
            "config": {"workload": "C4 contiguous: one LRIT stream per step cut in n_gpus time slices, edge-sample exchange "
                                   "over RCCL (xrit_group_process_slice_device); slices resident before the clock starts",
                       "samples_per_step_per_gpu": n_burst, "decimation": D, "halo_samples": r["halo_samples"],
                       "halo_bytes_per_boundary": r["halo_bytes_per_boundary"], "slices_reused": r["slices_reused"],
                       "rccl_ranks": r["rccl_ranks"], "polarity_flips": r["polarity_flips"],
                       "parallelism": f"time-slice x{world}"}}))
    if world > 1:
        dist.destroy_process_group()


# ---- the alternative legs: the same stream from its first burst through ANOTHER configuration, never `value`
LEGS = {
    "parity_mode": dict(clock_exact=0, front_exact=2, steps=None, ahead=2,
                        what="cfg.front_exact = 2 (opt-in, round 6): the front end bit for bit the CPU chain's through the Costas loop -- "
                             "both filters summed in the CPU chain's order without FMA, the AGC and the Costas loop walked exactly "
                             "(csrc/exact_walk.h) --; the clock recovery is the default's, fed like `value`"),
    "warm_mode": dict(clock_exact=0, front_exact=1, steps=None, ahead=2,
                      what="cfg.front_exact = 1 (opt-in, round 5; `parity_mode` of round 5's bench): the Costas loop's final pass warms every "
                           "chain up over the four chains in front of it; everything else the default configuration, fed like `value`"),
    "exact_mode": dict(clock_exact=1, front_exact=0, steps=5, ahead=1,
                       what="cfg.clock_exact = 1: clock recovery relayed to closure, symbols bit-identical to the serial "
                            "float32 recurrence on this chain's Costas output"),
    "fast_mode": dict(clock_exact=-2, front_exact=0, steps=10, ahead=1,
                      what="cfg.clock_exact = -2 (the default of rounds 2-3): hand-off passes only, five on this signal, "
                           "relayed only when they stall; soft symbols 2.2e-4 .. 2.6e-4 rms from the CPU chain"),
    "quick_mode": dict(clock_exact=-3, front_exact=0, steps=10, ahead=1,
                       what="cfg.clock_exact = -3 (round 4): the default's relay with the passes in front of its last walked "
                            "approximately (one guess round, then two; no verification, nothing stored); soft symbols "
                            "1.15e-4 rms from the serial trajectory"),
}


def run_leg(key, xa, torch, cfg_of, bursts, nbuf, n_burst, soft, cap, stream, dev, K, prefetch, generate):
    """One alternative leg on a handle of its own (in this process).  Returns (result dict, soft symbols of burst 0, of burst 1)."""
    leg = LEGS[key]
    xd = xa.Demodulator(cfg_of(clock_exact=leg["clock_exact"], front_exact=leg["front_exact"]))
    ahead_n = max(1, min(leg["ahead"], xd.prefetch_depth(n_burst)))
    Kx, Wx = min(K, leg["steps"] or K), 2
    if generate is not None:
        for b in range(min(Wx + Kx, nbuf)):
            generate(b)
    torch.cuda.synchronize(dev)
    closed, rp, cp = True, [], []
    s0 = s1 = None
    for b in range(Wx):
        ns = xd.process_device(bursts[b % nbuf].data_ptr(), n_burst, soft.data_ptr(), cap, stream=stream.cuda_stream)
        if b == 0:
            s0 = soft[:ns].clone()
        if b == 1 and leg["front_exact"]:
            s1 = soft[:ns].clone()
    torch.cuda.synchronize(dev)
    x0 = time.perf_counter()
    # (streamed like the headline: inputs registered ahead of their calls)
    if prefetch:
        for q in range(min(ahead_n, Kx)):
            xd.prefetch_device(bursts[(Wx + q) % nbuf].data_ptr(), n_burst, stream=stream.cuda_stream)
    sx = None
    for b in range(Wx, Wx + Kx):
        if prefetch and b + ahead_n < Wx + Kx:
            xd.prefetch_device(bursts[(b + ahead_n) % nbuf].data_ptr(), n_burst, stream=stream.cuda_stream)
        xd.process_device(bursts[b % nbuf].data_ptr(), n_burst, soft.data_ptr(), cap, stream=stream.cuda_stream)
        sx = xd.stats()
        closed = closed and bool(sx.clock_relay_closed)
        rp.append(int(sx.clock_relay_passes))
        cp.append(int(sx.clock_passes))
    torch.cuda.synchronize(dev)
    x1 = time.perf_counter()
    res = {"what": leg["what"], "value": round(n_burst * Kx / (x1 - x0) / 1e6, 2), "unit": "Msamples/s", "steps": Kx,
           "ms_per_step": round((x1 - x0) / Kx * 1e3, 3), "front_end_of_next_burst_overlaps_loops": bool(prefetch),
           "hand_off_passes": cp, "relay_passes": rp,
           "closed": closed, "relay_segments": int(sx.clock_relay_segments) if sx is not None else 0}
    del xd
    return res, s0, s1


def leg_child(args):
    """bench.py --leg KEY --leg-dir DIR: one alternative leg in a process of its own (the first handle of the process: what a
    handle's speed is does not depend on what other legs created before).  Writes DIR/KEY.json and the soft symbols of
    bursts 0 / 1 as DIR/KEY_soft0.npy / _soft1.npy for the parent's parity comparison."""
    import torch
    import xritdemod_amd as xa
    from xritdemod_amd import _capi
    local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    n_burst, D, mode = 1 << args.burst_log2, args.decimation, args.mode
    fs_in = (1.25e6 if mode == "lrit" else 2.5e6) * D
    sym_rate, alpha = (293883.0, 0.5) if mode == "lrit" else (927000.0, 0.3)
    leg = LEGS[args.leg]
    Kx = min(args.steps, leg["steps"] or args.steps)
    free_b, _ = torch.cuda.mem_get_info(dev)
    nbuf = max(2, min(2 + Kx, int(free_b * 0.7) // (n_burst * 8)))
    bursts = torch.empty((nbuf, n_burst, 2), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev)
    sp = _capi.synth_params(fs_in=fs_in, symbol_rate=sym_rate, alpha=alpha, seed=0x58524954)

    def generate(b):
        _capi.synth_generate_device(sp, b * n_burst, n_burst, bursts[b % nbuf].data_ptr(), device=local_rank, stream=stream.cuda_stream)

    def cfg_of(**kw):
        return xa.Demodulator.config(mode, fs_in, D, device=local_rank, costas_chain_len=args.costas_chain,
                                     clock_chain_syms=args.clock_chain, **kw)

    sps = fs_in / D / sym_rate
    cap = int(n_burst / (D * sps * 0.99)) + 64
    soft = torch.empty((cap,), dtype=torch.float32, device=dev)
    res, s0, s1 = run_leg(args.leg, xa, torch, cfg_of, bursts, nbuf, n_burst, soft, cap, stream, dev, args.steps, not args.no_prefetch, generate)
    res["process"] = "a process of its own (the handle is its first)"
    if s0 is not None:
        np.save(os.path.join(args.leg_dir, args.leg + "_soft0.npy"), s0.cpu().numpy())
    if s1 is not None:
        np.save(os.path.join(args.leg_dir, args.leg + "_soft1.npy"), s1.cpu().numpy())
    with open(os.path.join(args.leg_dir, args.leg + ".json"), "w") as f:
        json.dump(res, f)


def run_children(cmds, timeout):
    """Fresh processes of this script, one after the other; returns [(returncode, stdout, stderr tail)]."""
    import subprocess
    outs = []
    for cmd in cmds:
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__)] + cmd, capture_output=True, text=True, timeout=timeout)
            outs.append((r.returncode, r.stdout, r.stderr[-600:]))
        except subprocess.TimeoutExpired:
            outs.append((-9, "", "timed out after %d s" % timeout))
    return outs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--burst-log2", type=int, default=28, help="log2 of samples per burst (28 = 256 Mi)")
    ap.add_argument("--decimation", type=int, default=5)
    ap.add_argument("--cpu-sample-log2", type=int, default=27, help="log2 of samples timed on the CPU oracle")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--costas-chain", type=int, default=0, help="samples per Costas chain (0 = library default)")
    ap.add_argument("--clock-chain", type=int, default=0, help="symbols per clock-recovery chain (0 = library default)")
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help="also time the oracle on this many host threads, one independent stream segment each "
                         "(SURVEY.md 8(d)(ii)); 0 = every logical CPU of the host; 1 = off.  The single-thread figure "
                         "stays the cpu_baseline")
    ap.add_argument("--no-prefetch", action="store_true",
                    help="one burst at a time: do not run the front end of the next burst (xrit_demod_prefetch_device, second "
                         "stream) under the feedback loops of the current one")
    ap.add_argument("--prefetch-depth", type=int, default=2,
                    help="inputs registered behind the call in progress (xrit_demod_prefetch_device): 2 (round 5: the walkers of "
                         "bursts b and b + 1 beside the front end and Costas loop of burst b + 2) or 1 (round 4)")
    ap.add_argument("--no-exact", action="store_true", help="skip the exact-mode, fast-mode and quick-mode legs (cfg.clock_exact = 1, -2, -3)")
    ap.add_argument("--no-serial-floor", action="store_true",
                    help="skip the serial-device run of the parity leg (cfg.clock_serial: ~0.3 us per symbol)")
    ap.add_argument("--front-exact", type=int, default=0, choices=[-1, 0, 1, 2],
                    help="cfg.front_exact of the measured handle (include/xritdemod_amd.h): 1 = the Costas loop's final pass warmed up over "
                         "four chains; 2 = the front end bit for bit the CPU chain's through the Costas loop (round 6); the default, 0, "
                         "is the configuration `value` is quoted on (the fast front end on bursts of a million symbols and more, the bit-exact "
                         "one on smaller calls); -1 = the fast one on calls of every size")
    ap.add_argument("--mode", choices=["lrit", "hrit"], default="lrit",
                    help="lrit: 293 883 sym/s, alpha 0.5, circuit rate 1.25 Msps (C2, C5; --decimation 1 = C1's chain); "
                         "hrit: 927 000 sym/s, alpha 0.3, circuit rate 2.5 Msps (C3)")
    ap.add_argument("--contiguous-leg", action="store_true",
                    help="add the `contiguous` leg (second timed region, same JSON line) also at N = 1, where the RCCL communicator "
                         "has one rank; with N > 1 over RCCL the leg is always run")
    ap.add_argument("--contiguous-timeout", type=int, default=240, help="seconds the contiguous leg may take before the line is printed without it")
    ap.add_argument("--leg", default=None, help=argparse.SUPPRESS)          # (internal: run ONE alternative leg in this process)
    ap.add_argument("--leg-dir", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--legs-in-process", action="store_true",
                    help="run the alternative legs (parity_mode, exact_mode, fast_mode, quick_mode) as further handles of THIS process, as "
                         "rounds 2-5 did; the default since round 6 is a fresh process per leg: how fast a handle is depends on which "
                         "hardware queues HIP deals its streams onto, and that depends on what the process created before")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short legs on BASELINE's other single-GPU configurations (C5, C1's chain at burst size, C3), each a "
                         "fresh process of this script")
    ap.add_argument("--contiguous", action="store_true",
                    help="N ranks demodulate ONE stream cut in N slices, with RCCL edge-sample exchange "
                         "(SURVEY.md 8(e); the default is N independent segments, no data-path collective)")
    args = ap.parse_args()
    if args.leg:
        return leg_child(args)

    import torch
    import torch.distributed as dist
    import xritdemod_amd as xa
    from xritdemod_amd import _capi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the chain has no CPU path)")
    # (XRIT_BENCH_SHARE_DEVICE=1, for the 1-GPU test box: the ranks share the devices there are and meet over gloo -- RCCL
    # refuses two ranks on one device; the driver's runs never set it)
    share = os.environ.get("XRIT_BENCH_SHARE_DEVICE") == "1"
    if share:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    rdev = torch.device("cpu") if share else dev          # where the reductions' tensors live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    n_burst = 1 << args.burst_log2
    D = args.decimation
    mode = args.mode
    fs_in = (1.25e6 if mode == "lrit" else 2.5e6) * D
    sym_rate, alpha = (293883.0, 0.5) if mode == "lrit" else (927000.0, 0.3)
    K, W = args.steps, args.warmup
    if args.contiguous:
        return bench_contiguous(args, torch, dist, xa, _capi, world, rank, local_rank, dev, n_burst, D, fs_in)
    detail = not args.no_profile        # a second, untimed set of K steps with every kernel bracketed
    nb = K + W + (K if detail else 0)

    # ---- synthetic stream: nb consecutive bursts of this rank's capture segment
    sp = _capi.synth_params(fs_in=fs_in, symbol_rate=sym_rate, alpha=alpha, seed=0x58524954 + 2 * rank)
    # The warm-up and timed bursts are all resident before the clock starts; the bursts of the detail pass are
    # generated into the same buffers afterwards.  If --steps asks for more bursts than fit in HBM the buffers
    # wrap (the stream then jumps back once per lap: config.bursts_reused says so).
    free_b, _total_b = torch.cuda.mem_get_info(dev)
    nbuf = max(2, min(W + K, int(free_b * 0.7) // (n_burst * 8)))
    bursts = torch.empty((nbuf, n_burst, 2), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev)

    def generate(b):
        _capi.synth_generate_device(sp, b * n_burst, n_burst, bursts[b % nbuf].data_ptr(), device=local_rank,
                                    stream=stream.cuda_stream)

    for b in range(min(W + K, nbuf)):
        generate(b)
    torch.cuda.synchronize(dev)
    host0 = None
    cpu_threads = args.cpu_threads if args.cpu_threads > 0 else (os.cpu_count() or 1)
    if rank == 0 and world == 1 and not args.no_cpu:
        # the CPU baseline's samples, taken before the detail pass reuses the buffers
        n_cpu0 = min(n_burst, 1 << args.cpu_sample_log2)
        # (a whole number of decimation periods: the chain drops n mod D samples per call like the reference,
        # demodulator.cpp:137, and the steady-state leg below goes on from here through the rest of the stream)
        n_cpu0 -= n_cpu0 % D
        host0 = bursts[0, :n_cpu0].cpu().numpy().view(np.complex64).reshape(-1)
        # (the parity leg also looks at a steady-state burst: the rest of burst 0 and burst 1, through the oracle after the
        # timed sample)
        host0_rest = bursts[0, n_cpu0:].cpu().numpy().view(np.complex64).reshape(-1) if n_cpu0 < n_burst else None
        host1 = bursts[1].cpu().numpy().view(np.complex64).reshape(-1) if (W >= 2 and nbuf >= 2 and not args.no_serial_floor) else None
        host_segs = []
        if cpu_threads > 1:
            # one segment per thread: consecutive pieces of the resident bursts (4 Mi samples each, 32 MB, so that
            # a 256-thread host holds them in 8 GB); a thread runs its piece several times to get past start-up noise
            n_t = min(n_cpu0, 1 << 22)
            per = n_burst // n_t
            for i in range(cpu_threads):
                b_, o_ = (i // per) % min(W + K, nbuf), (i % per) * n_t
                host_segs.append(bursts[b_, o_:o_ + n_t].cpu().numpy().view(np.complex64).reshape(-1))

    cfg = xa.Demodulator.config(mode, fs_in, D, device=local_rank, costas_chain_len=args.costas_chain,
                                clock_chain_syms=args.clock_chain, front_exact=args.front_exact)
    dem = xa.Demodulator(cfg)
    sps = dem.sps
    cap = int(n_burst / (D * sps * 0.99)) + 64
    soft = torch.empty((cap,), dtype=torch.float32, device=dev)
    soft0 = None

    def step(b):
        return dem.process_device(bursts[b % nbuf].data_ptr(), n_burst, soft.data_ptr(), cap, stream=stream.cuda_stream)

    soft1 = None
    # (the warm-up steps are fed like the timed ones -- the next warm-up burst registered ahead -- so that both sets of the
    # handle's buffers exist before the clock starts; the last warm-up step has nothing registered behind it: the timed region
    # begins on an empty pipeline)
    # inputs registered behind the call in progress: what the library takes for calls of this size (2 for bursts whose clock
    # recovery walks overlapping blocks, else 1)
    depth = max(1, min(args.prefetch_depth, dem.prefetch_depth(n_burst)))
    if not args.no_prefetch:
        for q in range(min(depth, W)):
            dem.prefetch_device(bursts[q % nbuf].data_ptr(), n_burst, stream=stream.cuda_stream)
    for b in range(W):
        if not args.no_prefetch and b + depth < W:
            dem.prefetch_device(bursts[(b + depth) % nbuf].data_ptr(), n_burst, stream=stream.cuda_stream)
        ns = step(b)
        if b == 0:
            soft0 = soft[:ns].clone()
        if b == 1:
            soft1 = soft[:ns].clone()
    if W == 0:
        soft0 = None

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    if not args.no_profile:
        dem.profile(2)       # events around the decimating FIR only: an event record is a queue barrier
    # Streaming (round 5): the inputs of bursts b + 1 and b + 2 are registered while burst b is processed; front end, Costas loop
    # and the clock recovery's walkers of those bursts run ahead on the handle's own streams (the walkers of a burst wait for no
    # other burst: csrc/clock_overlap.h) -- every timed step still pays one front end and one set of loops, all inside the timed
    # region (nothing is registered before the clock starts: the first timed steps fill the pipeline).
    prefetch = not args.no_prefetch

    def ahead(b):
        dem.prefetch_device(bursts[b % nbuf].data_ptr(), n_burst, stream=stream.cuda_stream)

    barrier()
    t0 = time.perf_counter()
    nsym_total = 0
    if prefetch:
        for q in range(min(depth, K)):
            ahead(W + q)
    for b in range(W, W + K):
        if prefetch and b + depth < W + K:
            ahead(b + depth)
        nsym_total += step(b)
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    st = dem.stats()
    prof_timed = dem.profile_read() if not args.no_profile else []
    dem.profile(False)
    prof = []
    dec_samples = {}
    if detail:
        # per-kernel table: the next K bursts of the stream with every launch bracketed (outside the timed region).
        # As many of them as the buffers hold are generated BEFORE the steps run: in round 2 every step of this pass
        # followed the 39 ms FP64 generator of its own burst, and the decimator -- the first kernel behind it -- then
        # read 0.79 ms where every other run of it reads 0.55 (profiles/r3_decimator_ab.txt reproduces both).
        dem.profile(1)
        b = W + K
        while b < W + 2 * K:
            chunk = min(nbuf - 1 if nbuf > 1 else 1, W + 2 * K - b)
            for q in range(chunk):
                generate(b + q)
            torch.cuda.synchronize(dev)
            for q in range(chunk):
                step(b + q)
            b += chunk
        torch.cuda.synchronize(dev)
        prof = dem.profile_read()
        for nm in ("fir_decim", "fir_rrc", "clock_pass", "costas_pass", "clock_overlap", "clock_relay"):
            v = sorted(dem.profile_samples(nm))
            if v:
                dec_samples[nm] = {"min_ms": round(v[0], 4), "median_ms": round(v[len(v) // 2], 4), "max_ms": round(v[-1], 4),
                                   "launches": len(v)}
        dem.profile(False)
        timed = {n: (ms, c) for n, ms, c in prof_timed}
        alone = {n: ms / c for n, ms, c in prof}          # every kernel by itself (the detail pass does not prefetch)
        prof = [(n, timed[n][0], timed[n][1]) if n in timed else (n, ms, c) for n, ms, c in prof]

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=rdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        s = torch.tensor([float(nsym_total)], dtype=torch.float64, device=rdev)
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        nsym_all = float(s.item())
    else:
        nsym_all = float(nsym_total)

    total_samples = float(n_burst) * K * world
    value = total_samples / elapsed / 1e6
    b_alg = 8.0 + 4.0 / (D * sps)          # read cf32 once + one f32 soft symbol per symbol

    # ---- roofline (HIP events recorded by the library on the launch stream, around every launch)
    # Algorithmic bytes each kernel has to move per INPUT sample of the burst (cf32 = 8 B; circuit-rate
    # samples are 1/D of the input samples, symbols 1/(D*sps)); DESIGN.md section 5 derives them.
    c8 = 8.0 / D
    own_bytes = {
        "fir_decim": 8.0 + c8,            # read the input once, write the decimated stream
        "agc_apply": 2 * c8, "fir_rrc": 2 * c8,      # the AGC reduce sweep lives in fir_decim's epilogue
        "costas_pass": c8, "costas_final": 2 * c8,      # the guesses read per-chain statistics only (fused upstream)
        "clock_pass_jac": c8, "clock_pass": c8, "clock_output": c8 + 4.0 / (D * sps),
        # (one launch bracket = the call's relay passes, three by default: each reads the stream and leaves a soft symbol and
        # a record word per symbol)
        "clock_relay": 3 * (c8 + 8.0 / (D * sps)),
        # (round 5: ONE launch of overlapping exactly walked blocks: algorithmically one read of the stream and one soft symbol
        # per symbol; the walkers read every sample 1 + history / range times)
        "clock_overlap": c8 + 4.0 / (D * sps),
    }
    roofline = None
    kernels = {}
    alone = locals().get("alone", {})
    chain_gbs = b_alg * n_burst * K / elapsed / 1e9
    if prof:
        for name, ms, cnt in prof:
            avg = ms / cnt
            k = {"total_ms": round(ms, 4), "launches": cnt, "avg_launch_ms": round(avg, 4),
                 "measured": ("timed region" + (", under the loops of the previous burst (second stream)" if prefetch else ""))
                 if name == "fir_decim" else "detail pass"}
            if name in dec_samples:
                k["alone"] = dec_samples[name]       # detail pass: one burst at a time, every launch bracketed
            if name == "fir_decim" and name in alone:
                k["avg_launch_ms_alone"] = round(alone[name], 4)
                if name in own_bytes:
                    k["hbm_frac_alone"] = round(own_bytes[name] * n_burst / (alone[name] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            if name in own_bytes:
                gbs = own_bytes[name] * n_burst / (avg * 1e-3) / 1e9
                k["algorithmic_bytes_per_launch"] = own_bytes[name] * n_burst
                k["achieved_gbs"] = round(gbs, 1)
                k["hbm_frac"] = round(gbs / HBM_PEAK_GBS, 4)
            kernels[name] = k
        # roofline.achieved is the CHAIN's algorithmic traffic -- (8 + 4/(D*sps)) bytes per input
        # sample x samples per step / step time -- against the HBM peak; that is the figure BASELINE.json's 0.40
        # refers to.  The kernel that moves those bytes (the decimating FIR reads every input sample once; without a
        # decimator the longest single launch) is listed beside it, priced with ITS OWN algorithmic bytes and timed
        # with HIP events on the launch stream inside the timed region.
        in_name = "fir_decim" if "fir_decim" in kernels else max(kernels, key=lambda n: kernels[n]["avg_launch_ms"])
        dom_name = max(prof, key=lambda r: r[1])[0]          # the kernel the step spends most of its time in
        fd = kernels[dom_name]
        fi = kernels[in_name]
        roofline = {"bound": "hbm", "achieved": round(chain_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(chain_gbs / HBM_PEAK_GBS, 4), "traffic": None,
                    "what": "whole chain: algorithmic_bytes_per_sample x samples per step / ms_per_step",
                    "algorithmic_bytes_per_step": b_alg * n_burst,
                    "dominant_kernel": {"kernel": dom_name, "avg_launch_ms": fd["avg_launch_ms"],
                                        "launches_per_step": fd["launches"] / K,
                                        "algorithmic_bytes_per_launch": fd.get("algorithmic_bytes_per_launch"),
                                        "achieved": fd.get("achieved_gbs"), "frac": fd.get("hbm_frac"),
                                        "measured": fd.get("measured"),
                                        "avg_launch_ms_alone": fd.get("avg_launch_ms_alone"),
                                        "frac_alone": fd.get("hbm_frac_alone")},
                    "dominant_kernel_frac": fd.get("hbm_frac"),
                    # the kernel that moves the algorithmic bytes (reads every input sample once), priced on its own bytes
                    "input_kernel": {"kernel": in_name, "avg_launch_ms": fi["avg_launch_ms"],
                                     "algorithmic_bytes_per_launch": fi.get("algorithmic_bytes_per_launch"),
                                     "achieved": fi.get("achieved_gbs"), "frac": fi.get("hbm_frac"),
                                     "measured": fi.get("measured"), "avg_launch_ms_alone": fi.get("avg_launch_ms_alone"),
                                     "frac_alone": fi.get("hbm_frac_alone")}}
        tot = max(prof, key=lambda r: r[1])
        roofline["by_total_time"] = {"kernel": tot[0], "total_ms_per_step": round(tot[1] / K, 4),
                                     "achieved": kernels[tot[0]].get("achieved_gbs"),
                                     "frac": kernels[tot[0]].get("hbm_frac"),
                                     "algorithmic_bytes_per_launch": kernels[tot[0]].get("algorithmic_bytes_per_launch")}
        tpath = os.path.join(HERE, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath) and mode == "lrit" and D == 5:      # the committed PMC passes are C2's
            try:
                tj = json.load(open(tpath))
                if tj.get("_step", {}).get("burst_log2") == args.burst_log2:
                    roofline["traffic"] = tj["_step"]["hbm_bytes_per_step"]
                    roofline["traffic_measured_in_this_run"] = False
                    roofline["traffic_source"] = "NOT measured in this run: read from the committed profiles/hbm_traffic.json <- " + str(tj["_step"].get("source"))
                for nm_, key_ in ((dom_name, "dominant_kernel"), (in_name, "input_kernel")):
                    ent = tj.get({"clock_relay": "clock_relay_pass"}.get(nm_, nm_))      # (the relay's entry is per pass)
                    if ent and ent.get("burst_log2") == args.burst_log2:
                        roofline[key_]["traffic"] = ent["hbm_bytes_per_launch"]
            except Exception:
                pass
    # secondary figures of SURVEY.md 8(d): FP32 rate of the arithmetic the chain has to do, and what a plain
    # read-only sweep reaches on this very device
    flops_per_sample = 4.0 * ((dem.decimator_ntaps if D > 1 else 0) + 63) / D + 30.0 / D
    fp32_tflops = flops_per_sample * n_burst * K / elapsed / 1e12
    read_bw = None
    try:
        read_bw = _capi.device_read_bandwidth(bursts[0].data_ptr(), n_burst * 8, reps=5, device=local_rank,
                                              stream=stream.cuda_stream)
    except Exception:
        pass

    out = {
        "metric": "Msamples/s in -> soft-symbols/s out (LRIT 293 ksym/s chain); % HBM roofline",
        "value": round(value, 2), "unit": "Msamples/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": round(elapsed / K * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s: %s BPSK chain (%sAGC -> RRC 63 a=%.1f -> Costas -> M&M), "
                               "%d Mi-sample cf32 burst per step per GPU, consecutive bursts of one stream"
                               % ({("lrit", 5): "C2", ("lrit", 32): "C5", ("lrit", 1): "C1 chain", ("hrit", 1): "C3"}.get((mode, D), "custom"),
                                  mode.upper(), ("decimating LPF %d taps d=%d -> " % (dem.decimator_ntaps, D)) if D > 1 else "",
                                  alpha, n_burst >> 20),
                   "samples_per_step_per_gpu": n_burst, "decimation": D, "input_rate_sps": fs_in, "sps": round(float(sps), 6),
                   "front_exact": int(args.front_exact),
                   "segments": world, "bursts_reused": bool(W + K > nbuf),
                   "clock_recovery": ("cfg.clock_exact = 0 (default): %d overlapping exactly walked blocks, every walker started from the timing guess one history in front of its range (csrc/clock_overlap.h)" % int(st.clock_relay_segments))
                                     if int(st.clock_relay_passes) == 1 and int(st.clock_passes) == 0 else
                                     ("cfg.clock_exact = 0 (default): %d relay passes over %d exactly walked segments (last step: %d hand-off passes)" % (int(st.clock_relay_passes), int(st.clock_relay_segments), int(st.clock_passes))),
                   "front_end_of_next_burst_overlaps_loops": bool(prefetch),
                   "inputs_registered_ahead": (depth if prefetch else 0),
                   "ahead_of_a_call": "front end, Costas loop and clock-recovery walkers of the next two bursts" if prefetch and depth >= 2 else ("front end and Costas loop of the next burst" if prefetch else "nothing"), "costas_chain_len": args.costas_chain or 256,
                   "clock_chain_syms": args.clock_chain or "auto: whole generations of resident waves (112 at C2), 64..256"},
        "soft_symbols_per_s": round(nsym_all / elapsed, 1),
        "algorithmic_bytes_per_sample": round(b_alg, 4),
        "fp32_tflops": round(fp32_tflops, 2),
        "fp32_flops_per_sample": round(flops_per_sample, 1),
        "read_bw_measured_gbs": None if read_bw is None else round(read_bw, 1),
        "loop_passes": {"costas": st.costas_passes, "clock": st.clock_passes, "clock_relay": st.clock_relay_passes,
                        "clock_relay_closed": st.clock_relay_closed, "clock_relay_segments": int(st.clock_relay_segments),
                        "costas_unconverged": st.costas_unconverged, "clock_open_large": st.clock_open_large,
                        "clock_boundaries_at_the_floor": st.clock_unconverged},
    }
    if roofline:
        out["roofline"] = roofline
        out["kernels"] = kernels

    # ---- two other configurations beside it, the same stream from its first burst through a second handle each, timed over
    # their own steady-state steps: cfg.clock_exact = 1 (csrc/clock_relay.h: relayed to closure -- bit for bit the serial
    # trajectory) and -2 (hand-off passes only).  Not `value`: the headline is the default configuration (no hand-off
    # passes, three relay passes from the timing guess).
    soft0_alt, soft1_alt = {}, {}
    if rank == 0 and world == 1 and not args.no_exact:
        def cfg_of(**kw):
            return xa.Demodulator.config(mode, fs_in, D, device=local_rank, costas_chain_len=args.costas_chain,
                                         clock_chain_syms=args.clock_chain, **kw)
        if args.legs_in_process:
            for key in LEGS:
                out[key], s0_, s1_ = run_leg(key, xa, torch, cfg_of, bursts, nbuf, n_burst, soft, cap, stream, dev, K, prefetch, generate)
                out[key]["process"] = "a further handle of the bench's process"
                if s0_ is not None:
                    soft0_alt[key] = s0_
                if s1_ is not None:
                    soft1_alt[key] = s1_
        else:
            # a fresh process per leg (round 6): this process keeps its handle and its bursts, both idle meanwhile
            import tempfile
            with tempfile.TemporaryDirectory() as td:
                fwd = ["--burst-log2", str(args.burst_log2), "--decimation", str(D), "--mode", mode, "--steps", str(K),
                       "--costas-chain", str(args.costas_chain), "--clock-chain", str(args.clock_chain)] + (["--no-prefetch"] if args.no_prefetch else [])
                res = run_children([["--leg", key, "--leg-dir", td] + fwd for key in LEGS], 300)
                for key, (rc, _so, se) in zip(LEGS, res):
                    try:
                        out[key] = json.load(open(os.path.join(td, key + ".json")))
                    except Exception:
                        out[key] = {"error": "leg process failed (rc %d): %s" % (rc, se[-300:])}
                        continue
                    for nm, dst in (("_soft0.npy", soft0_alt), ("_soft1.npy", soft1_alt)):
                        f_ = os.path.join(td, key + nm)
                        if os.path.exists(f_):
                            dst[key] = torch.from_numpy(np.load(f_))

    # ---- BASELINE's other single-GPU configurations, a short leg each (round 6): a fresh process of this script per
    # configuration -- C5 (configs[4], the "HBM-bound roofline point": 40 Msps, 963 taps, d = 32), C1's chain at burst size
    # (configs[0]'s chain, 2^28 samples at the circuit rate), C3 (configs[2], HRIT) -- the same timed region as `value` with
    # fewer steps; parity against the oracle on the first 2^25 samples of the cold-started burst.  Never `value`.
    if (rank == 0 and world == 1 and not args.no_other_configs and (mode, D) == ("lrit", 5) and args.burst_log2 == 28
            and args.front_exact == 0):
        cfgs = {"C5": ["--decimation", "32", "--steps", "8", "--warmup", "4"],
                "C1": ["--decimation", "1", "--steps", "5", "--warmup", "3"],
                "C3": ["--mode", "hrit", "--decimation", "1", "--steps", "5", "--warmup", "3"],
                # ... and in the parity mode (cfg.front_exact = 2), the only one that holds HRIT to 1e-4
                "C5, parity mode": ["--decimation", "32", "--steps", "5", "--warmup", "3", "--front-exact", "2"],
                "C1, parity mode": ["--decimation", "1", "--steps", "4", "--warmup", "3", "--front-exact", "2"],
                "C3, parity mode": ["--mode", "hrit", "--decimation", "1", "--steps", "4", "--warmup", "3", "--front-exact", "2"]}
        common = ["--no-exact", "--no-other-configs", "--no-serial-floor", "--no-profile", "--cpu-sample-log2", "25", "--cpu-threads", "1"]
        res = run_children([v + common for v in cfgs.values()], 240)
        out["other_configs"] = {}
        for key, (rc, so_, se_) in zip(cfgs, res):
            try:
                d_ = json.loads(so_.strip().splitlines()[-1])
                pv = d_.get("parity_vs_oracle") or {}
                out["other_configs"][key] = {
                    "workload": d_["config"]["workload"], "front_exact": d_["config"].get("front_exact", 0), "steps": d_["steps"], "warmup": d_["warmup"],
                    "ms_per_step": d_["ms_per_step"], "value": d_["value"], "unit": d_["unit"],
                    "roofline": {"bound": "hbm", "what": "whole chain: algorithmic bytes per step / ms_per_step",
                                 "achieved": round(d_["algorithmic_bytes_per_sample"] * d_["config"]["samples_per_step_per_gpu"] / d_["ms_per_step"] / 1e6, 1),
                                 "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": round(d_["algorithmic_bytes_per_sample"] * d_["config"]["samples_per_step_per_gpu"] / d_["ms_per_step"] / 1e6 / HBM_PEAK_GBS, 4)},
                    "parity_vs_oracle": {k_: pv.get(k_) for k_ in ("symbols", "rms", "max", "sign_mismatches")},
                    "cpu_baseline": {k_: (d_.get("cpu_baseline") or {}).get(k_) for k_ in ("value", "unit", "cores", "kind")},
                    "process": "a process of its own"}
            except Exception:
                out["other_configs"][key] = {"error": "rc %d: %s" % (rc, se_[-300:])}

    # ---- CPU baseline: the oracle (a CPU restatement; the reference binary cannot be built here) on a
    # bounded sample of the same workload, one thread like the reference's DSP thread.
    if rank == 0 and world == 1 and not args.no_cpu:
        import oracle
        n_cpu = len(host0)
        host = host0
        od = oracle.Demod(oracle.config(mode, fs_in, D))
        c0 = time.perf_counter()
        so = od.process(host)
        c1 = time.perf_counter()
        out["cpu_baseline"] = {"value": round(n_cpu / (c1 - c0) / 1e6, 3), "unit": "Msamples/s", "cores": 1,
                               "kind": "port",
                               "sample": "first %d samples (%.0f Mi) of burst 0, oracle/xrit_oracle.c single thread "
                                         "(gcc -O3 -mavx2 -mfma -ffp-contract=off; the FMA only where the C library's sincosf has it)" % (n_cpu, n_cpu / 2.0 ** 20),
                               "seconds": round(c1 - c0, 3)}
        if cpu_threads > 1:
            # N independent segments on N threads (the C call releases the GIL): what the node's cores do together
            import threading
            T = cpu_threads
            n_t = len(host_segs[0])
            reps = 8
            dems = [oracle.Demod(oracle.config(mode, fs_in, D)) for _ in range(T)]

            def work(i):
                for _ in range(reps):
                    dems[i].process(host_segs[i])

            th = [threading.Thread(target=work, args=(i,)) for i in range(T)]
            m0 = time.perf_counter()
            for t_ in th:
                t_.start()
            for t_ in th:
                t_.join()
            m1 = time.perf_counter()
            out["cpu_baseline"]["all_threads"] = {"value": round(T * reps * n_t / (m1 - m0) / 1e6, 3), "unit": "Msamples/s",
                                                  "cores": T, "host_logical_cpus": os.cpu_count(),
                                                  "sample": "%d threads, one %d Mi-sample segment of the stream each, "
                                                            "%d passes per thread" % (T, n_t >> 20, reps),
                                                  "seconds": round(m1 - m0, 3)}
        if soft0 is not None:
            def compare(g, ref):
                n = min(len(g), len(ref))
                e = np.abs(g[:n] - ref[:n])
                big = np.abs(ref[:n]) > 1e-3
                return {"symbols": int(n), "rms": float(np.sqrt(np.mean(e ** 2))), "max": float(e.max()),
                        "sign_mismatches": int((np.sign(g[:n])[big] != np.sign(ref[:n])[big]).sum()),
                        # same-arm symbols differ by the Costas-level 1e-6; one step of the 128-arm interpolator
                        # moves a symbol by up to ~3e-3
                        "arm_flips": int((e > 3e-5).sum())}
            g = soft0[:len(so)].cpu().numpy()
            out["parity_vs_oracle"] = compare(g, so)
            out["parity_vs_oracle"]["arm_flips_definition"] = "symbols whose soft value differs by more than 3e-5"
            if not args.no_serial_floor:
                # the floor: the same device chain with the clock recovery as ONE serial trajectory (no hand-offs)
                # on the same samples -- what any float32 M&M fed by this chain's own Costas output differs from the
                # CPU chain by
                sd = xa.Demodulator(xa.Demodulator.config(mode, fs_in, D, device=local_rank, clock_serial=1))
                ser = sd.process(host)
                fl = compare(ser, so)
                out["parity_vs_oracle"]["serial_gpu_rms"] = fl["rms"]
                out["parity_vs_oracle"]["serial_gpu"] = fl
                out["parity_vs_oracle"]["tiled_vs_serial_gpu_rms"] = compare(g, ser)["rms"]
                for key, sa in soft0_alt.items():
                    ge = sa[:len(so)].cpu().numpy()
                    ex = compare(ge, so)
                    ex["vs_serial_gpu_rms"] = compare(ge, ser)["rms"]
                    ex["words_differing_from_serial_gpu"] = int((ge[:min(len(ge), len(ser))].view(np.uint32) !=
                                                                 ser[:min(len(ge), len(ser))].view(np.uint32)).sum())
                    out["parity_vs_oracle"][key] = ex
                # a steady-state burst: burst 1 of the same stream (the oracle and the serial device chain go on from where the
                # samples above ended: the rest of burst 0, then burst 1)
                steady = None
                if host1 is not None and soft1 is not None:
                    if host0_rest is not None:
                        od.process(host0_rest)
                        sd.process(host0_rest)
                    so1, ser1 = od.process(host1), sd.process(host1)
                    g1 = soft1.cpu().numpy()
                    if len(g1) == len(so1) == len(ser1):
                        steady = {"burst": 1, **compare(g1, so1), "serial_gpu_rms": compare(ser1, so1)["rms"],
                                  "vs_serial_gpu_rms": compare(g1, ser1)["rms"]}
                    else:
                        steady = {"burst": 1, "symbol_count": [len(g1), len(so1), len(ser1)]}
                    for key, sa in soft1_alt.items():
                        ga = sa.cpu().numpy()
                        if steady is not None and len(ga) == len(so1):
                            steady[key] = {**compare(ga, so1), "vs_serial_gpu_rms": compare(ga, ser1)["rms"]}
                    out["parity_vs_oracle"]["steady_state"] = steady
                out["parity_vs_oracle"]["target_rms"] = 1e-4
                out["parity_vs_oracle"]["target_met"] = {
                    "default (value), steady-state burst": None if not steady or "rms" not in steady else bool(steady["rms"] <= 1e-4),
                    **{k + ", steady-state burst": bool(steady[k]["rms"] <= 1e-4) for k in soft1_alt if steady and k in steady},
                    "serial_gpu (the floor), steady-state burst": None if not steady or "rms" not in steady else bool(steady["serial_gpu_rms"] <= 1e-4),
                    "default (value), burst 0 (cold start)": bool(out["parity_vs_oracle"]["rms"] <= 1e-4),
                    **{k + ", burst 0": bool(out["parity_vs_oracle"][k]["rms"] <= 1e-4) for k in soft0_alt if k in out["parity_vs_oracle"]},
                    "serial_gpu (the floor), burst 0": bool(fl["rms"] <= 1e-4),
                    "note": "the serial trajectory (cfg.clock_serial, bit-identical to the CPU recurrence on identical input) is the "
                            "floor of any float32 M&M on this chain's Costas output, which differs from the oracle's by 1e-6; where "
                            "the floor itself is above 1e-4 no configuration can meet the target (DESIGN.md section 6)"}
                out["parity_vs_oracle"]["floor_note"] = ("serial_gpu_rms is the measured floor of a hand-off-free float32 "
                                                         "M&M on this chain's Costas output; the time-tiled evaluation "
                                                         "adds the rest")
    # ---- BASELINE config 4 names "RCCL edge-sample exchange": with N > 1 the driver's command also times ONE stream cut in N
    # slices through xrit_group_process_slice_device (a second timed region, LAST: `value` above is the independent-segments
    # figure and is complete by now).  A watchdog keeps a stuck collective from taking the headline with it: on expiry rank 0
    # prints the line it has, with the reason, and every rank leaves.
    if (world > 1 and not share) or (args.contiguous_leg and world == 1):
        import threading

        def give_up():
            if rank == 0:
                out["contiguous"] = {"error": "the leg did not finish within %d s (a collective is stuck?)" % args.contiguous_timeout}
                print(json.dumps(out), flush=True)
            os._exit(0)

        dog = threading.Timer(args.contiguous_timeout, give_up)
        dog.daemon = True
        dog.start()
        try:
            out["contiguous"] = run_contiguous(torch, dist, xa, _capi, world, rank, local_rank, dev, n_burst, D, fs_in, min(K, 8), 2,
                                               slices=bursts)
        except Exception as e:      # (a failure of one rank reaches every rank through the group's all-gathers)
            out["contiguous"] = {"error": "%s: %s" % (type(e).__name__, e)}
        dog.cancel()
    elif world > 1:
        out["contiguous"] = {"skipped": "the ranks share a device and meet over gloo (XRIT_BENCH_SHARE_DEVICE): RCCL refuses two ranks on one GPU"}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
