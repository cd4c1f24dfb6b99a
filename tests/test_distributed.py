"""CPU tier: the N>1 path of bench.py (independent segments per rank, barrier, max-over-ranks
timing, summed units) on two gloo ranks."""
import os
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import dist_twin as xd
    import synth
    d = xd.init("gloo")
    w, r, lr = xd.env_world()
    assert (w, r) == (world, rank)
    # every rank generates only its own segment, from its own seed
    p = synth.SynthParams(seed=xd.segment_seed(rank))
    x = synth.generate(p, 4096)
    d.barrier()
    tmax, units = xd.aggregate(0.5 * (rank + 1), 1000 * (rank + 1))
    q.put((rank, tmax, units, float(np.abs(x).sum()), xd.throughput_msps(1 << 20, 5, world, tmax)))
    d.destroy_process_group()


def test_two_rank_segments_and_timing():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 400)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, t0, u0, s0, v0), (r1, t1, u1, s1, v1) = res
    assert t0 == t1 == 1.0            # max over ranks
    assert u0 == u1 == 3000.0         # sum over ranks
    assert s0 != s1                   # different segments
    assert abs(v0 - (1 << 20) * 5 * 2 / 1.0 / 1e6) < 1e-9 and v0 == v1


def test_segment_seeds_do_not_collide():
    sys.path.insert(0, ROOT)
    import dist_twin as xd
    seeds = [xd.segment_seed(r) for r in range(8)]
    used = set()
    for s in seeds:
        assert s not in used and s + 1 not in used
        used.update((s, s + 1))


# ---- contiguous stream split with edge-sample exchange ------------------------------------------
def _split_worker(rank, world, port, q, same_lock):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import oracle                       # the CPU tier plays the chain with the oracle; on GPUs it is the HIP chain
    import dist_twin as xd
    import synth
    d = xd.init("gloo")
    n = 900000
    # every rank holds only its own slice of ONE stream (the generator is counter based: any slice on its own)
    body = synth.generate(synth.SynthParams(), n, start=rank * n)
    halo_len = xd.halo_samples(1, 4.2534, 0, warm_symbols=24576)      # with this halo rank 1 locks pi away
    make = lambda: oracle.Demod(oracle.config("lrit", 1.25e6, 1)).process
    soft, offset = xd.demodulate_contiguous(make, body, d, rank, world, min(halo_len, n), same_lock=same_lock)
    q.put((rank, offset, soft))
    d.barrier()
    d.destroy_process_group()


def _run_split(same_lock):
    import oracle
    import synth
    world, n = 2, 900000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + (os.getpid() % 90) + (7 if same_lock else 0)
    procs = [ctx.Process(target=_split_worker, args=(r, world, port, q, same_lock)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = oracle.Demod(oracle.config("lrit", 1.25e6, 1)).process(synth.generate(synth.SynthParams(), world * n))
    got = np.concatenate([r[2] for r in res])
    assert res[0][1] == 0 and res[1][1] == len(res[0][2])          # offsets = prefix sums of the counts
    assert len(got) == len(ref)                                     # no symbol lost or doubled at the boundary
    return got, ref, len(res[0][2])


def test_contiguous_split_two_ranks_polarity_and_boundary():
    got, ref, n0 = _run_split(same_lock=False)
    assert np.array_equal(got[:n0], ref[:n0])                       # rank 0 is the uninterrupted stream
    assert np.array_equal(np.sign(got), np.sign(ref))               # hard decisions: every symbol, right polarity
    e = float(np.sqrt(np.mean((got[n0:] - ref[n0:]) ** 2)))
    assert 3e-4 < e < 3e-3                                          # the other lock of the M&M loop: ~1e-3


def test_contiguous_split_two_ranks_same_lock():
    got, ref, n0 = _run_split(same_lock=True)
    assert np.array_equal(got[:n0], ref[:n0])
    assert np.array_equal(np.sign(got), np.sign(ref))
    # rank 1 ran again on the other lock: it now follows the uninterrupted trajectory to the chaos level
    assert float(np.sqrt(np.mean((got[n0:] - ref[n0:]) ** 2))) < 3e-4


# ---- round 6: ONE loop state across the slices (the twin of csrc/group.hip steps 3b / 3c) ----------------------------------------
def _one_state_worker(rank, world, port, q, phases):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import dist_twin as xd
    import synth
    d = xd.init("gloo")
    n = 600000
    halo_len = xd.halo_samples(1, 4.2534, 0, warm_symbols=49152)
    for ph in phases:
        body = synth.generate(synth.SynthParams(phase0=ph, seed=4242), n, start=rank * n)
        soft, offset, info = xd.demodulate_contiguous_one_state(lambda: xd.OracleChain("lrit", 1.25e6, 1), body, d, rank, world, halo_len)
        q.put((ph, rank, offset, soft, info))
    d.barrier()
    d.destroy_process_group()


def test_contiguous_split_carries_one_loop_state_across_ranks():
    """World of two over gloo, the CPU oracle as the chain: whichever Costas lock rank 1's cold start falls into (the capture's
    start phase is moved until both have been seen), the joined symbols are the uninterrupted chain's WORD FOR WORD -- a rank pi
    away starts once more from a phase of pi, the clock recovery starts from the state rank 0 ended in (or had reached it).
    The same protocol as xrit_group_process_slice_device (tests/test_gpu_parity.py runs that one on the GPU)."""
    import oracle
    import synth
    world, n = 2, 600000
    phases = (0.7, 1.5, 2.3, 3.9)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29800 + (os.getpid() % 90)
    procs = [ctx.Process(target=_one_state_worker, args=(r, world, port, q, phases)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world * len(phases)):
        ph, rank, offset, soft, info = q.get(timeout=600)
        res[(ph, rank)] = (offset, soft, info)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    locks = set()
    for ph in phases:
        ref = oracle.Demod(oracle.config("lrit", 1.25e6, 1)).process(synth.generate(synth.SynthParams(phase0=ph, seed=4242), world * n))
        (o0, s0, i0), (o1, s1, i1) = res[(ph, 0)], res[(ph, 1)]
        got = np.concatenate([s0, s1])
        assert o0 == 0 and o1 == len(s0) and len(got) == len(ref), (ph, o0, o1, len(s0), len(s1), len(ref))
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (ph, i1, float(np.sqrt(np.mean((got - ref) ** 2))))
        assert not i0["second_start"] and not i0["handed"] and not i0["joined"]
        assert i1["second_start"] == (i1["first_lock"] < 0) and (i1["handed"] or i1["joined"]), (ph, i1)
        locks.add(i1["first_lock"])
    assert locks == {1, -1}, locks
