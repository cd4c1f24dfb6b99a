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
    from xritdemod_amd import dist as xd, synth
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
    from xritdemod_amd import dist as xd
    seeds = [xd.segment_seed(r) for r in range(8)]
    used = set()
    for s in seeds:
        assert s not in used and s + 1 not in used
        used.update((s, s + 1))
