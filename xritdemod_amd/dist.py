"""Multi-GPU plumbing for the chain: one process per GPU, independent capture
segments per rank (the path shards by segment with no data-path collective).
Kept apart from bench.py so that the N>1 logic is covered by world_size-2 gloo
tests on CPU (tests/test_distributed.py)."""
import os

BASE_SEED = 0x58524954  # "XRIT"


def env_world():
    return (int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")),
            int(os.environ.get("LOCAL_RANK", "0")))


def segment_seed(rank):
    """Each rank demodulates its own capture segment: distinct bit and noise streams.
    The generator uses seed for the bits and seed+1 for the noise, hence the stride of 2."""
    return BASE_SEED + 2 * rank


def init(backend, device=None):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    kw = {}
    if device is not None and backend == "nccl":
        kw["device_id"] = device
    dist.init_process_group(backend=backend, **kw)
    return dist


def aggregate(elapsed_s, units, device="cpu"):
    """(max over ranks of the elapsed time, sum over ranks of the processed units)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(elapsed_s), float(units)
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    u = torch.tensor([float(units)], dtype=torch.float64, device=device)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), float(u.item())


def throughput_msps(samples_per_rank_per_step, steps, world, elapsed_max_s):
    """Whole-job Msamples/s: every rank's samples over the slowest rank's time."""
    return float(samples_per_rank_per_step) * steps * world / elapsed_max_s / 1e6
