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


# ---------------------------------------------------------------------------------------------------------
# Contiguous stream split across ranks: rank g holds input samples [g*N, (g+1)*N) of ONE
# stream.  The only data that crosses ranks is
#   1. the halo: the last H input samples of rank g-1, sent to rank g (H*8 bytes, ~1-4 MB: xGMI point to point),
#      which rank g demodulates first, from a cold start, so that its filters are primed and its loops locked
#      when the slice proper begins (the loops forget: Costas tau ~1e3 samples, M&M tau ~1.8e3 symbols);
#   2. the last TAIL soft symbols of rank g-1, against which rank g settles the two things a cold-started loop
#      cannot know: the Costas pi ambiguity (polarity) and whether the symbol that straddles the slice boundary
#      was emitted on this side or the other (the chain emits a symbol while its read index is < end-24 and
#      carries the rest, so both ranks apply the same rule at the same stream position and normally agree);
#   3. one all-gather of (relative polarity, symbol count): prefix product / prefix sum give every rank its
#      absolute polarity and output offset.
# No data-path collective beyond these.  The steps are separate functions so that the same code runs under
# torch.distributed (gloo on CPU with the engine of the caller's choice, nccl = RCCL on GPUs) and, for
# single-GPU tests, with the ranks played one after the other.

TAIL = 256          # soft symbols compared across a boundary
KEEP = 8            # halo symbols kept in front of the slice output, for the straddling symbol


def halo_samples(decimation, sps, lpf_taps, warm_symbols=24576, rrc_taps=63):
    """Input samples rank g needs from rank g-1: FIR histories + loop warm-up (M&M is the slow one: ~13 time
    constants of 1800 symbols for 1e-4-level agreement with the uninterrupted stream) + interpolator look-ahead."""
    circuit = (rrc_taps - 1) + int(warm_symbols * sps) + 8 + 24
    return (lpf_taps - 1 if decimation > 1 else 0) + decimation * circuit


def split_process(process, halo, body):
    """Steps 1b/2 on one rank: demodulate the halo (cold start), then the slice.  `process` is the stateful chain
    call (array of complex samples -> array of soft symbols).  Returns (kept halo symbols, slice symbols)."""
    import numpy as np
    if halo is None:
        return np.zeros(0, np.float32), process(body)
    h = process(halo)
    return h[-(TAIL + KEEP):].copy(), process(body)


def split_align(prev_tail, halo_syms, syms):
    """Step 2 on rank g > 0.  prev_tail: the last TAIL symbols rank g-1 produced (its polarity); halo_syms / syms:
    what this rank produced over the halo / its slice.  Returns (relative polarity +-1, aligned slice symbols in
    this rank's own polarity).  lag = how many symbols this rank's split point lies after rank g-1's."""
    import numpy as np
    if len(prev_tail) == 0:
        return 1, syms
    seq = np.concatenate([halo_syms, syms[:KEEP]])
    nh = len(halo_syms)
    best = (0.0, 0, 1)
    m = len(prev_tail)
    for lag in range(-KEEP + 1, KEEP):
        end = nh + lag                 # prev_tail's last symbol would be seq[end - 1]
        if end - m < 0 or end > len(seq):
            continue
        c = float(np.dot(prev_tail, seq[end - m:end]))
        if abs(c) > best[0]:
            best = (abs(c), lag, 1 if c >= 0 else -1)
    _, lag, pol = best
    if lag < 0:        # rank g-1 stopped earlier: the symbols in between were only emitted here, over the halo
        out = np.concatenate([halo_syms[nh + lag:], syms])
    elif lag > 0:      # rank g-1 already emitted the first `lag` symbols of this slice
        out = syms[lag:]
    else:
        out = syms
    return pol, out


def split_finish(all_pol, all_count, rank):
    """Step 3: absolute polarity (product of the relative ones up to this rank) and output offset."""
    pol = 1
    for p in all_pol[1:rank + 1]:
        pol *= p
    return pol, int(sum(all_count[:rank]))


def demodulate_contiguous(make_process, body, dist, rank, world, halo_len, same_lock=False, to_comm=None, from_comm=None):
    """The whole exchange under torch.distributed.  make_process() returns a fresh, cold chain call (array of
    complex samples -> array of soft symbols, stateful).  body: this rank's slice as a complex64 array (host) --
    for device-resident slices pass to_comm/from_comm that wrap device tensors, the calls are the same.
    Returns (soft symbols of this rank in the stream's polarity, offset of the first one in the stream's output).

    same_lock=False: a rank that locked pi away from the stream's polarity has its symbols negated.  That is the
    right hard decision for every symbol, but not the trajectory the uninterrupted chain follows: the M&M detector
    slices to {0,1}, not {-1,+1}, so the two locks are different (equally valid) loops and the soft symbols differ
    at the 1e-3 level.  same_lock=True: such a rank demodulates its halo and slice once more with the input
    negated (= the other lock).  It then tracks the uninterrupted trajectory to the level at which the M&M
    recurrence is chaotic anyway (~1e-4 rms); with the CPU oracle the two trajectories
    often merge bit for bit after 1e5-2e5 symbols."""
    import numpy as np
    import torch
    to_comm = to_comm or (lambda a: torch.from_numpy(np.ascontiguousarray(a)))
    from_comm = from_comm or (lambda t: t.numpy())
    body = np.ascontiguousarray(body, np.complex64)
    halo = None
    reqs = []
    if rank + 1 < world:
        out_halo = to_comm(body[-halo_len:].view(np.float32))
        reqs.append(dist.isend(out_halo, dst=rank + 1))
    if rank > 0:
        buf = to_comm(np.zeros(2 * halo_len, np.float32))
        dist.recv(buf, src=rank - 1)
        halo = from_comm(buf).view(np.complex64).copy()
    for r in reqs:
        r.wait()
    h_syms, syms = split_process(make_process(), halo, body)
    # tails travel in each rank's own (first-run) polarity; fixed size so that the receive can be posted blind
    tail = np.zeros(TAIL, np.float32)
    k = min(TAIL, len(syms))
    if k:
        tail[TAIL - k:] = syms[-k:]
    reqs = []
    if rank + 1 < world:
        reqs.append(dist.isend(to_comm(tail), dst=rank + 1))
    prev_tail = np.zeros(0, np.float32)
    if rank > 0:
        buf = to_comm(np.zeros(TAIL, np.float32))
        dist.recv(buf, src=rank - 1)
        prev_tail = from_comm(buf).copy()
    for r in reqs:
        r.wait()
    pol_rel, out = split_align(prev_tail, h_syms, syms)

    def gather(a, b):
        if world == 1:
            return [a], [b]
        mine = torch.tensor([a, b], dtype=torch.int64)
        got = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(got, mine)
        return [int(g[0]) for g in got], [int(g[1]) for g in got]

    pols, counts = gather(pol_rel, len(out))
    pol, offset = split_finish(pols, counts, rank)
    if not same_lock:
        return (out * np.float32(pol)).astype(np.float32), offset
    if pol < 0:
        h_syms, syms = split_process(make_process(), -halo, -body)
        _, out = split_align(prev_tail, h_syms, syms)
    _, counts = gather(1, len(out))
    return out.astype(np.float32), int(sum(counts[:rank]))


# ---------------------------------------------------------------------------------------------------------
# Round 6: ONE loop state across the slices -- the twin of csrc/group.hip steps 3b / 3c.  The reference's five objects carry one
# state across every chunk (demodulator.cpp:446-450); a stream cut across ranks keeps that true in two moves:
#   * a rank whose cold start fell into the OTHER Costas lock starts once more from a phase of pi (the detector I*Q does not see
#     a half turn: same pull-in, other lock) and there meets the stream's own float32 trajectory inside its halo;
#   * the clock recovery's carried state (mu, omega, last symbols and decisions, unread samples) travels from every rank to the
#     rank behind it, which walks its slice's clock recovery again from it unless its own warm-up had reached the very same one.
# `make_chain()` returns an object with process(x) -> soft symbols, flip_costas_phase(), clock_start() / clock_carry() (float32
# words: the state the last call started from / the next starts from) and redo_clock_from(words) -> soft symbols of the last
# call again.  On GPUs that object is the product's chain handle (xrit_demod_flip_costas_phase, xrit_demod_export_clock_carry,
# xrit_demod_redo_clock_from -- and the whole exchange is xrit_group_process_slice_device); here the CPU oracle plays it.

class OracleChain:
    """The chain interface of demodulate_contiguous_one_state played by the CPU oracle (tests only)."""

    def __init__(self, mode="lrit", sample_rate=1.25e6, decimation=1):
        import oracle
        self.d = oracle.Demod(oracle.config(mode, sample_rate, decimation))
        self._start = None

    def flip_costas_phase(self):
        import numpy as np
        c = self.d.costas
        c.phase = float(np.float32(c.phase) - np.float32(np.pi)) if c.phase > 0 else float(np.float32(c.phase) + np.float32(np.pi))

    def process(self, x):
        self._start = self.d.clock.export_carry()
        return self.d.process(x)

    def clock_start(self):
        return self._start

    def clock_carry(self):
        return self.d.clock.export_carry()

    def redo_clock_from(self, words):
        import numpy as np
        mm = self.d.clock
        mm.import_carry(words)
        self._start = np.array(words, np.float32)
        return np.ascontiguousarray(mm.Work(self.d.stage("costas")).real, np.float32)


def _align_lag(prev_tail, halo_syms, syms):
    """split_align's search on its own: (relative polarity, lag)."""
    import numpy as np
    seq = np.concatenate([halo_syms, syms[:KEEP]])
    nh, m = len(halo_syms), len(prev_tail)
    best = (0.0, 0, 1)
    for lag in range(-KEEP + 1, KEEP):
        end = nh + lag
        if end - m < 0 or end > len(seq):
            continue
        c = float(np.dot(prev_tail, seq[end - m:end]))
        if abs(c) > best[0]:
            best = (abs(c), lag, 1 if c >= 0 else -1)
    return best[2], best[1]


def demodulate_contiguous_one_state(make_chain, body, dist, rank, world, halo_len):
    """The exchange of csrc/group.hip (round 6) under torch.distributed, host arrays.  Returns (soft symbols of this rank in the
    stream's polarity, offset of the first one in the stream's output, what happened at this rank's boundary: a dict with
    first_lock (+-1), second_start, handed, joined)."""
    import numpy as np
    import torch
    body = np.ascontiguousarray(body, np.complex64)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    halo = None
    reqs = []
    if rank + 1 < world:
        reqs.append(dist.isend(t(body[-halo_len:].view(np.float32)), dst=rank + 1))
    if rank > 0:
        buf = torch.zeros(2 * halo_len, dtype=torch.float32)
        dist.recv(buf, src=rank - 1)
        halo = buf.numpy().view(np.complex64).copy()
    for r in reqs:
        r.wait()

    def run(chain):
        h = chain.process(halo)[-(TAIL + KEEP):].copy() if halo is not None else np.zeros(0, np.float32)
        return h, chain.process(body)

    chain = make_chain()
    h_syms, syms = run(chain)
    # 2b. boundary symbols, in each rank's FIRST polarity
    tail = np.zeros(TAIL, np.float32)
    k = min(TAIL, len(syms))
    if k:
        tail[TAIL - k:] = syms[-k:]
    reqs = []
    if rank + 1 < world:
        reqs.append(dist.isend(t(tail), dst=rank + 1))
    prev_tail = np.zeros(0, np.float32)
    if rank > 0:
        buf = torch.zeros(TAIL, dtype=torch.float32)
        dist.recv(buf, src=rank - 1)
        prev_tail = buf.numpy().copy()
    for r in reqs:
        r.wait()
    pol_rel, lag = _align_lag(prev_tail, h_syms, syms) if rank > 0 else (1, 0)

    def gather(a, b):
        if world == 1:
            return [a], [b]
        got = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(got, torch.tensor([a, b], dtype=torch.int64))
        return [int(g[0]) for g in got], [int(g[1]) for g in got]

    pols, _ = gather(pol_rel, 0)
    pol = 1
    for p in pols[:rank + 1]:
        pol *= p
    info = {"first_lock": pol, "second_start": False, "handed": False, "joined": False}
    # 3b. the other lock: once more, from a Costas phase of pi
    if pol < 0 and rank > 0:
        chain = make_chain()
        chain.flip_costas_phase()
        h_syms, syms = run(chain)
        p2, lag2 = _align_lag(prev_tail, h_syms, syms)
        info["second_start"] = True
        if p2 == -pol_rel:
            lag = lag2
        else:                   # (fell on the same side again: the symbols negated, as before round 6)
            syms, h_syms = -syms, -h_syms
    # 3c. the clock recovery's carried state, from rank to rank: receive, settle, then send what THIS slice ended in
    words = len(chain.clock_carry())
    if rank > 0:
        buf = torch.zeros(words, dtype=torch.float32)
        dist.recv(buf, src=rank - 1)
        theirs = buf.numpy().copy()
        if np.array_equal(theirs.view(np.uint32), chain.clock_start().view(np.uint32)):
            info["joined"] = True
        else:
            syms = chain.redo_clock_from(theirs)
            info["handed"] = True
        lag = 0                 # (a slice that continues from the very state the slice in front ended in has no straddling symbol)
    if rank + 1 < world:
        dist.send(t(chain.clock_carry()), dst=rank + 1)
    if lag < 0:
        out = np.concatenate([h_syms[len(h_syms) + lag:], syms])
    elif lag > 0:
        out = syms[lag:]
    else:
        out = syms
    _, counts = gather(1, len(out))
    return out.astype(np.float32), int(sum(counts[:rank])), info


def demodulate_contiguous_device(make_demod, body_t, dist, rank, world, halo_len, same_lock=False, stream=None):
    """Device-resident form of demodulate_contiguous (what bench.py --contiguous runs, one process per GPU, nccl =
    RCCL): body_t is this rank's slice as a float32 cuda tensor of shape (n, 2); the halo travels GPU to GPU with
    dist.send / dist.recv (over xGMI), the boundary symbols and the (polarity, count) all-gather likewise; only
    TAIL + KEEP symbols per rank are looked at on the host.  make_demod() returns a fresh xritdemod_amd.Demodulator.
    Returns (soft symbols as a cuda tensor view, in the stream's polarity; offset in the stream's output)."""
    import numpy as np
    import torch
    dev = body_t.device
    n = body_t.shape[0]
    st = stream if stream is not None else torch.cuda.current_stream(dev).cuda_stream

    def run(dem, halo_t, body, scale):
        cap = n + 64
        h_syms = np.zeros(0, np.float32)
        if halo_t is not None:
            hs = torch.empty(halo_t.shape[0] + 64, dtype=torch.float32, device=dev)
            src = halo_t if scale > 0 else -halo_t
            k = dem.process_device(src.data_ptr(), halo_t.shape[0], hs.data_ptr(), hs.shape[0], stream=st)
            h_syms = hs[max(0, k - (TAIL + KEEP)):k].cpu().numpy()
        soft = torch.empty(cap, dtype=torch.float32, device=dev)
        src = body if scale > 0 else -body
        k = dem.process_device(src.data_ptr(), n, soft.data_ptr(), cap, stream=st)
        return h_syms, soft, k

    halo_t = None
    reqs = []
    if rank + 1 < world:
        reqs.append(dist.isend(body_t[n - halo_len:].contiguous(), dst=rank + 1))
    if rank > 0:
        halo_t = torch.empty((halo_len, 2), dtype=torch.float32, device=dev)
        dist.recv(halo_t, src=rank - 1)
    for r in reqs:
        r.wait()
    h_syms, soft, k = run(make_demod(), halo_t, body_t, +1)
    tail_t = torch.zeros(TAIL, dtype=torch.float32, device=dev)
    m = min(TAIL, k)
    if m:
        tail_t[TAIL - m:] = soft[k - m:k]
    reqs = []
    if rank + 1 < world:
        reqs.append(dist.isend(tail_t, dst=rank + 1))
    prev_tail = np.zeros(0, np.float32)
    if rank > 0:
        buf = torch.empty(TAIL, dtype=torch.float32, device=dev)
        dist.recv(buf, src=rank - 1)
        prev_tail = buf.cpu().numpy()
    for r in reqs:
        r.wait()

    def align(h_syms, soft, k):
        # split_align on the few symbols that matter; the bulk stays on the device
        head = soft[:min(KEEP, k)].cpu().numpy()
        pol_rel, out_head = split_align(prev_tail, h_syms, head)
        lag = len(head) - len(out_head)            # > 0: drop, < 0: prepend halo symbols
        if lag >= 0:
            return pol_rel, soft[lag:k]
        pre = torch.from_numpy(np.ascontiguousarray(h_syms[len(h_syms) + lag:])).to(dev)
        return pol_rel, torch.cat([pre, soft[:k]])

    def gather(a, b):
        if world == 1:
            return [a], [b]
        mine = torch.tensor([a, b], dtype=torch.int64, device=dev)
        got = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(got, mine)
        return [int(g[0]) for g in got], [int(g[1]) for g in got]

    pol_rel, out = align(h_syms, soft, k)
    pols, counts = gather(pol_rel, int(out.shape[0]))
    pol, offset = split_finish(pols, counts, rank)
    if not same_lock:
        return (out if pol > 0 else -out), offset
    if pol < 0:
        h_syms, soft, k = run(make_demod(), halo_t, body_t, -1)
        _, out = align(h_syms, soft, k)
    _, counts = gather(1, int(out.shape[0]))
    return out, int(sum(counts[:rank]))
