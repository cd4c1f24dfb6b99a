"""Synthetic xRIT IQ bursts: BPSK + RRC pulse + carrier /
timing offsets + AWGN, generated from a counter-based hash so that any time
slice can be produced independently (each GPU rank generates only its slice).

This numpy version is the specification; the HIP generator in
csrc/synth.hip implements the same formulae for the large bench bursts.
The reference has no signal source of its own besides a cf32 file reader
(/root/reference/demodulator/src/CFileFrontend.cpp:34-56), so this stands in
for the capture file of demodulator/xritdemod.cfg:15.
"""
from dataclasses import dataclass

import numpy as np

MASK64 = np.uint64(0xFFFFFFFFFFFFFFFF)
GOLD = np.uint64(0x9E3779B97F4A7C15)
M1 = np.uint64(0xBF58476D1CE4E5B9)
M2 = np.uint64(0x94D049BB133111EB)

DEFAULT_SEED = 0x58524954  # "XRIT"
SPAN = 16                  # pulse half-span in symbols


@dataclass
class SynthParams:
    fs_in: float = 1.25e6          # input sample rate (Hz)
    symbol_rate: float = 293883.0  # LRIT, Parameters.h:23
    alpha: float = 0.5             # TX RRC roll-off (same as RX, Parameters.h:24)
    amplitude: float = 0.1
    carrier_hz: float = 500.0
    phase0: float = 0.7
    timing_offset: float = 0.3     # symbols
    clock_ppm: float = 20.0
    esn0_db: float = 12.0
    seed: int = DEFAULT_SEED

    @property
    def sps_in(self):
        return self.fs_in / self.symbol_rate

    @property
    def sigma(self):
        """std-dev of the complex noise sample (total, both components)."""
        if self.esn0_db is None:
            return 0.0
        return float(np.sqrt(self.amplitude ** 2 * self.sps_in / (10.0 ** (self.esn0_db / 10.0))))


def hash64(seed, counter):
    """splitmix64 finaliser of seed + counter*GOLD (all arithmetic mod 2^64)."""
    with np.errstate(over="ignore"):
        z = (np.uint64(seed) + counter.astype(np.uint64) * GOLD) & MASK64
        z = ((z ^ (z >> np.uint64(30))) * M1) & MASK64
        z = ((z ^ (z >> np.uint64(27))) * M2) & MASK64
        z = z ^ (z >> np.uint64(31))
    return z


def symbol_bits(seed, k):
    """Transmitted BPSK symbol k -> +1/-1 (int64 k, may be negative)."""
    h = hash64(seed, np.asarray(k, dtype=np.int64).view(np.uint64))
    return np.where((h >> np.uint64(63)) == 1, 1.0, -1.0)


def rrc_pulse(t, alpha):
    """Unit-energy root-raised-cosine impulse response, t in symbol periods."""
    t = np.asarray(t, dtype=np.float64)
    out = np.empty_like(t)
    a = alpha
    z = np.abs(t) < 1e-9
    s = np.abs(np.abs(4 * a * t) - 1.0) < 1e-7
    g = ~(z | s)
    out[z] = 1.0 - a + 4 * a / np.pi
    out[s] = (a / np.sqrt(2.0)) * ((1 + 2 / np.pi) * np.sin(np.pi / (4 * a))
                                   + (1 - 2 / np.pi) * np.cos(np.pi / (4 * a)))
    tg = t[g]
    out[g] = (np.sin(np.pi * tg * (1 - a)) + 4 * a * tg * np.cos(np.pi * tg * (1 + a))) / (
        np.pi * tg * (1 - (4 * a * tg) ** 2))
    return out


def generate(p: SynthParams, n: int, start: int = 0, chunk: int = 1 << 18, symbols=None):
    """cf32 samples [start, start+n) of the burst described by p.  symbols (optional): array of +1/-1 that
    replaces the pseudo-random symbols k = 0 .. len-1 (a framed, coded bit stream, see ccsds_frames)."""
    out = np.empty(n, dtype=np.complex64)
    rate = p.symbol_rate * (1.0 + p.clock_ppm * 1e-6) / p.fs_in
    sigma = p.sigma
    for c0 in range(0, n, chunk):
        c1 = min(n, c0 + chunk)
        idx = np.arange(start + c0, start + c1, dtype=np.int64)
        u = idx.astype(np.float64) * rate - p.timing_offset
        k0 = np.floor(u).astype(np.int64)
        acc = np.zeros(c1 - c0, dtype=np.float64)
        for j in range(-SPAN + 1, SPAN + 1):
            k = k0 + j
            b = symbol_bits(p.seed, k)
            if symbols is not None:
                inside = (k >= 0) & (k < len(symbols))
                b = np.where(inside, symbols[np.clip(k, 0, len(symbols) - 1)], b)
            acc += b * rrc_pulse(u - k, p.alpha)
        ph = (2.0 * np.pi * p.carrier_hz / p.fs_in) * idx.astype(np.float64) + p.phase0
        sig = p.amplitude * acc * np.exp(1j * ph)
        if sigma > 0:
            h = hash64(p.seed + 1, idx.view(np.uint64))
            u1 = ((h >> np.uint64(40)).astype(np.float64) + 0.5) / float(1 << 24)
            u2 = ((h & np.uint64(0xFFFFFF)).astype(np.float64) + 0.5) / float(1 << 24)
            r = sigma * np.sqrt(-np.log(u1))
            sig = sig + r * np.exp(2j * np.pi * u2)
        out[c0:c1] = sig.astype(np.complex64)
    return out


def transmitted_symbols(p: SynthParams, k0: int, n: int):
    return symbol_bits(p.seed, np.arange(k0, k0 + n, dtype=np.int64))


# ---- a framed, convolutionally coded symbol stream (what the reference's decoder locks to) ---------------
CCSDS_ASM = 0x1ACFFC1D          # attached sync marker of a 1024-byte CADU
CODED_FRAME_SYMBOLS = 16384     # 8192 bits, rate 1/2 (decoder/src/parameters.h:28-31)


def conv_encode_k7(bits):
    """Rate-1/2, k=7 convolutional code, generators 0x4F / 0x6D on a register that takes the new bit at its low end,
    G1 symbol first, start state 0.  With this convention the coded sync marker is the reference decoder's
    LRIT_UW2 (newdecoder.cpp:24) and its complement LRIT_UW0 (the 180-degree lock)."""
    bits = np.asarray(bits, dtype=np.uint8)
    padded = np.concatenate([np.zeros(6, np.uint8), bits])
    n = len(bits)
    out = np.zeros(2 * n, dtype=np.uint8)
    for tap in range(7):                       # register bit `tap` holds the bit that arrived `tap` steps ago
        d = padded[6 - tap:6 - tap + n]
        if (0x4F >> tap) & 1:
            out[0::2] ^= d
        if (0x6D >> tap) & 1:
            out[1::2] ^= d
    return out


def ccsds_frames(n_frames, seed=1):
    """+1/-1 symbols of n_frames coded frames: sync marker + pseudo-random payload, encoded as one stream.
    Coded bit 0 -> +1, 1 -> -1: with the correlator's hard decision (non-negative soft byte = one) a chain locked
    at 0 degrees then finds the decoder's word 0 (LRIT_UW0, DEG_0) and one locked at 180 degrees its word 1."""
    rng = np.random.default_rng(seed)
    asm = np.array([(CCSDS_ASM >> (31 - i)) & 1 for i in range(32)], dtype=np.uint8)
    frames = [np.concatenate([asm, rng.integers(0, 2, CODED_FRAME_SYMBOLS // 2 - 32).astype(np.uint8)])
              for _ in range(n_frames)]
    coded = conv_encode_k7(np.concatenate(frames))
    return np.where(coded == 1, -1.0, 1.0)
