"""ctypes binding of the CPU oracle (oracle/xrit_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package xritdemod_amd.
PARITY UNPINNED (see xrit_oracle.h): the reference has no golden vectors and its
DSP library libSatHelper is absent.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libxrit_oracle.so")

SAMPLE_FLOATIQ, SAMPLE_S16IQ, SAMPLE_S8IQ, SAMPLE_U8IQ = 0, 1, 2, 3


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "xrit_oracle.c")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(src),
                                                   os.path.getmtime(os.path.join(_HERE, "xrit_oracle.h")))):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


class Config(C.Structure):
    _fields_ = [("sample_rate", C.c_float), ("decimation", C.c_uint32), ("symbol_rate", C.c_uint32),
                ("rrc_alpha", C.c_float), ("rrc_taps", C.c_int),
                ("agc_rate", C.c_float), ("agc_reference", C.c_float), ("agc_gain", C.c_float),
                ("agc_max_gain", C.c_float), ("pll_alpha", C.c_float),
                ("clock_mu", C.c_float), ("clock_alpha", C.c_float), ("clock_gain_omega", C.c_float),
                ("clock_omega_limit", C.c_float)]


class Agc(C.Structure):
    _fields_ = [("rate", C.c_float), ("reference", C.c_float), ("gain", C.c_float), ("max_gain", C.c_float)]


class Costas(C.Structure):
    _fields_ = [("phase", C.c_float), ("freq", C.c_float), ("alpha", C.c_float), ("beta", C.c_float),
                ("max_freq", C.c_float), ("min_freq", C.c_float), ("wrap_pi", C.c_int), ("imag_axis", C.c_int)]


class Rtl(C.Structure):
    _fields_ = [("lut", C.c_float * 256), ("alpha", C.c_float), ("iavg", C.c_float), ("qavg", C.c_float)]


class Knobs(C.Structure):
    """xo_knobs: the semantic choices that cannot be checked against the absent libSatHelper (xrit_oracle.h)."""
    _fields_ = [("fir_phase_last", C.c_int), ("mm_fudge", C.c_int), ("mm_drop_tail", C.c_int),
                ("costas_wrap_pi", C.c_int), ("costas_imag_axis", C.c_int)]


class MMState(C.Structure):
    _fields_ = [("mu", C.c_float), ("omega", C.c_float), ("omega_mid", C.c_float), ("omega_lim", C.c_float),
                ("gain_omega", C.c_float), ("gain_mu", C.c_float),
                ("p_2t", C.c_float * 2), ("p_1t", C.c_float * 2), ("p_0t", C.c_float * 2),
                ("c_2t", C.c_float * 2), ("c_1t", C.c_float * 2), ("c_0t", C.c_float * 2),
                ("carry", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        vp, fp, ip = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int)
        L.xo_lowpass_ntaps.restype = C.c_int
        L.xo_lowpass_ntaps.argtypes = [C.c_double, C.c_double]
        L.xo_lowpass_taps.restype = C.c_int
        L.xo_lowpass_taps.argtypes = [C.c_double] * 4 + [vp, C.c_int]
        L.xo_rrc_taps.restype = C.c_int
        L.xo_rrc_taps.argtypes = [C.c_double] * 4 + [C.c_int, vp, C.c_int]
        L.xo_mmse_table.argtypes = [vp]
        L.xo_fir_create.restype = vp
        L.xo_fir_create.argtypes = [C.c_uint, vp, C.c_int]
        L.xo_fir_destroy.argtypes = [vp]
        L.xo_fir_work.argtypes = [vp, vp, vp, C.c_int]
        L.xo_agc_init.argtypes = [C.POINTER(Agc)] + [C.c_float] * 4
        L.xo_agc_work.argtypes = [C.POINTER(Agc), vp, vp, C.c_int]
        L.xo_costas_init.argtypes = [C.POINTER(Costas), C.c_float]
        L.xo_costas_work.argtypes = [C.POINTER(Costas), vp, vp, C.c_int]
        L.xo_sincosf.argtypes = [C.c_float, fp, fp]
        L.xo_mm_create.restype = vp
        L.xo_mm_create.argtypes = [C.c_float] * 5
        L.xo_mm_destroy.argtypes = [vp]
        L.xo_mm_work.restype = C.c_int
        L.xo_mm_work.argtypes = [vp, vp, C.c_int, vp]
        L.xo_mm_work_trace.restype = C.c_int
        L.xo_mm_work_trace.argtypes = [vp, vp, C.c_int, vp, vp, vp]
        L.xo_mm_get_state.argtypes = [vp, C.POINTER(MMState)]
        L.xo_mm_export.argtypes = [vp, C.POINTER(MMState), vp]
        L.xo_mm_import.argtypes = [vp, C.POINTER(MMState), vp]
        L.xo_config_lrit.argtypes = [C.POINTER(Config), C.c_float, C.c_uint32]
        L.xo_config_hrit.argtypes = [C.POINTER(Config), C.c_float, C.c_uint32]
        L.xo_demod_create.restype = vp
        L.xo_demod_create.argtypes = [C.POINTER(Config)]
        L.xo_demod_destroy.argtypes = [vp]
        L.xo_demod_process.restype = C.c_int
        L.xo_demod_process.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.c_int]
        L.xo_demod_stage.restype = vp
        L.xo_demod_stage.argtypes = [vp, C.c_int, ip]
        L.xo_demod_decimator_ntaps.restype = C.c_int
        L.xo_demod_decimator_ntaps.argtypes = [vp]
        L.xo_demod_decimator_taps.restype = vp
        L.xo_demod_decimator_taps.argtypes = [vp]
        L.xo_demod_rrc_taps.restype = vp
        L.xo_demod_rrc_taps.argtypes = [vp]
        L.xo_demod_sps.restype = C.c_float
        L.xo_demod_sps.argtypes = [vp]
        L.xo_demod_costas.restype = C.POINTER(Costas)
        L.xo_demod_costas.argtypes = [vp]
        L.xo_demod_mm.restype = vp
        L.xo_demod_mm.argtypes = [vp]
        L.xo_quantize_i8.argtypes = [vp, vp, C.c_size_t]
        L.xo_sync_correlate.argtypes = [vp, C.c_uint32, vp, C.c_int, vp, vp, vp]
        L.xo_sync_correlate.restype = None
        L.xo_sync_fix_frames.argtypes = [vp, C.c_size_t, vp, vp, vp, C.c_uint32, C.c_uint32, vp, vp]
        L.xo_sync_fix_frames.restype = None
        L.xo_convert_samples.argtypes = [vp, C.c_int, vp, C.c_size_t]
        L.xo_rtl_init.argtypes = [C.POINTER(Rtl), C.c_float]
        L.xo_rtl_work.argtypes = [C.POINTER(Rtl), vp, C.c_uint, vp]
        L.xo_knobs_default.argtypes = [C.POINTER(Knobs)]
        L.xo_set_knobs.argtypes = [C.POINTER(Knobs)]
        L.xo_get_knobs.argtypes = [C.POINTER(Knobs)]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class knobs:
    """Context manager: objects created inside see the given knob values (defaults elsewhere)."""

    def __init__(self, **over):
        self.k = Knobs()
        lib().xo_knobs_default(C.byref(self.k))
        for name, v in over.items():
            setattr(self.k, name, int(v))

    def __enter__(self):
        self.old = Knobs()
        lib().xo_get_knobs(C.byref(self.old))
        lib().xo_set_knobs(C.byref(self.k))
        return self.k

    def __exit__(self, *exc):
        lib().xo_set_knobs(C.byref(self.old))
        return False


def _c64(a):
    a = np.ascontiguousarray(a, dtype=np.complex64)
    return a


# ---- tap designers -------------------------------------------------------
def lowpass_taps(gain, fs, cutoff, tw):
    n = lib().xo_lowpass_ntaps(fs, tw)
    t = np.zeros(n, np.float32)
    lib().xo_lowpass_taps(gain, fs, cutoff, tw, _p(t), n)
    return t


def rrc_taps(gain, fs, symbol_rate, alpha, ntaps):
    n = ntaps | 1
    t = np.zeros(n, np.float32)
    lib().xo_rrc_taps(gain, fs, symbol_rate, alpha, ntaps, _p(t), n)
    return t


def mmse_table():
    t = np.zeros((129, 8), np.float32)
    lib().xo_mmse_table(_p(t))
    return t


# ---- blocks (mirror the SatHelper class API used by demodulator.cpp) -----
class FirFilter:
    def __init__(self, decimation, taps):
        self.taps = np.ascontiguousarray(taps, np.float32)
        self.D = int(decimation)
        self._h = lib().xo_fir_create(self.D, _p(self.taps), len(self.taps))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().xo_fir_destroy(self._h)
            self._h = None

    def Work(self, x, n_out):
        x = _c64(x)
        assert len(x) >= n_out * self.D
        out = np.zeros(n_out, np.complex64)
        lib().xo_fir_work(self._h, _p(x), _p(out), n_out)
        return out


class AGC:
    def __init__(self, rate, reference, gain, max_gain):
        self.s = Agc()
        lib().xo_agc_init(C.byref(self.s), rate, reference, gain, max_gain)

    def Work(self, x):
        x = _c64(x)
        out = np.zeros(len(x), np.complex64)
        lib().xo_agc_work(C.byref(self.s), _p(x), _p(out), len(x))
        return out


class CostasLoop:
    def __init__(self, loop_bw, order=2):
        assert order == 2
        self.s = Costas()
        lib().xo_costas_init(C.byref(self.s), loop_bw)

    def Work(self, x):
        x = _c64(x)
        out = np.zeros(len(x), np.complex64)
        lib().xo_costas_work(C.byref(self.s), _p(x), _p(out), len(x))
        return out


def sincosf(x):
    """xo_sincosf (glibc 2.35's __sincosf_fma restated) on an array of float32: (sin, cos)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    s = np.empty_like(x)
    c = np.empty_like(x)
    sv, cv = C.c_float(), C.c_float()
    f = lib().xo_sincosf
    for i, v in enumerate(x.tolist()):
        f(v, C.byref(sv), C.byref(cv))
        s[i] = sv.value
        c[i] = cv.value
    return s, c


class RtlIngest:
    """RtlFrontend's byte -> float conversion (RtlFrontend.cpp:26-28,57,102-116)."""

    def __init__(self, sample_rate):
        self.s = Rtl()
        lib().xo_rtl_init(C.byref(self.s), sample_rate)

    def Work(self, data):
        d = np.ascontiguousarray(data, np.uint8)
        out = np.zeros(len(d), np.float32)
        lib().xo_rtl_work(C.byref(self.s), _p(d), len(d), _p(out))
        return out.view(np.complex64)


class ClockRecovery:
    def __init__(self, omega, gain_omega, mu, gain_mu, omega_rel_limit):
        self._h = lib().xo_mm_create(omega, gain_omega, mu, gain_mu, omega_rel_limit)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().xo_mm_destroy(self._h)
            self._h = None

    def Work(self, x, trace=False):
        x = _c64(x)
        out = np.zeros(len(x) + 64, np.complex64)
        if trace:
            arm = np.zeros(len(x) + 64, np.int32)
            mu = np.zeros(len(x) + 64, np.float32)
            n = lib().xo_mm_work_trace(self._h, _p(x), len(x), _p(out), _p(arm), _p(mu))
            return out[:n].copy(), arm[:n].copy(), mu[:n].copy()
        n = lib().xo_mm_work(self._h, _p(x), len(x), _p(out))
        return out[:n].copy()

    def state(self):
        s = MMState()
        lib().xo_mm_get_state(self._h, C.byref(s))
        return s

    # what the object carries from one Work call to the next, as one array of float32 words (tests/dist_twin.py hands it from
    # rank to rank): the MMState, then the samples not consumed yet
    CARRY_WORDS = C.sizeof(MMState) // 4 + 2 * 2048

    def export_carry(self):
        s = MMState()
        tail = np.zeros(2048, np.complex64)
        lib().xo_mm_export(self._h, C.byref(s), _p(tail))
        out = np.zeros(self.CARRY_WORDS, np.float32)
        out[:C.sizeof(MMState) // 4] = np.frombuffer(bytes(s), np.float32)
        out[C.sizeof(MMState) // 4:] = tail.view(np.float32)
        return out

    def import_carry(self, words):
        words = np.ascontiguousarray(words, np.float32)
        assert len(words) == self.CARRY_WORDS
        s = MMState.from_buffer_copy(words[:C.sizeof(MMState) // 4].tobytes())
        tail = words[C.sizeof(MMState) // 4:].copy()
        lib().xo_mm_import(self._h, C.byref(s), _p(tail))


class _BorrowedClock(ClockRecovery):
    """The clock recovery INSIDE a Demod (not owned: no destroy)."""

    def __init__(self, handle, owner):
        self._h = handle
        self._owner = owner

    def __del__(self):
        self._h = None


def config(mode="lrit", sample_rate=1.25e6, decimation=1):
    c = Config()
    if mode == "lrit":
        lib().xo_config_lrit(C.byref(c), sample_rate, decimation)
    elif mode == "hrit":
        lib().xo_config_hrit(C.byref(c), sample_rate, decimation)
    else:
        raise ValueError(mode)
    return c


class Demod:
    """The chain of demodulator.cpp:100-168 (processSamples)."""
    STAGES = ("decimator", "agc", "rrc", "costas", "clock")

    def __init__(self, cfg):
        self.cfg = cfg
        self._h = lib().xo_demod_create(C.byref(cfg))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().xo_demod_destroy(self._h)
            self._h = None

    @property
    def sps(self):
        return lib().xo_demod_sps(self._h)

    @property
    def costas(self):
        """The chain's own Costas loop (ctypes view: .phase / .freq may be read and set between calls)."""
        return lib().xo_demod_costas(self._h).contents

    @property
    def clock(self):
        """The chain's own clock recovery (export_carry / import_carry between calls)."""
        return _BorrowedClock(lib().xo_demod_mm(self._h), self)

    def decimator_taps(self):
        n = lib().xo_demod_decimator_ntaps(self._h)
        ptr = lib().xo_demod_decimator_taps(self._h)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), (n,)).copy()

    def rrc_taps(self):
        ptr = lib().xo_demod_rrc_taps(self._h)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), (self.cfg.rrc_taps | 1,)).copy()

    def process(self, samples, sample_type=SAMPLE_FLOATIQ):
        if sample_type == SAMPLE_FLOATIQ:
            a = _c64(samples)
            n = len(a)
        elif sample_type == SAMPLE_S16IQ:
            a = np.ascontiguousarray(samples, np.int16)
            n = len(a) // 2
        elif sample_type == SAMPLE_U8IQ:
            a = np.ascontiguousarray(samples, np.uint8)
            n = len(a) // 2
        else:
            a = np.ascontiguousarray(samples, np.int8)
            n = len(a) // 2
        out = np.zeros(n + 64, np.float32)
        ns = lib().xo_demod_process(self._h, _p(a), n, sample_type, _p(out), len(out))
        if ns < 0:
            raise RuntimeError("oracle output capacity")
        return out[:ns].copy()

    def stage(self, name):
        idx = self.STAGES.index(name)
        n = C.c_int(0)
        ptr = lib().xo_demod_stage(self._h, idx, C.byref(n))
        if n.value == 0:
            return np.zeros(0, np.complex64)
        arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), (2 * n.value,)).copy()
        return arr.view(np.complex64)


def quantize_i8(x):
    x = np.ascontiguousarray(x, np.float32)
    out = np.zeros(len(x), np.int8)
    lib().xo_quantize_i8(_p(x), _p(out), len(x))
    return out


LRIT_UW0, LRIT_UW2 = 0xfca2b63db00d9794, 0x035d49c24ff2686b       # decoder/src/newdecoder.cpp:21-24
HRIT_UW0, HRIT_UW2 = 0xfc4ef4fd0cc2df89, 0x25010b02f33d2076


def sync_correlate(data, words=(LRIT_UW0, LRIT_UW2), frame=16384):
    """SatHelper::Correlator::correlate over consecutive windows of `frame` soft bytes: (word, position,
    correlation) per window, as decoder/src/newdecoder.cpp:218-245 reads them."""
    d = np.ascontiguousarray(data, np.int8)
    w = np.asarray(words, np.uint64)
    nf = len(d) // frame
    out = np.zeros((nf, 3), np.uint32)
    a, b, c = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
    for f in range(nf):
        seg = d[f * frame:(f + 1) * frame]
        lib().xo_sync_correlate(_p(seg), frame, _p(w), len(w), C.byref(a), C.byref(b), C.byref(c))
        out[f] = (a.value, b.value, c.value)
    return out


def sync_fix_frames(data, hits, frame=16384, min_correlation=46):
    """Frame alignment + phase fix as decoder/src/newdecoder.cpp:239-270 does them: (frames, valid)."""
    d = np.ascontiguousarray(data, np.int8)
    nf = len(d) // frame
    h = np.asarray(hits, np.uint32)[:nf]
    word, pos, corr = (np.ascontiguousarray(h[:, i]) for i in range(3))
    frames = np.zeros((nf, frame), np.int8)
    valid = np.zeros(nf, np.uint8)
    lib().xo_sync_fix_frames(_p(d), len(d), _p(word), _p(pos), _p(corr), frame, min_correlation, _p(frames), _p(valid))
    return frames, valid
