"""ctypes binding of libxritdemod_amd.so (include/xritdemod_amd.h)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "lib", "libxritdemod_amd.so")

SAMPLE_FLOATIQ, SAMPLE_S16IQ, SAMPLE_S8IQ, SAMPLE_U8IQ = 0, 1, 2, 3


class XritError(RuntimeError):
    def __init__(self, code, text):
        super().__init__(f"xritdemod_amd error {code}: {text}")
        self.code = code


class DemodConfig(C.Structure):
    _fields_ = [("sample_rate", C.c_float), ("decimation", C.c_uint32), ("symbol_rate", C.c_uint32),
                ("rrc_alpha", C.c_float), ("rrc_taps", C.c_int32),
                ("agc_rate", C.c_float), ("agc_reference", C.c_float), ("agc_gain", C.c_float),
                ("agc_max_gain", C.c_float), ("pll_alpha", C.c_float),
                ("clock_mu", C.c_float), ("clock_alpha", C.c_float), ("clock_gain_omega", C.c_float),
                ("clock_omega_limit", C.c_float),
                ("device", C.c_int32), ("costas_chain_len", C.c_int32), ("clock_chain_syms", C.c_int32),
                ("max_passes", C.c_int32), ("strict", C.c_int32), ("clock_min_passes", C.c_int32),
                ("slices", C.c_int32), ("clock_serial", C.c_int32), ("clock_exact", C.c_int32),
                ("clock_exact_window", C.c_int32), ("front_exact", C.c_int32), ("reserved", C.c_int32 * 2)]


class DemodStats(C.Structure):
    _fields_ = [("samples_in", C.c_uint64), ("circuit_samples", C.c_uint64), ("symbols_out", C.c_uint64),
                ("costas_passes", C.c_int32), ("clock_passes", C.c_int32),
                ("costas_unconverged", C.c_uint32), ("clock_unconverged", C.c_uint32),
                ("costas_max_residual", C.c_float), ("clock_max_residual", C.c_float),
                ("agc_serial_fallback", C.c_int32), ("clock_open_large", C.c_uint32), ("costas_serial_walk", C.c_int32),
                ("clock_relay_passes", C.c_int32), ("clock_relay_closed", C.c_int32), ("clock_relay_segments", C.c_int32)]


class SynthParams(C.Structure):
    _fields_ = [("fs_in", C.c_double), ("symbol_rate", C.c_double), ("alpha", C.c_double),
                ("amplitude", C.c_double), ("carrier_hz", C.c_double), ("phase0", C.c_double),
                ("timing_offset", C.c_double), ("clock_ppm", C.c_double), ("esn0_db", C.c_double),
                ("seed", C.c_uint64)]


# every symbol include/xritdemod_amd.h declares: (restype, argtypes)
_vp, _sz = C.c_void_p, C.c_size_t
_SIGNATURES = {
    "xrit_last_error": (C.c_char_p, []),
    "xrit_version": (C.c_char_p, []),
    "xrit_device_count": (C.c_int, []),
    "xrit_build_experiments": (C.c_int, []),
    "xrit_lowpass_taps": (C.c_int, [C.c_double] * 4 + [_vp, C.c_int]),
    "xrit_rrc_taps": (C.c_int, [C.c_double] * 4 + [C.c_int, _vp, C.c_int]),
    "xrit_mmse_table": (None, [_vp]),
    "xrit_demod_config_lrit": (None, [C.POINTER(DemodConfig), C.c_float, C.c_uint32]),
    "xrit_demod_config_hrit": (None, [C.POINTER(DemodConfig), C.c_float, C.c_uint32]),
    "xrit_demod_create": (C.c_int, [C.POINTER(DemodConfig), C.POINTER(_vp)]),
    "xrit_demod_destroy": (None, [_vp]),
    "xrit_demod_process": (C.c_int, [_vp, _vp, _sz, C.c_int, _vp, _sz, C.POINTER(_sz)]),
    "xrit_demod_process_device": (C.c_int, [_vp, _vp, _sz, C.c_int, _vp, _sz, C.POINTER(_sz), _vp]),
    "xrit_demod_prefetch_device": (C.c_int, [_vp, _vp, _sz, C.c_int, _vp]),
    "xrit_demod_reset": (C.c_int, [_vp, _vp]),
    "xrit_demod_stream": (_vp, [_vp]),
    "xrit_demod_sps": (C.c_float, [_vp]),
    "xrit_demod_decimator_ntaps": (C.c_int, [_vp]),
    "xrit_demod_keep_stages": (C.c_int, [_vp, C.c_int]),
    "xrit_demod_read_stage": (C.c_int, [_vp, C.c_int, _vp, _sz, C.POINTER(_sz)]),
    "xrit_demod_get_stats": (C.c_int, [_vp, C.POINTER(DemodStats)]),
    "xrit_demod_profile": (C.c_int, [_vp, C.c_int]),
    "xrit_demod_profile_read": (C.c_int, [_vp, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.POINTER(C.c_int), C.c_int]),
    "xrit_group_restart": (C.c_int, [_vp]),
    "xrit_demod_prepare_flipped": (C.c_int, [_vp, _vp]),
    "xrit_demod_flip_costas_phase": (C.c_int, [_vp, _vp]),
    "xrit_demod_front_exact_for": (C.c_int, [_vp, _sz]),
    "xrit_demod_clock_carry_bytes": (_sz, []),
    "xrit_demod_export_clock_carry": (C.c_int, [_vp, C.c_int, _vp, _vp]),
    "xrit_demod_redo_clock_from": (C.c_int, [_vp, _vp, _vp, _sz, C.POINTER(_sz), _vp]),
    "xrit_demod_last_clock_exact": (C.c_int, [_vp]),
    "xrit_demod_prefetch_depth": (C.c_int, [_vp, _sz]),
    "xrit_demod_redo_clock_flipped": (C.c_int, [_vp, _vp, _sz, C.POINTER(_sz), _vp]),
    "xrit_demod_profile_samples": (C.c_int, [_vp, C.c_char_p, C.POINTER(C.c_float), C.c_int]),
    "xrit_quantize_i8_device": (C.c_int, [_vp, _vp, _sz, C.c_int, _vp]),
    "xrit_quantize_i8": (C.c_int, [_vp, _vp, _vp, _sz]),
    "xrit_sync_correlate_device": (C.c_int, [_vp, _sz, _vp, C.c_int, C.c_uint32, _vp, C.c_int, _vp]),
    "xrit_sync_correlate": (C.c_int, [_vp, _sz, _vp, C.c_int, C.c_uint32, _vp, C.c_int]),
    "xrit_sync_fix_frames_device": (C.c_int, [_vp, _sz, _vp, C.c_uint32, C.c_uint32, _vp, _vp, C.c_int, _vp]),
    "xrit_sync_fix_frames": (C.c_int, [_vp, _sz, _vp, C.c_uint32, C.c_uint32, _vp, _vp, C.c_int]),
    "xrit_fir_create": (C.c_int, [C.c_uint, _vp, C.c_int, C.c_int, C.POINTER(_vp)]),
    "xrit_fir_work": (C.c_int, [_vp, _vp, _vp, _sz]),
    "xrit_fir_set_exact": (C.c_int, [_vp, C.c_int]),
    "xrit_loop_sincosf": (C.c_int, [_vp, _vp, _vp, _sz, C.c_int]),
    "xrit_agc_set_exact": (C.c_int, [_vp, C.c_int]),
    "xrit_agc_exact_stats": (C.c_int, [_vp, C.POINTER(C.c_uint32)]),
    "xrit_costas_set_exact": (C.c_int, [_vp, C.c_int, C.c_int]),
    "xrit_costas_exact_stats": (C.c_int, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                          C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "xrit_fir_destroy": (None, [_vp]),
    "xrit_agc_create": (C.c_int, [C.c_float] * 4 + [C.c_int, C.POINTER(_vp)]),
    "xrit_agc_work": (C.c_int, [_vp, _vp, _vp, _sz]),
    "xrit_agc_gain": (C.c_float, [_vp]),
    "xrit_agc_destroy": (None, [_vp]),
    "xrit_costas_create": (C.c_int, [C.c_float, C.c_int, C.c_int, C.POINTER(_vp)]),
    "xrit_costas_work": (C.c_int, [_vp, _vp, _vp, _sz]),
    "xrit_costas_state": (C.c_int, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "xrit_costas_destroy": (None, [_vp]),
    "xrit_clock_create": (C.c_int, [C.c_float] * 5 + [C.c_int, C.POINTER(_vp)]),
    "xrit_clock_work": (C.c_int, [_vp, _vp, _sz, _vp, _sz, C.POINTER(_sz)]),
    "xrit_clock_set_serial": (C.c_int, [_vp, C.c_int]),
    "xrit_clock_set_exact": (C.c_int, [_vp, C.c_int, C.c_int]),
    "xrit_clock_destroy": (None, [_vp]),
    "xrit_group_unique_id": (C.c_int, [_vp]),
    "xrit_group_create": (C.c_int, [C.POINTER(DemodConfig), C.c_int, C.c_int, _vp, C.POINTER(_vp)]),
    "xrit_group_create_all": (C.c_int, [C.POINTER(DemodConfig), C.POINTER(C.c_int), C.c_int, C.POINTER(_vp)]),
    "xrit_local_fabric_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "xrit_local_fabric_destroy": (None, [_vp]),
    "xrit_group_create_local": (C.c_int, [C.POINTER(DemodConfig), C.c_int, _vp, C.POINTER(_vp)]),
    "xrit_group_destroy": (None, [_vp]),
    "xrit_group_chain": (_vp, [_vp]),
    "xrit_group_rank": (C.c_int, [_vp]),
    "xrit_group_world": (C.c_int, [_vp]),
    "xrit_group_halo_samples": (_sz, [_vp]),
    "xrit_group_counters": (None, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "xrit_group_rccl_ranks": (C.c_int, [_vp]),
    "xrit_group_process_slice_device": (C.c_int, [_vp, _vp, _sz, C.c_int, _vp, _sz, C.POINTER(_sz), C.POINTER(C.c_uint64),
                                                  C.POINTER(C.c_int), _vp]),
    "xrit_group_process_slice_host": (C.c_int, [_vp, _vp, _sz, C.c_int, _vp, _sz, C.POINTER(_sz), C.POINTER(C.c_uint64),
                                                C.POINTER(C.c_int)]),
    "xrit_group_allreduce_max": (C.c_int, [_vp, C.POINTER(C.c_double), _vp]),
    "xrit_device_read_bandwidth": (C.c_int, [_vp, _sz, C.c_int, C.c_int, _vp, C.POINTER(C.c_double)]),
    "xrit_rtl_create": (C.c_int, [C.c_float, C.c_int, C.POINTER(_vp)]),
    "xrit_rtl_work": (C.c_int, [_vp, _vp, _sz, _vp]),
    "xrit_rtl_destroy": (None, [_vp]),
    "xrit_synth_defaults": (None, [C.POINTER(SynthParams)]),
    "xrit_synth_generate_device": (C.c_int, [C.POINTER(SynthParams), C.c_uint64, _sz, _vp, C.c_int, _vp]),
}

_lib = None


def lib_path():
    return _LIB


def build(force=False):
    """Compile libxritdemod_amd.so for gfx950 with hipcc (csrc/Makefile)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j8", "-s"]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd)
    return _LIB


def lib():
    """Load the HIP library; fails loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            raise XritError(-2, f"{_LIB} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                "(the HIP extension is required; there is no CPU path)")
        L = C.CDLL(_LIB)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise XritError(rc, lib().xrit_last_error().decode("utf-8", "replace"))


def device_count():
    return lib().xrit_device_count()


def build_experiments():
    """True when the library carries the measurement switches (make EXTRA=-DXRIT_EXPERIMENTS)."""
    return bool(lib().xrit_build_experiments())


def version():
    return lib().xrit_version().decode()


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _c64(a):
    return np.ascontiguousarray(a, dtype=np.complex64)


class Filters:
    """SatHelper::Filters (demodulator.cpp:443-444)."""

    @staticmethod
    def RRC(gain, sample_rate, symbol_rate, alpha, ntaps):
        t = np.zeros(ntaps | 1, np.float32)
        n = lib().xrit_rrc_taps(gain, sample_rate, symbol_rate, alpha, ntaps, _p(t), len(t))
        assert n == len(t)
        return t

    @staticmethod
    def lowPass(gain, sample_rate, cutoff, transition_width, window="HAMMING", beta=6.76):
        if window != "HAMMING":
            raise ValueError("the reference only uses FFTWindows::HAMMING (demodulator.cpp:444)")
        n = -lib().xrit_lowpass_taps(gain, sample_rate, cutoff, transition_width, None, 0)
        t = np.zeros(n, np.float32)
        assert lib().xrit_lowpass_taps(gain, sample_rate, cutoff, transition_width, _p(t), n) == n
        return t

    @staticmethod
    def mmse_table():
        t = np.zeros((129, 8), np.float32)
        lib().xrit_mmse_table(_p(t))
        return t


def loop_sincosf(x, device=0):
    """The exact Costas loop's sincosf (csrc/exact_sincos.h) on the device: (sin, cos) of an array of float32, |x| < 120."""
    x = np.ascontiguousarray(x, np.float32)
    s, c = np.empty_like(x), np.empty_like(x)
    _check(lib().xrit_loop_sincosf(_p(x), _p(s), _p(c), len(x), device))
    return s, c


class _Handle:
    _destroy = None

    def __init__(self):
        self._h = C.c_void_p()

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            getattr(lib(), self._destroy)(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FirFilter(_Handle):
    """SatHelper::FirFilter(decimation, taps); Work(in, out, nOut)."""
    _destroy = "xrit_fir_destroy"

    def __init__(self, decimation, taps, device=0, exact=False):
        super().__init__()
        taps = np.ascontiguousarray(taps, np.float32)
        self.D = int(decimation)
        _check(lib().xrit_fir_create(self.D, _p(taps), len(taps), device, C.byref(self._h)))
        if exact:       # summed in the CPU chain's order (cfg.front_exact = 2)
            _check(lib().xrit_fir_set_exact(self._h, 1))

    def Work(self, x, n_out):
        x = _c64(x)
        assert len(x) >= n_out * self.D
        out = np.zeros(n_out, np.complex64)
        _check(lib().xrit_fir_work(self._h, _p(x), _p(out), n_out))
        return out


class AGC(_Handle):
    """SatHelper::AGC(rate, reference, gain, maxGain); Work(in, out, n)."""
    _destroy = "xrit_agc_destroy"

    def __init__(self, rate, reference, gain, max_gain, device=0, exact=False):
        super().__init__()
        _check(lib().xrit_agc_create(rate, reference, gain, max_gain, device, C.byref(self._h)))
        if exact:       # the float32 recurrence walked literally (cfg.front_exact = 2)
            _check(lib().xrit_agc_set_exact(self._h, 1))

    def Work(self, x):
        x = _c64(x)
        out = np.zeros(len(x), np.complex64)
        _check(lib().xrit_agc_work(self._h, _p(x), _p(out), len(x)))
        return out

    @property
    def gain(self):
        return lib().xrit_agc_gain(self._h)

    def exact_stats(self):
        c = (C.c_uint32 * 8)()
        _check(lib().xrit_agc_exact_stats(self._h, c))
        return {"joints_open": c[0], "blocks": c[1], "picard_rounds": c[2], "at_round_limit": c[3], "lattice_segments": c[4],
                "lattice_fallbacks": c[5]}


class CostasLoop(_Handle):
    """SatHelper::CostasLoop(loopBw, order); Work(in, out, n)."""
    _destroy = "xrit_costas_destroy"

    def __init__(self, loop_bw, order=2, device=0, exact=False, history=0):
        super().__init__()
        _check(lib().xrit_costas_create(loop_bw, order, device, C.byref(self._h)))
        if exact:       # the output on the serial float32 trajectory (cfg.front_exact = 2)
            _check(lib().xrit_costas_set_exact(self._h, 1, int(history)))

    def exact_stats(self):
        b, r, o, f, sg, fb = C.c_uint64(), C.c_uint64(), C.c_uint32(), C.c_uint32(), C.c_uint64(), C.c_uint64()
        _check(lib().xrit_costas_exact_stats(self._h, C.byref(b), C.byref(r), C.byref(o), C.byref(f), C.byref(sg), C.byref(fb)))
        return {"blocks": b.value, "picard_rounds": r.value, "joints_open_after_batch": o.value, "host_rounds": f.value,
                "lattice_segments": sg.value, "lattice_fallbacks": fb.value}

    def Work(self, x):
        x = _c64(x)
        out = np.zeros(len(x), np.complex64)
        _check(lib().xrit_costas_work(self._h, _p(x), _p(out), len(x)))
        return out

    def state(self):
        ph, fr = C.c_float(), C.c_float()
        _check(lib().xrit_costas_state(self._h, C.byref(ph), C.byref(fr)))
        return ph.value, fr.value


class ClockRecovery(_Handle):
    """SatHelper::ClockRecovery(omega, gainOmega, mu, gainMu, omegaRelativeLimit); Work(in, out, n) -> symbols."""
    _destroy = "xrit_clock_destroy"

    def __init__(self, omega, gain_omega, mu, gain_mu, omega_rel_limit, device=0, serial=False, exact=0, window=0):
        super().__init__()
        _check(lib().xrit_clock_create(omega, gain_omega, mu, gain_mu, omega_rel_limit, device, C.byref(self._h)))
        if serial:
            _check(lib().xrit_clock_set_serial(self._h, 1))
        if exact:
            _check(lib().xrit_clock_set_exact(self._h, int(exact), int(window)))

    def Work(self, x):
        x = _c64(x)
        cap = len(x) + 64
        out = np.zeros(cap, np.complex64)
        n = C.c_size_t(0)
        _check(lib().xrit_clock_work(self._h, _p(x), len(x), _p(out), cap, C.byref(n)))
        return out[:n.value].copy()


class RtlIngest(_Handle):
    """RtlFrontend's byte -> float conversion with its DC tracker (RtlFrontend.cpp:26-28,57,102-116)."""
    _destroy = "xrit_rtl_destroy"

    def __init__(self, sample_rate, device=0):
        super().__init__()
        _check(lib().xrit_rtl_create(sample_rate, device, C.byref(self._h)))

    def Work(self, data):
        d = np.ascontiguousarray(data, np.uint8)
        n = len(d) // 2
        out = np.zeros(n, np.complex64)
        _check(lib().xrit_rtl_work(self._h, _p(d), n, _p(out)))
        return out


class Demodulator(_Handle):
    """The chain of processSamples() (demodulator.cpp:100-168) behind one handle."""
    _destroy = "xrit_demod_destroy"
    STAGES = ("decimator", "agc", "rrc", "costas", "clock")

    @staticmethod
    def config(mode="lrit", sample_rate=1.25e6, decimation=1, device=0, **over):
        c = DemodConfig()
        if mode == "lrit":
            lib().xrit_demod_config_lrit(C.byref(c), sample_rate, decimation)
        elif mode == "hrit":
            lib().xrit_demod_config_hrit(C.byref(c), sample_rate, decimation)
        else:
            raise ValueError(mode)
        c.device = device
        for k, v in over.items():
            setattr(c, k, v)
        return c

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        _check(lib().xrit_demod_create(C.byref(cfg), C.byref(self._h)))

    @property
    def sps(self):
        return lib().xrit_demod_sps(self._h)

    @property
    def decimator_ntaps(self):
        return lib().xrit_demod_decimator_ntaps(self._h)

    def reset(self, stream=None):
        """Back to the freshly created state (another stream begins); device buffers are kept."""
        _check(lib().xrit_demod_reset(self._h, C.c_void_p(stream) if stream else None))

    def prefetch_depth(self, n):
        """Inputs of n samples that may wait behind the call in progress (xrit_demod_prefetch_depth)."""
        return lib().xrit_demod_prefetch_depth(self._h, n)

    def redo_clock_flipped(self, d_soft_ptr, cap, stream=None):
        """The last call's clock recovery once more on the sign-flipped Costas output (xrit_demod_redo_clock_flipped)."""
        n_out = C.c_size_t(0)
        _check(lib().xrit_demod_redo_clock_flipped(self._h, C.c_void_p(d_soft_ptr), cap, C.byref(n_out),
                                                   C.c_void_p(stream) if stream else None))
        return n_out.value

    def prepare_flipped(self, stream=None):
        _check(lib().xrit_demod_prepare_flipped(self._h, C.c_void_p(stream) if stream else None))

    def flip_costas_phase(self, stream=None):
        """Move the carried Costas phase by pi (a freshly reset chain then pulls into its other lock)."""
        _check(lib().xrit_demod_flip_costas_phase(self._h, C.c_void_p(stream) if stream else None))

    def export_clock_carry(self, d_record_ptr, which=0, stream=None):
        """The clock recovery's carried state into a device record of clock_carry_bytes() bytes."""
        _check(lib().xrit_demod_export_clock_carry(self._h, int(which), C.c_void_p(d_record_ptr), C.c_void_p(stream) if stream else None))

    def redo_clock_from(self, d_record_ptr, d_soft_ptr, cap, stream=None):
        """The last call's clock recovery once more from another handle's carried state; returns the symbol count."""
        n_out = _sz(0)
        _check(lib().xrit_demod_redo_clock_from(self._h, C.c_void_p(d_record_ptr), C.c_void_p(d_soft_ptr), cap, C.byref(n_out),
                                                C.c_void_p(stream) if stream else None))
        return n_out.value

    def last_clock_exact(self):
        return lib().xrit_demod_last_clock_exact(self._h) == 1

    def front_exact_for(self, n):
        """True if a call of n input samples takes the bit-exact front end on this handle."""
        r = lib().xrit_demod_front_exact_for(self._h, int(n))
        _check(r if r < 0 else 0)
        return r == 1

    def keep_stages(self, enable=True):
        _check(lib().xrit_demod_keep_stages(self._h, int(enable)))

    def process(self, samples, sample_type=SAMPLE_FLOATIQ):
        """Host buffers in, host soft symbols out."""
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
        cap = n + 64
        out = np.zeros(cap, np.float32)
        n_out = C.c_size_t(0)
        _check(lib().xrit_demod_process(self._h, _p(a), n, sample_type, _p(out), cap, C.byref(n_out)))
        return out[:n_out.value].copy()

    def process_device(self, d_samples_ptr, n, d_soft_ptr, cap, sample_type=SAMPLE_FLOATIQ, stream=None):
        """Device pointers (ints) in and out; returns the symbol count."""
        n_out = C.c_size_t(0)
        _check(lib().xrit_demod_process_device(self._h, C.c_void_p(d_samples_ptr), n, sample_type,
                                               C.c_void_p(d_soft_ptr), cap, C.byref(n_out),
                                               C.c_void_p(stream) if stream else None))
        return n_out.value

    def prefetch_device(self, d_samples_ptr, n, sample_type=SAMPLE_FLOATIQ, stream=None):
        """Start the front end of the NEXT process_device call's input now (it overlaps the loops of the call in between)."""
        _check(lib().xrit_demod_prefetch_device(self._h, C.c_void_p(d_samples_ptr), n, sample_type,
                                                C.c_void_p(stream) if stream else None))

    def stage(self, name):
        idx = self.STAGES.index(name)
        n = C.c_size_t(0)
        _check(lib().xrit_demod_read_stage(self._h, idx, None, 0, C.byref(n)))
        out = np.zeros(n.value, np.complex64)
        if n.value:
            _check(lib().xrit_demod_read_stage(self._h, idx, _p(out), n.value, C.byref(n)))
        return out

    def stats(self):
        s = DemodStats()
        _check(lib().xrit_demod_get_stats(self._h, C.byref(s)))
        return s

    def profile(self, enable=True):
        """True/1: bracket every kernel with HIP events; 2: only the decimating FIR; False/0: off."""
        _check(lib().xrit_demod_profile(self._h, int(enable)))

    def profile_read(self):
        cap = 64
        names = (C.c_char_p * cap)()
        ms = (C.c_float * cap)()
        cnt = (C.c_int * cap)()
        n = lib().xrit_demod_profile_read(self._h, names, ms, cnt, cap)
        return [(names[i].decode(), float(ms[i]), int(cnt[i])) for i in range(n)]

    def profile_samples(self, name, cap=4096):
        """Every bracket of one kernel name since profile(enable), in launch order (ms)."""
        ms = (C.c_float * cap)()
        n = lib().xrit_demod_profile_samples(self._h, name.encode(), ms, cap)
        return [float(ms[i]) for i in range(max(n, 0))]

    def quantize_i8(self, soft):
        soft = np.ascontiguousarray(soft, np.float32)
        out = np.zeros(len(soft), np.int8)
        _check(lib().xrit_quantize_i8(self._h, _p(soft), _p(out), len(soft)))
        return out


def synth_generate_device(params, start, n, d_out_ptr, device=0, stream=None):
    _check(lib().xrit_synth_generate_device(C.byref(params), start, n, C.c_void_p(d_out_ptr), device,
                                            C.c_void_p(stream) if stream else None))


GROUP_ID_BYTES = 128


def group_unique_id():
    """ncclGetUniqueId as bytes: rank 0 makes it, the launcher hands it to every rank."""
    buf = (C.c_char * GROUP_ID_BYTES)()
    _check(lib().xrit_group_unique_id(buf))
    return bytes(buf)


class LocalFabric(_Handle):
    """In-process exchange between the ranks of a Group (threads of one process)."""
    _destroy = "xrit_local_fabric_destroy"

    def __init__(self, world):
        super().__init__()
        self.world = world
        _check(lib().xrit_local_fabric_create(world, C.byref(self._h)))


class Group(_Handle):
    """One capture across the GPUs of a node (xrit_group_*): Group(cfg, rank, world, unique_id) with RCCL, or
    Group(cfg, rank, fabric=LocalFabric(world)) for ranks that are threads of one process."""
    _destroy = "xrit_group_destroy"

    def __init__(self, cfg, rank, world=None, unique_id=None, fabric=None):
        super().__init__()
        self.cfg = cfg
        self._fabric = fabric
        if fabric is not None:
            _check(lib().xrit_group_create_local(C.byref(cfg), rank, fabric._h, C.byref(self._h)))
        else:
            idb = (C.c_char * GROUP_ID_BYTES).from_buffer_copy(unique_id)
            _check(lib().xrit_group_create(C.byref(cfg), rank, world, idb, C.byref(self._h)))

    @property
    def halo_samples(self):
        return lib().xrit_group_halo_samples(self._h)

    def counters(self):
        """(relocks, handovers, joined): slices started a second time from the other Costas lock, slices whose clock recovery
        ran again from the loop state of the rank in front, slices that had met that state inside their halo."""
        a, b, c = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        lib().xrit_group_counters(self._h, C.byref(a), C.byref(b), C.byref(c))
        return int(a.value), int(b.value), int(c.value)

    @property
    def relocks(self):
        return self.counters()[0]

    @property
    def rccl_ranks(self):
        """Ranks of the RCCL communicator behind the group, as ncclCommCount reports them (0: in-process fabric)."""
        return lib().xrit_group_rccl_ranks(self._h)

    @property
    def rank(self):
        return lib().xrit_group_rank(self._h)

    @property
    def world(self):
        return lib().xrit_group_world(self._h)

    def chain_process_device(self, d_samples_ptr, n, d_soft_ptr, cap, sample_type=SAMPLE_FLOATIQ, stream=None):
        """Independent segments: the rank's own chain, no exchange."""
        n_out = C.c_size_t(0)
        _check(lib().xrit_demod_process_device(lib().xrit_group_chain(self._h), C.c_void_p(d_samples_ptr), n, sample_type,
                                               C.c_void_p(d_soft_ptr), cap, C.byref(n_out),
                                               C.c_void_p(stream) if stream else None))
        return n_out.value

    def process_slice_device(self, d_samples_ptr, n, d_soft_ptr, cap, sample_type=SAMPLE_FLOATIQ, stream=None):
        """Collective: (symbol count, offset in the burst's symbol sequence, absolute polarity)."""
        n_out, off, pol = C.c_size_t(0), C.c_uint64(0), C.c_int(1)
        _check(lib().xrit_group_process_slice_device(self._h, C.c_void_p(d_samples_ptr), n, sample_type,
                                                     C.c_void_p(d_soft_ptr), cap, C.byref(n_out), C.byref(off),
                                                     C.byref(pol), C.c_void_p(stream) if stream else None))
        return n_out.value, off.value, pol.value

    def restart(self):
        """The next slice call is a capture's first (rank 0 starts cold, nothing is taken over from the last rank)."""
        _check(lib().xrit_group_restart(self._h))

    def allreduce_max(self, value, stream=None):
        v = C.c_double(value)
        _check(lib().xrit_group_allreduce_max(self._h, C.byref(v), C.c_void_p(stream) if stream else None))
        return v.value


def device_read_bandwidth(d_buf_ptr, nbytes, reps=10, device=0, stream=None):
    """GB/s of a hand-written read-only sweep over a device buffer (measurement helper)."""
    out = C.c_double(0)
    _check(lib().xrit_device_read_bandwidth(C.c_void_p(d_buf_ptr), nbytes, reps, device,
                                            C.c_void_p(stream) if stream else None, C.byref(out)))
    return out.value


def synth_params(**over):
    p = SynthParams()
    lib().xrit_synth_defaults(C.byref(p))
    for k, v in over.items():
        setattr(p, k, v)
    return p


def quantize_i8_device(d_soft_ptr, d_out_ptr, n, device=0, stream=None):
    _check(lib().xrit_quantize_i8_device(C.c_void_p(d_soft_ptr), C.c_void_p(d_out_ptr), n, device,
                                         C.c_void_p(stream) if stream else None))


# ---- decoder front end: frame synchronisation (decoder/src/newdecoder.cpp:21-24,145-151,218-245) ----------------
LRIT_UW0, LRIT_UW2 = 0xfca2b63db00d9794, 0x035d49c24ff2686b
HRIT_UW0, HRIT_UW2 = 0xfc4ef4fd0cc2df89, 0x25010b02f33d2076
CODED_FRAME_SIZE = 16384
MIN_CORRELATION_BITS = 46


def sync_correlate(symbols, words=(LRIT_UW0, LRIT_UW2), frame=CODED_FRAME_SIZE, device=0):
    """SatHelper::Correlator over consecutive windows of `frame` int8 soft symbols: array of (word, position,
    correlation) rows, one per window."""
    d = np.ascontiguousarray(symbols, np.int8)
    w = np.asarray(words, np.uint64)
    nf = len(d) // frame
    hits = np.zeros((nf, 4), np.uint32)
    _check(lib().xrit_sync_correlate(_p(d), len(d), _p(w), len(w), frame, _p(hits), device))
    return hits[:, :3].copy()


def sync_fix_frames(symbols, hits, frame=CODED_FRAME_SIZE, min_correlation=MIN_CORRELATION_BITS, device=0):
    """Frame alignment + phase fix between correlator and Viterbi (newdecoder.cpp:239-270): (frames, valid) --
    frames[f] starts at hits[f]'s position, inverted when the 180-degree word won; valid[f] = 0 (zeros) below the
    acceptance or past the end of the buffer.  hits: rows of (word, position, correlation)."""
    d = np.ascontiguousarray(symbols, np.int8)
    nf = len(d) // frame
    h = np.zeros((nf, 4), np.uint32)
    h[:, :3] = np.asarray(hits, np.uint32)[:nf, :3]
    frames = np.zeros((nf, frame), np.int8)
    valid = np.zeros(nf, np.uint8)
    _check(lib().xrit_sync_fix_frames(_p(d), len(d), _p(h), frame, min_correlation, _p(frames), _p(valid), device))
    return frames, valid


def sync_fix_frames_device(d_symbols_ptr, n, d_hits_ptr, d_frames_ptr, d_valid_ptr, frame=CODED_FRAME_SIZE,
                           min_correlation=MIN_CORRELATION_BITS, device=0, stream=None):
    _check(lib().xrit_sync_fix_frames_device(C.c_void_p(d_symbols_ptr), n, C.c_void_p(d_hits_ptr), frame, min_correlation,
                                             C.c_void_p(d_frames_ptr), C.c_void_p(d_valid_ptr), device,
                                             C.c_void_p(stream) if stream else None))


def sync_correlate_device(d_symbols_ptr, n, d_hits_ptr, words=(LRIT_UW0, LRIT_UW2), frame=CODED_FRAME_SIZE, device=0,
                          stream=None):
    w = np.asarray(words, np.uint64)
    _check(lib().xrit_sync_correlate_device(C.c_void_p(d_symbols_ptr), n, _p(w), len(w), frame, C.c_void_p(d_hits_ptr),
                                            device, C.c_void_p(stream) if stream else None))
