"""xritdemod_amd -- MI355X-native xRIT BPSK demodulation chain.

Python mirror of the C ABI in include/xritdemod_amd.h (ctypes).  The classes keep
the names and argument order of the SatHelper classes the reference builds at
/root/reference/demodulator/src/demodulator.cpp:443-450 and calls at
:138,:143,:148,:152,:156 (FirFilter, AGC, CostasLoop, ClockRecovery, Filters).

The compute path is the HIP library only: importing works without a GPU (so the
CPU test tier can check the ABI), but creating any stage or chain object raises
XritError when the library or a HIP device is missing.  There is no CPU fallback.
"""
from ._capi import (  # noqa: F401
    XritError, lib, lib_path, build, device_count, build_experiments, version,
    SAMPLE_FLOATIQ, SAMPLE_S16IQ, SAMPLE_S8IQ, SAMPLE_U8IQ,
    Filters, FirFilter, AGC, CostasLoop, ClockRecovery, RtlIngest, Demodulator, Group, LocalFabric, group_unique_id, DemodConfig, DemodStats,
    loop_sincosf, SynthParams as DeviceSynthParams, synth_generate_device, quantize_i8_device, sync_correlate, sync_correlate_device, sync_fix_frames, sync_fix_frames_device,
)
