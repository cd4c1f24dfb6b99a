// agc_wave.h -- ordered composition of AGC gain maps across a 64-lane wave with DPP row shifts
// (VALU only: a ds_bpermute shuffle tree costs ~25 instructions and an LDS round trip per step).
// Lane order = sample order, lower lanes first.
#pragma once

#include <hip/hip_runtime.h>

#include "loop_core.h"

namespace xrit {

template <int CTRL> __device__ __forceinline__ float agc_dpp(float fill, float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), CTRL, 0xf, 0xf, false));
}

// lanes without a source lane in their row of 16 receive the identity map
template <int SH> __device__ __forceinline__ AgcMap agc_row_shr(const AgcMap &v)
{
    AgcMap r;
    r.a = agc_dpp<0x110 + SH>(1.0f, v.a);
    r.b = agc_dpp<0x110 + SH>(0.0f, v.b);
    r.c = agc_dpp<0x110 + SH>(INFINITY, v.c);
    return r;
}

// inclusive scan inside every row of 16 lanes
__device__ __forceinline__ AgcMap agc_row_scan(AgcMap v)
{
    v = agc_compose(agc_row_shr<1>(v), v);
    v = agc_compose(agc_row_shr<2>(v), v);
    v = agc_compose(agc_row_shr<4>(v), v);
    v = agc_compose(agc_row_shr<8>(v), v);
    return v;
}

__device__ __forceinline__ AgcMap agc_read_lane(const AgcMap &v, int lane)
{
    AgcMap r;
    r.a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.a), lane));
    r.b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.b), lane));
    r.c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.c), lane));
    return r;
}

// composition of all 64 lanes, the same value in every lane
__device__ __forceinline__ AgcMap agc_wave_total(const AgcMap &v)
{
    const AgcMap s = agc_row_scan(v);
    AgcMap t = agc_read_lane(s, 15);
    t = agc_compose(t, agc_read_lane(s, 31));
    t = agc_compose(t, agc_read_lane(s, 47));
    t = agc_compose(t, agc_read_lane(s, 63));
    return t;
}

// composition of the lanes before this one (identity in lane 0)
__device__ __forceinline__ AgcMap agc_wave_exclusive(const AgcMap &v)
{
    const AgcMap s = agc_row_scan(v);
    const AgcMap t0 = agc_read_lane(s, 15), t1 = agc_read_lane(s, 31), t2 = agc_read_lane(s, 47);
    const int row = (threadIdx.x & 63) >> 4;
    AgcMap pre = agc_identity();
    if (row >= 1) pre = t0;
    if (row >= 2) pre = agc_compose(pre, t1);
    if (row >= 3) pre = agc_compose(pre, t2);
    return agc_compose(pre, agc_row_shr<1>(s));
}

}  // namespace xrit
