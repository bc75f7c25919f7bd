// wave_dpp.h — wave-wide sums on the DPP / v_readlane data path (product code, gfx950).
// hipcc lowers __shfl_xor to ds_bpermute_b32: every step of a butterfly is an LDS-crossbar round trip (two for a 64-bit value), ~100+ cycles each and
// strictly dependent.  The forms below stay in the VALU: an inclusive scan inside each 16-lane row (row_shr 1, 2, 4, 8), then row_bcast15 and row_bcast31
// carry the row totals upwards; lane 63 holds the wave's total and v_readlane returns it lane-uniform.  ALL 64 lanes must be active.  The summation tree
// is fixed (deterministic), but it is not the butterfly's: floating-point results differ from __shfl_xor sums in the last bits, integer results do not.
#pragma once
#include <hip/hip_runtime.h>

// DPP control words: row_shr:n = 0x110 + n, row_bcast:15 = 0x142, row_bcast:31 = 0x143.  Source lanes out of range and rows outside ROW_MASK deliver 0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, true); }
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v) { return __hiloint2double(dpp_i32<CTRL, ROW_MASK>(__double2hiint(v)), dpp_i32<CTRL, ROW_MASK>(__double2loint(v))); }
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ long long dpp_i64(long long v) {
    const unsigned lo = (unsigned)dpp_i32<CTRL, ROW_MASK>((int)(unsigned)(unsigned long long)v), hi = (unsigned)dpp_i32<CTRL, ROW_MASK>((int)((unsigned long long)v >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double lane_bcast(double v, int src_lane) {      // src_lane wave-uniform
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row_scan_f64(double v) {          // lane 15 of every 16-lane row ends up with the row's sum
    v += dpp_f64<0x111, 0xf>(v); v += dpp_f64<0x112, 0xf>(v); v += dpp_f64<0x114, 0xf>(v); v += dpp_f64<0x118, 0xf>(v);
    return v;
}
__device__ __forceinline__ double wave_sum_f64(double v) {          // the wave's total in every lane
    v = row_scan_f64(v);
    v += dpp_f64<0x142, 0xa>(v);
    v += dpp_f64<0x143, 0xc>(v);
    return lane_bcast(v, 63);
}
// sum of 64 ints whose 8-term partial sums fit 32 bits (the LK sums: |v| * 8 < 2^31): 32-bit for the first three steps, 64-bit from there on.  Exact.
__device__ __forceinline__ long long wave_sum_i32_wide(int v) {
    v += dpp_i32<0x111, 0xf>(v); v += dpp_i32<0x112, 0xf>(v); v += dpp_i32<0x114, 0xf>(v);      // lane i: lanes i-7 .. i of its row
    long long s = v;
    s += dpp_i64<0x118, 0xf>(s);
    s += dpp_i64<0x142, 0xa>(s);
    s += dpp_i64<0x143, 0xc>(s);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(unsigned long long)s, 63), hi = (unsigned)__builtin_amdgcn_readlane((int)((unsigned long long)s >> 32), 63);
    return (long long)(((unsigned long long)hi << 32) | lo);
}
