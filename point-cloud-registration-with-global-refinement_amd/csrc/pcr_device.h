// pcr_device.h -- device-side helpers shared by the kernels: Morton keys, point/box distances in a fixed operation
// order, 8-lane (octet) and 16-lane (row) DPP reductions, deterministic wave reductions.  gfx950 only (64-lane
// wavefronts are hard-coded).  The spatial index itself is in pcr_octree.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "pcr_internal.h"

#define PCR_WAVE 64

// ------------------------------------------------------------------------------------------ Morton
__host__ __device__ static inline uint64_t pcr_spread21(uint64_t x) {
    x &= 0x1fffffull;
    x = (x | (x << 32)) & 0x1f00000000ffffull;
    x = (x | (x << 16)) & 0x1f0000ff0000ffull;
    x = (x | (x << 8)) & 0x100f00f00f00f00full;
    x = (x | (x << 4)) & 0x10c30c30c30c30c3ull;
    x = (x | (x << 2)) & 0x1249249249249249ull;
    return x;
}
__host__ __device__ static inline uint64_t pcr_morton3(uint32_t x, uint32_t y, uint32_t z) {
    return pcr_spread21(x) | (pcr_spread21(y) << 1) | (pcr_spread21(z) << 2);
}

// squared distance in a fixed operation order (the same helper is used for boxes and points so that the
// box bound is monotone w.r.t. the point distances it guards)
__device__ static inline float pcr_d2(float dx, float dy, float dz) { return __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, dx * dx)); }

__device__ static inline float pcr_box_d2(const float4 lo, const float4 hi, float qx, float qy, float qz) {
    float dx = fmaxf(fmaxf(lo.x - qx, qx - hi.x), 0.0f);
    float dy = fmaxf(fmaxf(lo.y - qy, qy - hi.y), 0.0f);
    float dz = fmaxf(fmaxf(lo.z - qz, qz - hi.z), 0.0f);
    return pcr_d2(dx, dy, dz);
}

// ------------------------------------------------------------------------------ 8-lane (octet) DPP ops
// Data-parallel-primitive moves stay inside the VALU (no LDS crossbar, no lgkmcnt wait), unlike __shfl_xor which
// compiles to ds_bpermute.  quad_perm covers xor 1 / xor 2, row_half_mirror (lane i <-> 7-i of each 8) joins the
// two quads of an octet; after the first two steps a quad is uniform, so the mirror acts as xor 4.
template <int CTRL>
__device__ static inline int pcr_dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
template <int CTRL>
__device__ static inline float pcr_dpp_f(float v) { return __int_as_float(pcr_dpp_i<CTRL>(__float_as_int(v))); }
#define PCR_DPP_XOR1 0xB1      // quad_perm [1,0,3,2]
#define PCR_DPP_XOR2 0x4E      // quad_perm [2,3,0,1]
#define PCR_DPP_HMIRROR 0x141  // row_half_mirror
#define PCR_DPP_MIRROR 0x140   // row_mirror (lane i <-> 15-i of each 16-lane row)
// max / min over the octet for values that are >= 0 or exactly -1 (the k-best sentinel): their float order equals the
// signed-integer order of their bit patterns, and integer max needs no NaN canonicalisation
__device__ static inline float pcr_octet_max(float v) {
    int i = __float_as_int(v);
    i = max(i, pcr_dpp_i<PCR_DPP_XOR1>(i));
    i = max(i, pcr_dpp_i<PCR_DPP_XOR2>(i));
    i = max(i, pcr_dpp_i<PCR_DPP_HMIRROR>(i));
    return __int_as_float(i);
}
__device__ static inline float pcr_octet_min(float v) {
    int i = __float_as_int(v);
    i = min(i, pcr_dpp_i<PCR_DPP_XOR1>(i));
    i = min(i, pcr_dpp_i<PCR_DPP_XOR2>(i));
    i = min(i, pcr_dpp_i<PCR_DPP_HMIRROR>(i));
    return __int_as_float(i);
}
__device__ static inline int pcr_octet_sum_i(int v) {
    v += pcr_dpp_i<PCR_DPP_XOR1>(v); v += pcr_dpp_i<PCR_DPP_XOR2>(v); v += pcr_dpp_i<PCR_DPP_HMIRROR>(v);
    return v;
}
__device__ static inline double pcr_octet_sum(double v) {       // fixed tree => deterministic; all 8 lanes get the sum
    union { double d; int i[2]; } a, b;
    a.d = v; b.i[0] = pcr_dpp_i<PCR_DPP_XOR1>(a.i[0]); b.i[1] = pcr_dpp_i<PCR_DPP_XOR1>(a.i[1]); v += b.d;
    a.d = v; b.i[0] = pcr_dpp_i<PCR_DPP_XOR2>(a.i[0]); b.i[1] = pcr_dpp_i<PCR_DPP_XOR2>(a.i[1]); v += b.d;
    a.d = v; b.i[0] = pcr_dpp_i<PCR_DPP_HMIRROR>(a.i[0]); b.i[1] = pcr_dpp_i<PCR_DPP_HMIRROR>(a.i[1]); v += b.d;
    return v;
}

__device__ static inline double pcr_row16_sum(double v) {       // sum over each 16-lane DPP row, result in all 16 lanes
    union { double d; int i[2]; } a, b;
    a.d = v; b.i[0] = pcr_dpp_i<PCR_DPP_XOR1>(a.i[0]); b.i[1] = pcr_dpp_i<PCR_DPP_XOR1>(a.i[1]); v += b.d;
    a.d = v; b.i[0] = pcr_dpp_i<PCR_DPP_XOR2>(a.i[0]); b.i[1] = pcr_dpp_i<PCR_DPP_XOR2>(a.i[1]); v += b.d;
    a.d = v; b.i[0] = pcr_dpp_i<PCR_DPP_HMIRROR>(a.i[0]); b.i[1] = pcr_dpp_i<PCR_DPP_HMIRROR>(a.i[1]); v += b.d;
    a.d = v; b.i[0] = pcr_dpp_i<PCR_DPP_MIRROR>(a.i[0]); b.i[1] = pcr_dpp_i<PCR_DPP_MIRROR>(a.i[1]); v += b.d;
    return v;
}

// Sums over each 16-lane row of N <= 32 doubles per lane by HALVING: at every step a lane keeps half of its values and hands the other half to its
// partner (lane ^ 1, ^ 2, ^ 4, ^ 8), so the adds shrink 16 + 8 + 4 + 2 instead of 32 four times (218 against 384 instructions for 30 sums).
// Lane r of the row ends with the sums of values r (lo) and 16 + r (hi).  The pairing tree is that of pcr_row16_sum -- (l, l ^ 1), then ^ 2,
// ^ 4, ^ 8, operands possibly swapped -- so every sum has the SAME bits.  (xor 4 / xor 8 between lanes that hold different values need real
// exchanges: row_shl:4 / row_shr:4 on alternate banks, row_ror:8; the mirror steps of pcr_row16_sum rely on uniform quads / octets.)
__device__ static inline double pcr_dpp_xchg_d(double v, int step) {      // value of lane ^ (1 << step) of the 16-lane row
    union { double d; int i[2]; } a, b; a.d = v;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        if (step == 0) b.i[h] = __builtin_amdgcn_update_dpp(0, a.i[h], 0xB1, 0xf, 0xf, true);
        else if (step == 1) b.i[h] = __builtin_amdgcn_update_dpp(0, a.i[h], 0x4E, 0xf, 0xf, true);
        else if (step == 2) { int t = __builtin_amdgcn_update_dpp(0, a.i[h], 0x104 /* row_shl:4 */, 0xf, 0x5, false); b.i[h] = __builtin_amdgcn_update_dpp(t, a.i[h], 0x114 /* row_shr:4 */, 0xf, 0xa, false); }
        else b.i[h] = __builtin_amdgcn_update_dpp(0, a.i[h], 0x128 /* row_ror:8 */, 0xf, 0xf, true);
    }
    return b.d;
}
template <int N>
__device__ static inline void pcr_row16_sum_halving(const double *acc, int lane, double *lo, double *hi) {
    static_assert(N <= 32, "at most 32 values per lane");
    double v[32];
#pragma unroll
    for (int k = 0; k < 32; k++) v[k] = k < N ? acc[k] : 0.0;
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
    double w[16], x[8], y[4];
#pragma unroll
    for (int j = 0; j < 16; j++) { const double keep = b0 ? v[2 * j + 1] : v[2 * j], send = b0 ? v[2 * j] : v[2 * j + 1]; w[j] = keep + pcr_dpp_xchg_d(send, 0); }
#pragma unroll
    for (int j = 0; j < 8; j++) { const double keep = b1 ? w[2 * j + 1] : w[2 * j], send = b1 ? w[2 * j] : w[2 * j + 1]; x[j] = keep + pcr_dpp_xchg_d(send, 1); }
#pragma unroll
    for (int j = 0; j < 4; j++) { const double keep = b2 ? x[2 * j + 1] : x[2 * j], send = b2 ? x[2 * j] : x[2 * j + 1]; y[j] = keep + pcr_dpp_xchg_d(send, 2); }
    { const double keep = b3 ? y[1] : y[0], send = b3 ? y[0] : y[1]; *lo = keep + pcr_dpp_xchg_d(send, 3); }
    { const double keep = b3 ? y[3] : y[2], send = b3 ? y[2] : y[3]; *hi = keep + pcr_dpp_xchg_d(send, 3); }
}

// ------------------------------------------------------------------------ whole-wavefront reductions inside the VALU
// gfx950 adds v_permlane16_swap / v_permlane32_swap: with both operands the same value, one instruction hands every lane the value of
// lane ^ 16 (resp. ^ 32) -- an exchange between the 16-lane rows without the LDS crossbar (a __shfl_xor is ds_bpermute + s_waitcnt, ~60-100
// cycles per step: the six steps of a 64-lane butterfly cost more than the 16 MFMAs they sat between in the feature screen).
__device__ static inline unsigned pcr_swap16(unsigned v, unsigned *other) { auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false); *other = r[1]; return r[0]; }
__device__ static inline unsigned pcr_swap32(unsigned v, unsigned *other) { auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false); *other = r[1]; return r[0]; }
// over the four lanes {l, l ^ 16, l ^ 32, l ^ 48} (same position in the four rows); all four get the result
__device__ static inline float pcr_xrow_min(float v) {
    unsigned o; unsigned a = pcr_swap16(__float_as_uint(v), &o);
    v = fminf(__uint_as_float(a), __uint_as_float(o));
    a = pcr_swap32(__float_as_uint(v), &o);
    return fminf(__uint_as_float(a), __uint_as_float(o));
}
__device__ static inline float pcr_xrow_max(float v) {
    unsigned o; unsigned a = pcr_swap16(__float_as_uint(v), &o);
    v = fmaxf(__uint_as_float(a), __uint_as_float(o));
    a = pcr_swap32(__float_as_uint(v), &o);
    return fmaxf(__uint_as_float(a), __uint_as_float(o));
}
__device__ static inline unsigned pcr_xrow_or(unsigned v) {
    unsigned o; unsigned a = pcr_swap16(v, &o);
    v = a | o;
    a = pcr_swap32(v, &o);
    return a | o;
}
// over the eight lanes {l ^ 8 a ^ 16 b ^ 32 c} (the same position in the eight octets of the wavefront): row_ror:8 inside each row, then the rows
#define PCR_DPP_ROR8 0x128     // row_ror:8 (lane i <-> i ^ 8 of each 16-lane row)
__device__ static inline float pcr_xoct_min(float v) { v = fminf(v, pcr_dpp_f<PCR_DPP_ROR8>(v)); return pcr_xrow_min(v); }
__device__ static inline float pcr_xoct_max(float v) { v = fmaxf(v, pcr_dpp_f<PCR_DPP_ROR8>(v)); return pcr_xrow_max(v); }
// over all 64 lanes, result in every lane: DPP inside each row (xor 1, xor 2, half mirror, mirror), then across the rows
__device__ static inline float pcr_wave_max_all(float v) {
    v = fmaxf(v, pcr_dpp_f<PCR_DPP_XOR1>(v)); v = fmaxf(v, pcr_dpp_f<PCR_DPP_XOR2>(v));
    v = fmaxf(v, pcr_dpp_f<PCR_DPP_HMIRROR>(v)); v = fmaxf(v, pcr_dpp_f<PCR_DPP_MIRROR>(v));
    return pcr_xrow_max(v);
}
__device__ static inline unsigned pcr_wave_or_all(unsigned v) {
    v |= (unsigned)pcr_dpp_i<PCR_DPP_XOR1>((int)v); v |= (unsigned)pcr_dpp_i<PCR_DPP_XOR2>((int)v);
    v |= (unsigned)pcr_dpp_i<PCR_DPP_HMIRROR>((int)v); v |= (unsigned)pcr_dpp_i<PCR_DPP_MIRROR>((int)v);
    return pcr_xrow_or(v);
}

// ------------------------------------------------------------------------------------- reductions
__device__ static inline double pcr_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, PCR_WAVE);
    return v;   // valid in lane 0; fixed tree => deterministic
}
__device__ static inline float pcr_wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_down(v, o, PCR_WAVE));
    return v;
}
__device__ static inline float pcr_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_down(v, o, PCR_WAVE));
    return v;
}
