// Device-side helpers shared by the half-precision kernels (wn_half.hip, wn_fused.hip): storage types, LDS-DMA,
// value <-> storage conversion, gate activations.  Not part of the C ABI.
#pragma once
#include "wn_half.h"

namespace wn {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef __bf16 b4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// LDS-DMA from inline asm (M0 = LDS destination base, saved / set / restored inside the statement), counted by hand with the
// loop's `s_waitcnt vmcnt(N)`: hipcc then knows of no pending LDS write and cannot decide to drain the ring in front of an LDS
// read (it did exactly that in hwgrad_kernel with the builtin; wn_half_wgrad.hip).
// Address form: wave-uniform 64-bit base in an SGPR pair + a 32-bit per-lane byte offset (one VGPR that never changes), so a
// piece costs no vector address arithmetic and no address registers.
__device__ __forceinline__ void glds16(const char* ubase, unsigned lane_off, const char* lds_dst) {
    const unsigned dst = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)lds_dst;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane_off), "s"(ubase), "s"(dst) : "memory");
}
#define WN_GLDS(ub, lo, lp) glds16((ub), (lo), (lp))

// ---------------------------------------------------------------------------------------------------------------
// value <-> storage helpers
// ---------------------------------------------------------------------------------------------------------------
template <bool BF> struct HT;
template <> struct HT<false> { typedef _Float16 t; typedef h4 v4; typedef h8 v8; };
template <> struct HT<true> { typedef __bf16 t; typedef b4 v4; typedef b8 v8; };

// The lo plane must be the remainder against EXACTLY the bits stored in the hi plane.  hipcc is free to convert the same
// fp32 value twice (a packed v_cvt_pk_f16_f32 for the store, a scalar v_cvt_f16_f32 for the subtraction), and on gfx950 the
// two disagree on exact ties -- measured: 4 of 153 600 elements came out with lo = -half-ulp instead of +half-ulp, an
// error of one fp16 ulp.  Passing the packed hi through an empty asm makes it opaque: the remainder is then computed from
// the register that is stored.
template <typename V>
__device__ __forceinline__ V pin(V v) {
    asm volatile("" : "+v"(v));
    return v;
}

// four consecutive channels of one time step -> 8 bytes in plane 0 (and the fp16 remainder in plane 1)
template <int P, bool BF, bool CHK = true>   // CHK = false: the values cannot exceed fp16's range (tanh, sigmoid and their product)
__device__ __forceinline__ void store4(char* p, long long pstride, const float (&v)[4], unsigned& ovf) {
    typedef typename HT<BF>::t T;
    typedef typename HT<BF>::v4 V4;
    V4 hi;
#pragma unroll
    for (int q = 0; q < 4; ++q) hi[q] = (T)v[q];
    if constexpr (P == 2) hi = pin(hi);
    *reinterpret_cast<V4*>(p) = hi;
    // (written as !(|v| <= max) so that a NaN trips it too: weights beyond fp16's range pack to inf and inf * 0 = NaN downstream)
    if constexpr (!BF && CHK) {
#pragma unroll
        for (int q = 0; q < 4; ++q) ovf |= (!(__builtin_fabsf(v[q]) <= 65504.0f)) ? 1u : 0u;
    }
    if constexpr (P == 2) {
        V4 lo;
#pragma unroll
        for (int q = 0; q < 4; ++q) lo[q] = (T)(v[q] - (float)hi[q]);
        *reinterpret_cast<V4*>(p + pstride) = lo;
    }
}

// The same for TWO row groups (i, i + 1) of one time step at once: this lane holds channels 4h .. 4h + 3 of both (h = lane >> 5).
// After v_permlane32_swap of the packed halves, lanes 0-31 own the whole 16-byte unit of group i and lanes 32-63 that of group
// i + 1: ONE 16-byte store per plane instead of two 8-byte ones (the epilogues are store-issue bursts: cdna guide T21).
// `p` = this lane's unit: group (i + h) of its column, WITHOUT the 8 h byte offset store4 takes.
template <int P, bool BF, bool CHK = true>
__device__ __forceinline__ void store8(char* p, long long pstride, const float (&va)[4], const float (&vb)[4], unsigned& ovf) {
    typedef typename HT<BF>::t T;
    typedef typename HT<BF>::v4 V4;
    V4 ha, hb;
#pragma unroll
    for (int q = 0; q < 4; ++q) { ha[q] = (T)va[q]; hb[q] = (T)vb[q]; }
    if constexpr (P == 2) { ha = pin(ha); hb = pin(hb); }
    if constexpr (!BF && CHK) {
#pragma unroll
        for (int q = 0; q < 4; ++q) ovf |= (!(__builtin_fabsf(va[q]) <= 65504.0f) || !(__builtin_fabsf(vb[q]) <= 65504.0f)) ? 1u : 0u;
    }
    {
        const u32x2 ua = __builtin_bit_cast(u32x2, ha), ub = __builtin_bit_cast(u32x2, hb);
        const auto sx = __builtin_amdgcn_permlane32_swap(ua[0], ub[0], false, false);
        const auto sy = __builtin_amdgcn_permlane32_swap(ua[1], ub[1], false, false);
        *reinterpret_cast<u32x4*>(p) = u32x4{sx[0], sy[0], sx[1], sy[1]};
    }
    if constexpr (P == 2) {
        V4 la, lb;
#pragma unroll
        for (int q = 0; q < 4; ++q) { la[q] = (T)(va[q] - (float)ha[q]); lb[q] = (T)(vb[q] - (float)hb[q]); }
        const u32x2 ua = __builtin_bit_cast(u32x2, la), ub = __builtin_bit_cast(u32x2, lb);
        const auto sx = __builtin_amdgcn_permlane32_swap(ua[0], ub[0], false, false);
        const auto sy = __builtin_amdgcn_permlane32_swap(ua[1], ub[1], false, false);
        *reinterpret_cast<u32x4*>(p + pstride) = u32x4{sx[0], sy[0], sx[1], sy[1]};
    }
}

template <int P, bool BF>
__device__ __forceinline__ void load4(const char* p, long long pstride, float (&v)[4]) {
    typedef typename HT<BF>::v4 V4;
    const V4 hi = *reinterpret_cast<const V4*>(p);
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = (float)hi[q];
    if constexpr (P == 2) {
        const V4 lo = *reinterpret_cast<const V4*>(p + pstride);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] += (float)lo[q];
    }
}

__device__ __forceinline__ float h_sigmoid(float x) {
    const float e = __builtin_amdgcn_exp2f(-1.44269504088896341f * x);
    return __builtin_amdgcn_rcpf(1.0f + e);
}
__device__ __forceinline__ float h_tanh(float x) {   // same formulation as the fp32 kernel (wn_gemm.hip): |err| <= 2e-7
    const float ax = __builtin_fabsf(x);
    const float e = __builtin_amdgcn_exp2f(-2.88539008177792681f * ax);
    const float big = (1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e);
    const float x2 = x * x;
    const float small = ax * (1.0f + x2 * (-0.333333333f + x2 * (0.133333333f + x2 * -0.0539682540f)));
    return __builtin_copysignf(ax < 0.125f ? small : big, x);
}

// The gate of the ONE-PLANE fused forward (wn_fused.hip), whose results are rounded to bf16 / fp16 anyway: five and four vector
// instructions instead of fourteen and four, no branch-free select for small |x|.  x arrives pre-multiplied:
//     tanh(a)    = 1 - 2 / (1 + 2^(2 a log2 e))         xt = a * 2 log2(e)       (2^xt = inf -> 1, = 0 -> -1: no NaN at either end)
//     sigmoid(g) = 1 / (1 + 2^(-g log2 e))              xs = -g * log2(e)
// Absolute error <= 1.3e-7 (the rounding of 2 r near r = 1/2); RELATIVE error of tanh grows as 6e-8 / |a| below |a| ~ 1e-3,
// which is under half an ulp of fp16 storage down to |a| = 2.4e-4 and always under the absolute resolution of the stored z.
// The f16x3 mode and the fp32 path keep h_tanh / h_sigmoid (|err| <= 2e-7 relative to the result near 0 as well).
constexpr float kLog2e = 1.44269504088896341f;
__device__ __forceinline__ float h_tanh_pre(float xt) {
    const float r = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(xt));
    return __builtin_fmaf(-2.0f, r, 1.0f);
}
__device__ __forceinline__ float h_sigmoid_pre(float xs) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(xs));
}

// backward of the gate from what the forward pass keeps: z = tanh(a) sigmoid(g) and s = sigmoid(g).  The tanh is not stored
// (one tensor less to write in the forward gate launch and to keep until backward): t = z / s.
//     da = dz s (1 - t^2) = dz (s - z t)         dg = dz t s (1 - s) = dz z (1 - s)
// s = 0 (sigmoid underflow) has z = 0 and both gradients 0.  An error in the recovered t enters da multiplied by z <= s.
__device__ __forceinline__ void dgate(float dz, float z, float s, float& da, float& dg) {
    const float t = s > 0.0f ? z / s : 0.0f;
    da = dz * (s - z * t);
    dg = dz * z * (1.0f - s);
}

}  // namespace wn
