// hcol_kernel: column-owner streaming GEMM (shared by wn_col.hip and wn_col_conv.hip).
//
// Backward-data GEMMs of a residual block of <= 128 channels in the one-plane half modes (bf16 / f16), column-owner form:
//
//     dz = W_skip'^T dS + W_res^T dr ;  da = dz s (1 - t^2),  dg = dz t s (1 - s)        (HEPI_DGATE, "dz")
//     dx = sum_j (W_a,j^T da(t - off_j) + W_g,j^T dg(t - off_j)) + W_proj^T dr           (HEPI_STORE, "dx")
//
// (the autograd of reference modules/block.py:65-79).  At these widths the products are HBM-bound -- 5 to 6 channel vectors
// move per column for 2 x 128 x 256..640 flop -- and the LDS-tiled hgemm_kernel, built for K loops long enough to hide a
// tile's prologue and epilogue, spends its time in latency: a 128 x 256 tile is a serial chain (stage ring -> K loop -> epilogue
// loads -> stores) with two workgroups per CU to overlap it, and 544 tiles on 512 slots are two rounds of it (48 us per launch
// at cfg2 where the bytes need 25).  Here, as in wn_fused.hip, ONE wavefront owns all output rows of a UNIT of 32 columns:
//   * the unit's B operand comes STRAIGHT from the half series into registers -- a 16-byte unit of the layout is lane
//     (column, k-half)'s fragment of v_mfma_f32_32x32x16, a dilated tap is the same load at another column -- all of it
//     16 KiB in flight per wave from the start and refilled as it is consumed, twelve waves per CU, no LDS and no barrier
//     for activations;
//   * the A operand (the block's transposed weights, <= 160 KiB, the same for every wave of the launch) streams through the
//     LDS ring of wn_fused.hip: 8 KiB stages, LDS-DMA, one counted wait + one barrier per stage;
//   * units are 32 consecutive VALID columns of the flattened (utterance, time) space, so no unit is ragged except the last
//     (lanes carry their own (b, t)); pad channels come out as exact zeros because their weight rows are zero.
// The packed weights are hgemm_kernel's own (plans KA / KB of wn_half_api.hip with 128-row slabs, 16-channel k-steps): a k-step
// image is [k-half][128 rows][16 B] = 4 KiB, consecutive k-steps are consecutive in memory, two of them are one ring stage.
#pragma once
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "wn_half.h"
#include "wn_half_dev.h"

namespace wn {

namespace {
constexpr int kCD = 6;                                  // ring depth: 48 KiB of LDS, three workgroups per CU
constexpr int kCStage = 8192;                           // two k-steps of 16 channels for 128 rows
constexpr int kCPW = kCStage / (1024 * 4);              // 1 KiB DMA pieces per wave and stage (2)
constexpr int kCRes = 16;                               // register budget of the activations, in k-steps (4 registers each)

template <bool BF>
__device__ __forceinline__ f32x16 cmfma(const typename HT<BF>::v8& a, const typename HT<BF>::v8& b, const f32x16& c) {
    if constexpr (BF) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

template <bool BF>
__device__ __forceinline__ unsigned cpack2(float lo, float hi) {
    typedef typename HT<BF>::t T;
    typedef T T2 __attribute__((ext_vector_type(2)));
    T2 v;
    v[0] = (T)lo;
    v[1] = (T)hi;
    return __builtin_bit_cast(unsigned, v);
}

// ---- the load schedule: which compiler-visible loads are issued in front of stage X -----------------------------------------
// A wave keeps at most kCRes k-steps' worth of activation registers (64): the first kCRes B fragments are loaded before the K
// loop, fragment j >= kCRes in front of the stage at which fragment j - kCRes has been consumed (groups of four: every second
// stage), and the dgate epilogue's inputs of row tile m (z and sigmoid: 8 loads, 16 registers = 4 k-steps' worth) as soon as
// enough fragments are dead.  X == NST means "after the K loop".
constexpr int frag_at(int j) { return j < kCRes ? 0 : 2 * ((j - kCRes) / 4 + 1); }
constexpr int epi_at(int m, int nks, int nst) {
    int x = (nks + 4 * (m + 1) - kCRes + 1) / 2;
    x = x < 0 ? 0 : x;
    return x > nst ? nst : x;
}
// epilogues: HEPI_STORE (dx), HEPI_DGATE (dz), and the two halves of HEPI_LEAKY (wn_half.h) as separate instantiations:
constexpr int kCEpiLeakyFwd = 100;      //   y = leaky((acc * oscale + bias) * oscale2)          (a 1x1 conv kept in the series)
constexpr int kCEpiLeakyBwd = 101;      //   dx = acc * oscale * (signbit(mask) ? leaky : 1)     (its backward-data / dx of the stack's first block)
constexpr int epi_loads(int epi) {      // epilogue input loads per row tile: z and sigmoid (4 row groups each) / bias (4 x float4) / mask (4)
    return epi == HEPI_DGATE ? 8 : ((epi == kCEpiLeakyFwd || epi == kCEpiLeakyBwd) ? 4 : 0);
}
constexpr int visible_at(int X, int nks, int nt, int epi) {   // loads issued in front of stage X (X >= 0; X = 0: after the prologue's wait)
    int n = 0;
    for (int j = kCRes; j < nks; ++j) n += (X > 0 && frag_at(j) == X) ? 1 : 0;
    for (int m = 0; m < nt; ++m) n += epi_at(m, nks, nks / 2) == X ? epi_loads(epi) : 0;
    return n;
}
constexpr int younger_visible(int S, int nks, int nt, int epi) {   // ... that are younger than the ring pieces of stage S at its wait
    int n = 0;
    for (int X = (S - kCD + 2 > 0 ? S - kCD + 2 : 0); X <= S; ++X) n += visible_at(X, nks, nt, epi);
    return n;
}
}  // namespace

// NT = row tiles of 32 output channels (1..4), NKS = 16-channel k-steps (even; <= kColMaxK): every loop bound is a compile-time
// constant and the stream of stages is straight-line code (a branch between a load and its use makes hipcc drain the ring:
// wn_fused.hip).  Waits: the ring's are exact counts over the one in-order vmcnt queue -- DMA pieces (inline asm, invisible to
// the compiler) and the compiler-visible loads of the schedule above.  <= 168 registers, 48 KiB of LDS: three workgroups per CU
// (the launch is a few rounds of latency-bound workgroups: at cfg2 1025 workgroups are 3 rounds on 512 slots, 2 on 768).
template <bool BF, int NT, int NKS, int EPI>
__global__ __launch_bounds__(256, 3) void hcol_kernel(const HColArgs a) {
    static_assert(NKS % 2 == 0 && NKS <= kColMaxK && NT >= 1 && NT <= 4, "shape");
    static_assert(EPI == HEPI_DGATE || EPI == HEPI_STORE || EPI == kCEpiLeakyFwd || EPI == kCEpiLeakyBwd, "epilogue");
    typedef typename HT<BF>::v8 V8;
    typedef typename HT<BF>::v4 V4;
    constexpr bool DG = EPI == HEPI_DGATE;
    constexpr int NST = NKS / 2;                              // ring stages of the launch
    constexpr int NUP = NKS < kCRes ? NKS : kCRes;            // fragments loaded before the K loop
    __shared__ __attribute__((aligned(1024))) char lds[kCD * kCStage];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;

    // ---- which unit: the workgroups of one XCD (ids congruent mod 8) take a contiguous range of units, so the columns a
    // unit's dilated taps share with its neighbours are served by that XCD's L2
    const int nwg = a.nwg;
    const int per = (nwg + 7) >> 3;
    const int wg = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
    if (wg >= nwg) return;                                    // the whole workgroup leaves together
    const long long ncol = (long long)a.B * a.L;
    const long long c_raw = ((long long)wg * 4 + wave) * 32 + r;   // flattened valid column of this lane
    const bool col_ok = c_raw < ncol && !(a.dbg & 1);
    const long long c = c_raw < ncol ? c_raw : ncol - 1;      // lanes past the end recompute the last column and store nothing
    const int b = (int)(c / a.L);
    const int t = (int)(c - (long long)b * a.L);
    const int ld = a.ld;
    const long long colb = ((long long)h * ld + a.halo + t) * 16;   // lane's unit inside a k-step's two channel groups

    // ---- weight ring: stage s of the stream lives in slot s mod kCD; exactly the NST stages are staged (no surplus: every
    // wait below counts the pieces that really are in flight) ------------------------------------------------------------------
    const unsigned lane16 = lane * 16u;
    auto issue = [&](int stage_no) {
        const char* src = a.wstream + (long long)stage_no * kCStage + wave * 1024;
        char* dst = lds + (stage_no % kCD) * kCStage + wave * 1024;
#pragma unroll
        for (int p = 0; p < kCPW; ++p) WN_GLDS(src + p * 4096, lane16, dst + p * 4096);
    };
#pragma unroll
    for (int s = 0; s < kCD - 1 && s < NST; ++s) issue(s);

    // ---- the first B fragments (after the ring's prologue, waited for with a wait the compiler can see: wn_fused.hip) --------
    V8 bfr[NKS];
    auto load_frag = [&](int kk) { bfr[kk] = *reinterpret_cast<const V8*>(a.kbase[kk] + (long long)b * a.kustride[kk] + colb); };
#pragma unroll
    for (int kk = 0; kk < NUP; ++kk) load_frag(kk);
    __builtin_amdgcn_s_waitcnt(0x0F70);                       // vmcnt(0) expcnt(7) lgkmcnt(15)
    // the epilogue's inputs of row tile m (dgate: z and sigmoid; leaky forward: the bias; leaky backward: the stored activation
    // whose sign bit is the mask): issued per the schedule, used after the K loop (the compiler waits for them there)
    constexpr bool EIN = epi_loads(EPI) > 0;
    constexpr bool LFWD = EPI == kCEpiLeakyFwd, LBWD = EPI == kCEpiLeakyBwd;
    V4 zin[(DG || LBWD) ? NT : 1][4], sin_[DG ? NT : 1][4];
    f32x4 bin[LFWD ? NT : 1][4];
    const HDst& zsrc = DG ? a.z : a.mask;
    const char* zb = (DG || LBWD) ? zsrc.base + (long long)b * zsrc.ustride + ((long long)a.halo + t) * 16 + 8 * h : nullptr;
    const char* sb = DG ? a.sg.base + (long long)b * a.sg.ustride + ((long long)a.halo + t) * 16 + 8 * h : nullptr;
    auto load_epi = [&](int m) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (DG || LBWD) zin[m][i] = *reinterpret_cast<const V4*>(zb + (long long)(4 * m + i) * ld * 16);
            if constexpr (DG) sin_[m][i] = *reinterpret_cast<const V4*>(sb + (long long)(4 * m + i) * ld * 16);
            if constexpr (LFWD) bin[m][i] = *reinterpret_cast<const f32x4*>(a.bias + 32 * m + 8 * i + 4 * h);   // rows 32 m + 8 i + 4 h + (0..3)
        }
    };
    auto scheduled_loads = [&](auto x_c) {                    // everything the schedule puts in front of stage X
        constexpr int X = decltype(x_c)::value;
        if constexpr (X > 0) {
#pragma unroll
            for (int j = kCRes; j < NKS; ++j)
                if (frag_at(j) == X) load_frag(j);
        }
        if constexpr (EIN) {
#pragma unroll
            for (int m = 0; m < NT; ++m)
                if (epi_at(m, NKS, NST) == X) load_epi(m);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    scheduled_loads(std::integral_constant<int, 0>{});
    __builtin_amdgcn_s_barrier();

    const char* a_rd = lds + (h * 128 + r) * 16;              // this lane's fragment of row tile 0, k-step 0 of slot 0
    f32x16 acc[NT];
#pragma unroll
    for (int m = 0; m < NT; ++m)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[m][q] = 0.0f;

    // One stage = two k-steps for the NT row tiles: wait for it, free the previous slot, prefetch the stage kCD - 1 ahead into
    // it, then per k-step NT fragment reads in flight and NT MFMAs.  The wait of stage S counts what is younger than its pieces
    // in the in-order queue: the pieces of the (at most kCD - 2) stages after it and the visible loads issued since.
    auto stage = [&](auto s_c, const V8& b0, const V8& b1) {
        constexpr int S = decltype(s_c)::value;
        constexpr int AHEAD = (NST - 1 - S) < (kCD - 2) ? (NST - 1 - S) : (kCD - 2);
        constexpr int YOUNGER = kCPW * AHEAD + younger_visible(S, NKS, NT, EPI);
        static_assert(YOUNGER < 64, "vmcnt is a 6-bit counter");
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNGER) : "memory");
        __builtin_amdgcn_s_barrier();
        if constexpr (S + kCD - 1 < NST) issue(S + kCD - 1);
        const char* st = a_rd + (S % kCD) * kCStage;
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            V8 af[NT];
#pragma unroll
            for (int m = 0; m < NT; ++m) af[m] = *reinterpret_cast<const V8*>(st + k2 * 4096 + m * 512);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < NT; ++m) acc[m] = cmfma<BF>(af[m], k2 ? b1 : b0, acc[m]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    [&]<int... S>(std::integer_sequence<int, S...>) {
        ([&] {
            if constexpr (S > 0) scheduled_loads(std::integral_constant<int, S>{});
            stage(std::integral_constant<int, S>{}, bfr[2 * S], bfr[2 * S + 1]);
        }(), ...);
    }(std::make_integer_sequence<int, NST>{});
    scheduled_loads(std::integral_constant<int, NST>{});     // (epilogue inputs that found no room earlier)

    // ---- epilogue -------------------------------------------------------------------------------------------------------------
    // a 32 x 32 tile of packed results -> the half series, 16 bytes per lane (v_permlane32_swap of the k-halves: wn_fused.hip)
    char* const dump = a.dump + lane * 16;
    auto store_tile = [&](const HDst& d, int tile, const unsigned (&pk)[8]) {
        char* base = d.base + (long long)b * d.ustride + ((long long)(4 * tile + h) * ld + a.halo + t) * 16;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            unsigned ax = pk[4 * p], ay = pk[4 * p + 1], bx = pk[4 * p + 2], by = pk[4 * p + 3];
            const auto sx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
            const auto sy = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
            const u32x4 v = {sx[0], sy[0], sx[1], sy[1]};
            *reinterpret_cast<u32x4*>(col_ok ? base + (long long)(2 * p) * ld * 16 : dump) = v;
        }
    };
    const float osc = a.oscale;
    unsigned ovf = 0;
    if constexpr (EPI == HEPI_STORE) {
#pragma unroll
        for (int m = 0; m < NT; ++m) {
            unsigned pk[8];
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                const float v0 = acc[m][2 * d] * osc, v1 = acc[m][2 * d + 1] * osc;
                if constexpr (!BF) ovf |= (!(__builtin_fabsf(v0) <= 65504.0f) || !(__builtin_fabsf(v1) <= 65504.0f)) ? 1u : 0u;
                pk[d] = cpack2<BF>(v0, v1);
            }
            store_tile(a.dst, m, pk);
        }
    } else if constexpr (LFWD || LBWD) {
        const float osc2 = a.oscale2, slope = a.leaky;
#pragma unroll
        for (int m = 0; m < NT; ++m) {
            unsigned pk[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if constexpr (LFWD) {
                        const float y = (acc[m][4 * i + q] * osc + bin[m][i][q]) * osc2;
                        v[q] = y > 0.0f ? y : -__builtin_fabsf(y * slope);     // y <= 0 is stored with the sign bit set (-0 for 0: torch's x > 0 rule)
                    } else {
                        const float g = acc[m][4 * i + q] * osc;
                        v[q] = __builtin_signbitf((float)zin[m][i][q]) ? g * slope : g;   // the SIGN BIT is the mask (wn_half.h)
                    }
                    if constexpr (!BF) ovf |= !(__builtin_fabsf(v[q]) <= 65504.0f) ? 1u : 0u;
                }
                pk[2 * i] = cpack2<BF>(v[0], v[1]); pk[2 * i + 1] = cpack2<BF>(v[2], v[3]);
            }
            store_tile(a.dst, m, pk);
        }
    } else {
#pragma unroll
        for (int m = 0; m < NT; ++m) {
            unsigned pa[8], pg[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float va[4], vg[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float z_ = (float)zin[m][i][q], s_ = (float)sin_[m][i][q];
                    const float dz = acc[m][4 * i + q] * osc;
                    // dgate() of wn_half_dev.h with the division as a reciprocal (the results are rounded to 8 / 11 bits)
                    const float t_ = s_ > 0.0f ? z_ * __builtin_amdgcn_rcpf(s_) : 0.0f;
                    va[q] = dz * (s_ - z_ * t_);
                    vg[q] = dz * z_ * (1.0f - s_);
                    if constexpr (!BF) ovf |= (!(__builtin_fabsf(va[q]) <= 65504.0f) || !(__builtin_fabsf(vg[q]) <= 65504.0f)) ? 1u : 0u;
                }
                pa[2 * i] = cpack2<BF>(va[0], va[1]); pa[2 * i + 1] = cpack2<BF>(va[2], va[3]);
                pg[2 * i] = cpack2<BF>(vg[0], vg[1]); pg[2 * i + 1] = cpack2<BF>(vg[2], vg[3]);
            }
            store_tile(a.da, m, pa);
            store_tile(a.dg, m, pg);
        }
    }
    if constexpr (!BF) {
        if (ovf && a.flag && col_ok) atomicOr(a.flag, 1u);
    }
}

}  // namespace wn
