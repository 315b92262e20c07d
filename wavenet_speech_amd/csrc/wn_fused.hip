// Fused forward of one residual block for blocks of <= 128 channels in the one-plane half modes (bf16 / f16):
//
//     a, g = dilated convs of x        z = tanh(a) sigmoid(g)        r = W_res z + W_proj x + b        [S += W_skip' z + b']
//
// in ONE launch, with z never leaving the chip between the two products (reference span: modules/block.py:65-79).
//
// How: with <= 128 channels one wavefront can own ALL output channels of its time columns.  A wave takes a UNIT of 32
// consecutive time steps of one utterance and computes, on v_mfma_f32_32x32x16:
//   gate phase(s)   [a ; g] rows of 64 channels at a time (4 accumulator tiles of 32 x 32 = 64 registers), K = taps x Ci.
//                   B operand = the x fragments of the unit, loaded STRAIGHT from the half series into registers: a 16-byte
//                   unit of the layout (8 channels of one time step) is exactly lane (column, k-half)'s fragment, and a
//                   dilated tap is the same load at another column -- no LDS, no sharing needed (every wave owns its columns).
//   gate epilogue   tanh * sigmoid on the accumulators.  The 32 x 32 result tile has its COLUMN on the lane and its rows in
//                   the 16 registers, so converted pairwise to bf16/f16 it IS the B operand of the next product (k = channel,
//                   in accumulator order: the weights of the z segment are packed in that order, HPACK_PERM) -- z goes from
//                   the MFMA result registers back into the MFMA with no lane movement and no LDS.  When training, z and
//                   sigmoid(g) are also stored (16 bytes per lane after a v_permlane32_swap of the two k-halves).
//   res phase       r rows (<= 128) x K = [z ; x(t)], accumulators reused; epilogue stores r.
//   skip phase      (inference) skip rows x K = [z]; epilogue accumulates into the dense fp32 skips_sum.
// The A operand (weights, ~192 KiB per block at 128 channels) is the same for every wave of every workgroup: it streams through
// an LDS ring as one linear sequence of 8 KiB stages (128 rows x 32 channels), staged by LDS-DMA, one counted wait + one
// barrier per stage -- the only thing the four waves of a workgroup share.
//
// HBM bytes per (time step, block) at 128 channels, training: read x 256 B, write z, sg, r 768 B (two launches: 1536 B);
// inference: read x, write r, read-modify-write S.  The kernel is HBM-bound: 196 kflop per column is ~10 us of MFMA per
// block-forward at cfg2 against ~27 us of traffic.
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "wn_half.h"
#include "wn_half_dev.h"

namespace wn {

namespace {
constexpr int kFD = 8;                                  // ring depth: 64 KiB of LDS, two workgroups per CU
constexpr int kFPW = kFStageBytes / (1024 * 4);         // 1 KiB DMA pieces per wave and stage (2)
constexpr int kFInflight = (kFD - 2) * kFPW;

template <bool BF> using FT = HT<BF>;

template <bool BF>
__device__ __forceinline__ f32x16 mfma32(const typename FT<BF>::v8& a, const typename FT<BF>::v8& b, const f32x16& c) {
    if constexpr (BF) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

template <bool BF>
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    typedef typename FT<BF>::t T;
    typedef T T2 __attribute__((ext_vector_type(2)));
    T2 v;
    v[0] = (T)lo;
    v[1] = (T)hi;
    return __builtin_bit_cast(unsigned, v);
}
}  // namespace

template <bool BF>
__global__ __launch_bounds__(256, 2) void hfused_fwd_kernel(const HFusedArgs a) {
    typedef typename FT<BF>::v8 V8;
    __shared__ __attribute__((aligned(1024))) char lds[kFD * kFStageBytes];
    __shared__ __attribute__((aligned(16))) float lbias[4 * kFRows];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;

    // ---- which unit: the workgroups of one XCD (ids congruent mod 8) take a contiguous range of units, so the halo columns a
    // unit's dilated taps share with its neighbours are served by that XCD's L2
    const int nwg = (a.nunit + 3) >> 2;
    const int per = (nwg + 7) >> 3;
    const int wg = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
    if (wg >= nwg) return;                                    // the whole workgroup leaves together
    const int unit = wg * 4 + wave;
    const bool active = unit < a.nunit;                       // wave-uniform; an idle wave still stages, waits and syncs
    const int b = active ? unit / a.units_per_row : 0;
    const int t0 = active ? (unit - b * a.units_per_row) * 32 : 0;
    const int t = t0 + r;
    const bool col_ok = active && t < a.L;
    const int ld = a.ld;

    // ---- accumulator start values (bias / output scale) of the four phases -> LDS ----------------------------------------
    for (int i = tid; i < 4 * kFRows; i += 256) lbias[i] = a.bias[i];

    // ---- the unit's x fragments for the gate product: k-step kk = (tap, 16 channels), straight from the half series -------
    const char* xb = a.x + (long long)b * a.x_ustride + ((long long)a.halo + t) * 16;
    const long long hld = (long long)h * ld * 16;
    V8 xf[kFMaxGateK];
#pragma unroll
    for (int kk = 0; kk < kFMaxGateK; ++kk) {
        if (kk < a.nkg && active) xf[kk] = *reinterpret_cast<const V8*>(xb + (long long)a.xunit[kk] * 16 + hld);
        else {
#pragma unroll
            for (int j = 0; j < 8; ++j) xf[kk][j] = 0;
        }
    }

    // ---- weight ring ------------------------------------------------------------------------------------------------------
    int is_stage = 0;                                         // next stage of the linear weight stream to issue
    const unsigned lane16 = lane * 16u;
    auto issue = [&](int slot) {
        // (a block without a residual output skips the res phase's stages of the stream: jump_at / jump)
        const char* src = a.wstream + (long long)(is_stage + (is_stage >= a.jump_at ? a.jump : 0)) * kFStageBytes + wave * 1024;
        char* dst = lds + slot * kFStageBytes + wave * 1024;
#pragma unroll
        for (int p = 0; p < kFPW; ++p) WN_GLDS(src + p * 4096, lane16, dst + p * 4096);
        if (is_stage + 1 < a.nstage) ++is_stage;             // past the end the last stage is staged again (never read)
    };
#pragma unroll
    for (int s = 0; s < kFD - 1; ++s) issue(s);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // this wave's lbias writes
    __builtin_amdgcn_s_barrier();

    int gs = 0;                                               // stages consumed so far
    const unsigned a_rd = (unsigned)((h * kFRows + r) * 16);  // this lane's fragment of row tile 0, k-step 0 of a stage
    f32x16 acc[4];
    auto init_acc = [&](int phase) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(&lbias[phase * kFRows + 32 * m + 8 * i + 4 * h]);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[m][4 * i + q] = v[q];
            }
    };
    // one stage = two k-steps of 16 channels for the 128 rows of the phase: wait for it, free the previous slot, prefetch
    // the stage kFD - 1 ahead into it, then 8 fragment reads + 8 MFMAs
    auto stage = [&](const V8& b0, const V8& b1) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kFInflight) : "memory");
        __builtin_amdgcn_s_barrier();
        const int slot = gs & (kFD - 1);
        issue((gs + kFD - 1) & (kFD - 1));
        ++gs;
        if (active) {
            const char* st = lds + slot * kFStageBytes + a_rd;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                V8 af[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) af[m] = *reinterpret_cast<const V8*>(st + kk * 4096 + m * 512);
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[m] = mfma32<BF>(af[m], kk ? b1 : b0, acc[m]);
            }
        }
    };

    // a 32 x 32 tile of packed results (8 dwords = 16 channels-in-accumulator-order per lane) -> the half series, 16 bytes per lane:
    // registers 4i..4i+3 are channels 8i + 4h .. + 3 of the tile = half a unit; v_permlane32_swap of groups (2p, 2p + 1) gives lanes
    // 0-31 the whole unit of group 2p and lanes 32-63 that of group 2p + 1
    auto store_tile = [&](const HDst& d, int tile, const unsigned (&pk)[8]) {
        char* base = d.base + (long long)b * d.ustride + ((long long)(4 * tile + h) * ld + a.halo + t) * 16;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            unsigned ax = pk[4 * p], ay = pk[4 * p + 1], bx = pk[4 * p + 2], by = pk[4 * p + 3];
            const auto sx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
            const auto sy = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
            const u32x4 v = {sx[0], sy[0], sx[1], sy[1]};
            if (col_ok) *reinterpret_cast<u32x4*>(base + (long long)(2 * p) * ld * 16) = v;
        }
    };

    // ---- gate phases: 64 z channels (two tiles) at a time ---------------------------------------------------------------------
    V8 zf[4][2];                                              // z as the B operand of the next products: [tile][k-step]
    const float og = a.osc_gate;
    const bool ragged_co = (a.co & 31) != 0;
    [&]<int... HF>(std::integer_sequence<int, HF...>) {
        ([&] {
            constexpr int hf = HF;
            if (2 * hf < a.nzt) {
                init_acc(hf);
#pragma unroll
                for (int s = 0; s < kFMaxGateK / 2; ++s)
                    if (2 * s < a.nkg) stage(xf[2 * s], xf[2 * s + 1]);
                if (active) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int tile = 2 * hf + j;
                        unsigned zpk[8], spk[8];
#pragma unroll
                        for (int d = 0; d < 8; ++d) {
                            float zv[2], sv[2];
#pragma unroll
                            for (int e = 0; e < 2; ++e) {
                                const int q = 2 * d + e;
                                const float ta = h_tanh(acc[2 * j][q] * og);
                                float s_ = h_sigmoid(acc[2 * j + 1][q] * og);
                                float z_ = ta * s_;
                                if (ragged_co) {              // pad channels stay exactly zero in the series
                                    const int ch = 32 * tile + (q & 3) + 8 * (q >> 2) + 4 * h;
                                    if (ch >= a.co) { s_ = 0.0f; z_ = 0.0f; }
                                }
                                zv[e] = z_; sv[e] = s_;
                            }
                            zpk[d] = pack2<BF>(zv[0], zv[1]);
                            spk[d] = pack2<BF>(sv[0], sv[1]);
                        }
                        zf[2 * hf + j][0] = __builtin_bit_cast(V8, u32x4{zpk[0], zpk[1], zpk[2], zpk[3]});
                        zf[2 * hf + j][1] = __builtin_bit_cast(V8, u32x4{zpk[4], zpk[5], zpk[6], zpk[7]});
                        if (tile < a.nzt) {
                            if (a.z.base) store_tile(a.z, tile, zpk);
                            if (a.sg.base) store_tile(a.sg, tile, spk);
                        }
                    }
                }
            }
        }(), ...);
    }(std::integer_sequence<int, 0, 1>{});

    // ---- res phase: r = [W_res | W_proj] [z ; x(t)] ---------------------------------------------------------------------------------
    if (a.do_res) {
        V8 xp[8];                                             // x(t): the projection's B operand (often no tap has offset 0)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            if (ks < a.nci16 && active) xp[ks] = *reinterpret_cast<const V8*>(xb + (long long)(2 * ks) * ld * 16 + hld);
            else {
#pragma unroll
                for (int j = 0; j < 8; ++j) xp[ks][j] = 0;
            }
        }
        init_acc(2);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < a.nzt) stage(zf[j][0], zf[j][1]);
#pragma unroll
        for (int s = 0; s < 4; ++s)
            if (2 * s < a.nci16) stage(xp[2 * s], xp[2 * s + 1]);
        if (active) {
            const float orr = a.osc_res;
            unsigned ovf = 0;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                if (m < a.nzt) {
                    unsigned pk[8];
#pragma unroll
                    for (int d = 0; d < 8; ++d) {
                        const float v0 = acc[m][2 * d] * orr, v1 = acc[m][2 * d + 1] * orr;
                        if constexpr (!BF) ovf |= (!(__builtin_fabsf(v0) <= 65504.0f) || !(__builtin_fabsf(v1) <= 65504.0f)) ? 1u : 0u;
                        pk[d] = pack2<BF>(v0, v1);
                    }
                    store_tile(a.r, m, pk);
                }
            }
            if constexpr (!BF) {
                if (ovf && a.flag) atomicOr(a.flag, 1u);
            }
        }
    }

    // ---- skip phase (inference): S (+)= W_skip' z + b' ------------------------------------------------------------------------------
    if (a.do_skip) {
        init_acc(3);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < a.nzt) stage(zf[j][0], zf[j][1]);
        if (active) {
            const float os = a.osc_skip;
            float* sp = a.skip + ((long long)b * a.skip_rows + 4 * h) * a.L + t;
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row0 = 32 * m + 8 * i + 4 * h;          // rows row0 .. row0 + 3
                    float old[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        old[q] = (a.skip_accum && col_ok && row0 + q < a.skip_rows) ? sp[(long long)(32 * m + 8 * i + q) * a.L] : 0.0f;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (col_ok && row0 + q < a.skip_rows) sp[(long long)(32 * m + 8 * i + q) * a.L] = acc[m][4 * i + q] * os + old[q];
                }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the surplus stages of the ring land before the LDS is released
}

hipError_t launch_hfused_fwd(int prec, const HFusedArgs& a, hipStream_t st) {
    if (a.nunit <= 0 || a.nstage <= 0) return hipSuccess;
    const int nwg = (a.nunit + 3) / 4;
    const unsigned grid = (unsigned)(((nwg + 7) / 8) * 8);
    if (prec == HP_BF16) hipLaunchKernelGGL((hfused_fwd_kernel<true>), dim3(grid), dim3(256), 0, st, a);
    else if (prec == HP_F16) hipLaunchKernelGGL((hfused_fwd_kernel<false>), dim3(grid), dim3(256), 0, st, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace wn
