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

// NZT = z tiles of 32 channels = round_up(C, 32) / 32 (Ci and Co round to the same value, two taps): every loop bound below
// is a compile-time constant and the stream of stages is straight-line code.  That is not cosmetic: with a branch anywhere
// between a global load and its use hipcc's wait insertion falls back to `s_waitcnt vmcnt(0)` -- a drain of the whole LDS-DMA
// ring in front of every stage (the first version: 1100 cycles per stage instead of ~300) -- and with a branch between the
// MFMAs of a stage it serialises fragment read -> wait -> MFMA.
// MODE 0 (training): z and sigmoid(g) are stored (both required), no skip phase (training forms skips_sum afterwards from every
// block's z).  MODE 1 (inference): sigmoid(g) is not stored, z only if the caller wants it, skips_sum accumulated here when
// asked for.  MODE 2: a stand-alone block that wants everything (z, sigmoid(g), r and its skip output).
template <bool BF, int NZT, int MODE>
__global__ __launch_bounds__(256, 2) void hfused_fwd_kernel(const HFusedArgs a) {
    constexpr bool TRAIN = MODE != 1, SKIP = MODE != 0;
    typedef typename FT<BF>::v8 V8;
    constexpr int NCI16 = 2 * NZT, NKG = 4 * NZT;             // k-steps of 16 channels per tap / of the gate product
    constexpr int NGH = (NZT + 1) / 2;                        // gate halves of 64 channels
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
    // a wave past the last unit recomputes the last one and stores nothing (no branch around the wave's work)
    const int unit_raw = wg * 4 + wave;
    const int unit = unit_raw < a.nunit ? unit_raw : a.nunit - 1;
    const int b = unit / a.units_per_row;
    const int t0 = (unit - b * a.units_per_row) * 32;
    const int t = t0 + r;
    const bool col_ok = unit_raw < a.nunit && t < a.L && !(a.dbg & 1);
    const int ld = a.ld;

    // ---- accumulator start values (bias / output scale) of the four phases -> LDS ----------------------------------------
    {
        const f32x4 bv = reinterpret_cast<const f32x4*>(a.bias)[tid & 127];
        if (tid < 128) reinterpret_cast<f32x4*>(lbias)[tid] = bv;
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
        is_stage = is_stage + 1 < a.nstage ? is_stage + 1 : is_stage;   // past the end the last stage is staged again (never read)
    };
#pragma unroll
    for (int s = 0; s < kFD - 1; ++s) issue(s);

    // ---- the unit's x fragments, straight from the half series: gate k-step kk = (tap, 16 channels), and x(t) for the
    // projection (often no tap has offset 0).  hipcc inserts its own waits for these registers and counts only the loads IT
    // knows: with the (inline-asm) DMA pieces of the ring in flight behind them, its `vmcnt(15) .. vmcnt(0)` ladder over the
    // first eight stages drained the ring step by step.  So the loads are issued AFTER the ring's prologue and waited for
    // right here with a wait the compiler can see: from then on it knows of no pending load and inserts none.  (The wait
    // also covers the prologue's seven stages, which were issued first and are needed next anyway.)
    const char* xb = a.x + (long long)b * a.x_ustride + ((long long)a.halo + t) * 16 + (long long)h * ld * 16;
    V8 xf[NKG], xp[NCI16];
#pragma unroll
    for (int kk = 0; kk < NKG; ++kk) xf[kk] = *reinterpret_cast<const V8*>(xb + (long long)a.xunit[kk] * 16);
#pragma unroll
    for (int ks = 0; ks < NCI16; ++ks) xp[ks] = *reinterpret_cast<const V8*>(xb + (long long)(2 * ks) * ld * 16);
    __builtin_amdgcn_s_waitcnt(0x0F70);                       // vmcnt(0) expcnt(7) lgkmcnt(15)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // this wave's lbias writes
    __builtin_amdgcn_s_barrier();

    unsigned long long tstamp[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // measurement (WN_FUSED_STAMPS): s_memtime at the phase boundaries of wave 0
    auto stamp = [&](int i) { if (a.stamps) tstamp[i] = __builtin_amdgcn_s_memtime(); };
    stamp(0);
    int gs = 0;                                               // stages consumed so far
    const char* a_rd = lds + (h * kFRows + r) * 16;           // this lane's fragment of row tile 0, k-step 0 of slot 0
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
    // One stage = two k-steps of 16 channels for the 128 rows of the phase: wait for it, free the previous slot, prefetch the
    // stage kFD - 1 ahead into it, then 8 fragment reads in flight and 8 MFMAs.
    // EXTRA = epilogue stores issued since this stage's pieces were: they sit in the same in-order vmcnt queue, so counted
    // exactly the wait never depends on a store being acknowledged by HBM.  Only stores that are CERTAINLY issued are counted
    // (store_tile is unconditional: masked lanes write to a dump line); anything uncounted makes a wait stricter, never weaker.
    auto stage = [&](auto extra_c, const V8& b0, const V8& b1) {
        constexpr int EXTRA = decltype(extra_c)::value;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kFInflight + EXTRA) : "memory");
        __builtin_amdgcn_s_barrier();
        const int slot = gs & (kFD - 1);
        issue((gs + kFD - 1) & (kFD - 1));
        ++gs;
        const char* st = a_rd + slot * kFStageBytes;
        // all eight fragment reads in flight, then the MFMAs, each waiting for its own fragment only (lgkmcnt counts down)
        V8 af[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) af[i] = *reinterpret_cast<const V8*>(st + (i >> 2) * 4096 + (i & 3) * 512);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i & 3] = mfma32<BF>(af[i], (i >> 2) ? b1 : b0, acc[i & 3]);
        __builtin_amdgcn_sched_barrier(0);
    };
    // stage number `s` of a phase whose predecessor epilogue issued NST stores: they are younger than the pieces of the first
    // kFD - 1 stages after it
    #define WN_EXTRA(s, NST) std::integral_constant<int, ((s) < kFD - 1 ? (NST) : 0)>{}

    // a 32 x 32 tile of packed results (8 dwords = 16 channels-in-accumulator-order per lane) -> the half series, 16 bytes per lane:
    // registers 4i..4i+3 are channels 8i + 4h .. + 3 of the tile = half a unit; v_permlane32_swap of groups (2p, 2p + 1) gives lanes
    // 0-31 the whole unit of group 2p and lanes 32-63 that of group 2p + 1.  Two store instructions, always issued.
    char* const dump = a.dump + lane * 16;                    // columns past the end of the utterance: same instruction, harmless address
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

    // ---- gate phases: 64 z channels (two tiles) at a time ---------------------------------------------------------------------
    V8 zf[NZT][2];                                            // z as the B operand of the next products: [tile][k-step]
    const float kt = a.osc_gate * (2.0f * kLog2e), ks = -a.osc_gate * kLog2e;   // accumulator -> exponent of the tanh / sigmoid
    const bool ragged_co = (a.co & 31) != 0;
    const bool keep_z = TRAIN || a.z.base != nullptr;
    constexpr int GST = NKG / 2;                              // stages per gate half
    [&]<int... HF>(std::integer_sequence<int, HF...>) {
        ([&] {
            constexpr int hf = HF;
            constexpr int TILES = NZT - 2 * hf >= 2 ? 2 : 1;  // z tiles of this half
            constexpr int PREV = (hf == 0 || !TRAIN) ? 0 : 8; // counted stores of the previous half's epilogue (two tiles, z and sg)
            init_acc(hf);
            [&]<int... S>(std::integer_sequence<int, S...>) {
                (stage(WN_EXTRA(S, PREV), xf[2 * S], xf[2 * S + 1]), ...);
            }(std::make_integer_sequence<int, GST>{});
            if (hf == 0) stamp(1);
            // (the pad-channel masking of channel counts that are not multiples of 32 lives in its own copy of the epilogue: a
            // per-element branch or select in the common path costs more than the tanh)
            auto epilogue = [&](auto ragged_c) {
                constexpr bool RAGGED = decltype(ragged_c)::value;
#pragma unroll
                for (int j = 0; j < TILES; ++j) {
                    const int tile = 2 * hf + j;
                    unsigned zpk[8], spk[8];
#pragma unroll
                    for (int d = 0; d < 8; ++d) {
                        float zv[2], sv[2];
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const int q = 2 * d + e;
                            const float ta = h_tanh_pre(acc[2 * j][q] * kt);
                            float s_ = h_sigmoid_pre(acc[2 * j + 1][q] * ks);
                            float z_ = ta * s_;
                            if constexpr (RAGGED) {           // pad channels stay exactly zero in the series
                                const int ch = 32 * tile + (q & 3) + 8 * (q >> 2) + 4 * h;
                                s_ = ch < a.co ? s_ : 0.0f;
                                z_ = ch < a.co ? z_ : 0.0f;
                            }
                            zv[e] = z_; sv[e] = s_;
                        }
                        zpk[d] = pack2<BF>(zv[0], zv[1]);
                        spk[d] = pack2<BF>(sv[0], sv[1]);
                    }
                    zf[2 * hf + j][0] = __builtin_bit_cast(V8, u32x4{zpk[0], zpk[1], zpk[2], zpk[3]});
                    zf[2 * hf + j][1] = __builtin_bit_cast(V8, u32x4{zpk[4], zpk[5], zpk[6], zpk[7]});
                    if constexpr (TRAIN) {
                        store_tile(a.z, tile, zpk);
                        store_tile(a.sg, tile, spk);
                    } else {
                        if (keep_z) store_tile(a.z, tile, zpk);   // (not counted by the waits: they are only stricter for it)
                    }
                }
            };
            if (ragged_co) epilogue(std::true_type{});
            else epilogue(std::false_type{});
            stamp(2 + hf);
        }(), ...);
    }(std::make_integer_sequence<int, NGH>{});
    // stores of the LAST gate epilogue when z and sg are both kept: 4 per tile
    constexpr int LAST_TILES = NZT - 2 * (NGH - 1) >= 2 ? 2 : 1;

    // ---- res phase: r = [W_res | W_proj] [z ; x(t)] ---------------------------------------------------------------------------------
    if (a.do_res) {
        init_acc(2);
        constexpr int PREV = TRAIN ? 4 * LAST_TILES : 0;
        [&]<int... S>(std::integer_sequence<int, S...>) {
            (stage(WN_EXTRA(S, PREV), zf[S][0], zf[S][1]), ...);
        }(std::make_integer_sequence<int, NZT>{});
        [&]<int... S>(std::integer_sequence<int, S...>) {
            (stage(WN_EXTRA(NZT + S, PREV), xp[2 * S], xp[2 * S + 1]), ...);
        }(std::make_integer_sequence<int, NZT>{});
        stamp(4);
        const float orr = a.osc_res;
        unsigned ovf = 0;
#pragma unroll
        for (int m = 0; m < NZT; ++m) {
            unsigned pk[8];
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                const float v0 = acc[m][2 * d] * orr, v1 = acc[m][2 * d + 1] * orr;
                if constexpr (!BF) ovf |= (!(__builtin_fabsf(v0) <= 65504.0f) || !(__builtin_fabsf(v1) <= 65504.0f)) ? 1u : 0u;
                pk[d] = pack2<BF>(v0, v1);
            }
            store_tile(a.r, m, pk);
        }
        if constexpr (!BF) {
            if (ovf && a.flag && col_ok) atomicOr(a.flag, 1u);
        }
        stamp(5);
    }

    // ---- skip phase (inference): S (+)= W_skip' z + b' ------------------------------------------------------------------------------
    if constexpr (SKIP) {
        if (a.do_skip) {
            init_acc(3);
            // (stores of earlier epilogues are not counted here: stricter waits, an inference-only phase)
            [&]<int... S>(std::integer_sequence<int, S...>) {
                (stage(WN_EXTRA(S, 0), zf[S][0], zf[S][1]), ...);
            }(std::make_integer_sequence<int, NZT>{});
            const float os = a.osc_skip;
            // wave-uniform row base in SGPRs + one 32-bit per-lane offset: no vector address arithmetic per access
            float* const ubase = a.skip + ((long long)b * a.skip_rows) * a.L + t0;
            const int loff = 4 * h * a.L + r;
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row0 = 32 * m + 8 * i + 4 * h;          // rows row0 .. row0 + 3
                    float old[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float* u = ubase + (long long)(32 * m + 8 * i + q) * a.L;
                        old[q] = (a.skip_accum && col_ok && row0 + q < a.skip_rows) ? u[loff] : 0.0f;
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float* u = ubase + (long long)(32 * m + 8 * i + q) * a.L;
                        if (col_ok && row0 + q < a.skip_rows) u[loff] = acc[m][4 * i + q] * os + old[q];
                    }
                }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the surplus stages of the ring land before the LDS is released
    stamp(7);
    if (a.stamps && wave == 0 && lane == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) a.stamps[(long long)blockIdx.x * 8 + i] = tstamp[i];
    }
    #undef WN_EXTRA
}

unsigned long long* g_fused_stamps = nullptr;
int g_fused_stamp_wgs = 0;

hipError_t launch_hfused_fwd(int prec, const HFusedArgs& a_in, hipStream_t st) {
    if (a_in.nunit <= 0 || a_in.nstage <= 0) return hipSuccess;
    HFusedArgs a = a_in;
    static const int dbg = getenv("WN_FUSED_DBG") ? atoi(getenv("WN_FUSED_DBG")) : 0;   // measurement: 1 no stores, 2 no tanh/sigmoid, 4 no MFMA
    a.dbg = dbg;
    static unsigned long long* stamps = nullptr;   // measurement (WN_FUSED_STAMPS=1): per-workgroup phase stamps, read back by wn_debug_fused_stamps
    static const bool want = getenv("WN_FUSED_STAMPS") != nullptr;
    if (want) {
        if (!stamps) (void)hipMalloc(&stamps, sizeof(unsigned long long) * 8 * 65536);
        a.stamps = stamps;
        g_fused_stamps = stamps;
        g_fused_stamp_wgs = ((a.nunit + 3) / 4 + 7) / 8 * 8;
    }
    const int nwg = (a.nunit + 3) / 4;
    const unsigned grid = (unsigned)(((nwg + 7) / 8) * 8);
    const bool bf = prec == HP_BF16;
    if (prec != HP_BF16 && prec != HP_F16) return hipErrorInvalidValue;
    const int mode = a.sg.base == nullptr ? 1 : (a.do_skip ? 2 : 0);   // sg stored <=> z stored (the API layer checks)
#define WN_LAUNCH_FUSED(N)                                                                                    \
    do {                                                                                                      \
        if (bf && mode == 0) hipLaunchKernelGGL((hfused_fwd_kernel<true, N, 0>), dim3(grid), dim3(256), 0, st, a);       \
        else if (bf && mode == 1) hipLaunchKernelGGL((hfused_fwd_kernel<true, N, 1>), dim3(grid), dim3(256), 0, st, a);  \
        else if (bf) hipLaunchKernelGGL((hfused_fwd_kernel<true, N, 2>), dim3(grid), dim3(256), 0, st, a);               \
        else if (mode == 0) hipLaunchKernelGGL((hfused_fwd_kernel<false, N, 0>), dim3(grid), dim3(256), 0, st, a);       \
        else if (mode == 1) hipLaunchKernelGGL((hfused_fwd_kernel<false, N, 1>), dim3(grid), dim3(256), 0, st, a);       \
        else hipLaunchKernelGGL((hfused_fwd_kernel<false, N, 2>), dim3(grid), dim3(256), 0, st, a);                      \
    } while (0)
    switch (a.nzt) {
        case 1: WN_LAUNCH_FUSED(1); break;
        case 2: WN_LAUNCH_FUSED(2); break;
        case 3: WN_LAUNCH_FUSED(3); break;
        case 4: WN_LAUNCH_FUSED(4); break;
        default: return hipErrorInvalidValue;
    }
#undef WN_LAUNCH_FUSED
    return hipGetLastError();
}

}  // namespace wn
// measurement only (tools/fused_stamps.py; not in include/wavenet_amd.h): copy the phase stamps of the last fused launch
extern "C" int wn_debug_fused_stamps(unsigned long long* host, int max_wgs) {
    if (!wn::g_fused_stamps) return 0;
    const int n = wn::g_fused_stamp_wgs < max_wgs ? wn::g_fused_stamp_wgs : max_wgs;
    if (hipMemcpy(host, wn::g_fused_stamps, sizeof(unsigned long long) * 8 * n, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return n;
}
