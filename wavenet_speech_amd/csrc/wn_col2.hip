// dx of a block and dz of the block below it in ONE launch (column-owner form, <= 128 channels, one-plane half modes):
//
//     dx_u = sum_j (W_a,j^T da_u(t - off_j) + W_g,j^T dg_u(t - off_j)) [+ W_proj^T dr_u]        the upper block's input gradient
//     dz_l = W_skip'^T dS + W_res^T dx_u ;   da_l, dg_l = dgate(dz_l, z_l, sigmoid_l)           the lower block's gate gradients
//
// dz is pointwise in time, and its `dr` operand is exactly the dx tile the wave has just computed for the same 32 columns: as in
// the fused forward (wn_fused.hip) the 32 x 32 result tile, converted pairwise to bf16 / f16, IS the B operand of the next
// product (k in accumulator order; the W_res^T segment is packed in that order, HPACK_PERM).  dx is still stored once (the
// weight gradients of the lower block need it), but dz no longer reads it back, and a block costs one backward-data launch
// instead of two: one pipeline fill (~11 us) and 1/6 of dz's bytes less per block.
// Everything else is hcol_kernel's (wn_col_dev.h): activations straight from the series into registers on a refill schedule,
// the two blocks' packed weights as ONE stream of 8 KiB stages through an exact-count LDS ring, flattened units of 32 columns.
#include "wn_col_dev.h"

namespace wn {

namespace {
// ---- schedule (stages are numbered through both products: NST1 of dx, then NT of dz's dS segment, then NT of its dx segment) ----
// fragment stream: the NKS1 fragments of dx, then the 2 NT fragments of dS; window of kCRes, refilled as hcol_kernel does
constexpr int epi2_at(int m, int nst1, int nst) {        // the dgate inputs of row tile m: when 4 (m + 1) k-steps of dS are dead
    const int x = nst1 + 2 * (m + 1);
    return x > nst ? nst : x;
}
constexpr int visible2_at(int X, int nf, int nt, int nst1, int nst) {
    int n = 0;
    for (int j = kCRes; j < nf; ++j) n += (X > 0 && frag_at(j) == X) ? 1 : 0;
    for (int m = 0; m < nt; ++m) n += epi2_at(m, nst1, nst) == X ? 8 : 0;
    return n;
}
constexpr int younger2(int S, int nf, int nt, int nst1, int nst) {
    int n = 0;
    for (int X = (S - kCD + 2 > 0 ? S - kCD + 2 : 0); X <= S; ++X) n += visible2_at(X, nf, nt, nst1, nst);
    // the dx stores (two per row tile) sit between stage nst1 - 1 and stage nst1: younger than the pieces of stages <= nst1 + kCD - 2
    if (S >= nst1 && S <= nst1 + kCD - 2) n += 2 * nt;
    return n;
}
}  // namespace

// (three workgroups per CU; the f16 instantiation of three row tiles spills at 168 registers -- overflow checks -- and takes two)
template <bool BF, int NT, bool HASDR>
__global__ __launch_bounds__(256, (NT == 3 && !BF) ? 2 : 3) void hcol2_kernel(const HCol2Args a) {
    typedef typename HT<BF>::v8 V8;
    typedef typename HT<BF>::v4 V4;
    constexpr int NKS1 = (HASDR ? 10 : 8) * NT, NST1 = NKS1 / 2;
    constexpr int NF = NKS1 + 2 * NT;                         // fragments loaded from memory: dx's operands, then dS
    constexpr int NST = NST1 + 2 * NT;                        // ring stages of the launch
    constexpr int NUP = NF < kCRes ? NF : kCRes;
    static_assert(NF <= kCol2MaxK, "fragment table");
    __shared__ __attribute__((aligned(1024))) char lds[kCD * kCStage];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;

    const int nwg = a.nwg;
    const int per = (nwg + 7) >> 3;
    const int wg = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
    if (wg >= nwg) return;
    const long long ncol = (long long)a.B * a.L;
    const long long c_raw = ((long long)wg * 4 + wave) * 32 + r;
    const bool col_ok = c_raw < ncol;
    const long long c = c_raw < ncol ? c_raw : ncol - 1;
    const int b = (int)(c / a.L);
    const int t = (int)(c - (long long)b * a.L);
    const int ld = a.ld;
    const long long colb = ((long long)h * ld + a.halo + t) * 16;

    // ---- weight ring over the two blocks' streams --------------------------------------------------------------------------
    const unsigned lane16 = lane * 16u;
    auto issue = [&](int stage_no) {
        const char* src = (stage_no < NST1 ? a.wstream1 + (long long)stage_no * kCStage : a.wstream2 + (long long)(stage_no - NST1) * kCStage) +
                          wave * 1024;
        char* dst = lds + (stage_no % kCD) * kCStage + wave * 1024;
#pragma unroll
        for (int p = 0; p < kCPW; ++p) WN_GLDS(src + p * 4096, lane16, dst + p * 4096);
    };
#pragma unroll
    for (int s = 0; s < kCD - 1 && s < NST; ++s) issue(s);

    V8 bfr[NF];
    auto load_frag = [&](int kk) { bfr[kk] = *reinterpret_cast<const V8*>(a.kbase[kk] + (long long)b * a.kustride[kk] + colb); };
#pragma unroll
    for (int kk = 0; kk < NUP; ++kk) load_frag(kk);
    __builtin_amdgcn_s_waitcnt(0x0F70);                       // vmcnt(0) expcnt(7) lgkmcnt(15)
    V4 zin[NT][4], sin_[NT][4];
    const char* zb = a.z.base + (long long)b * a.z.ustride + ((long long)a.halo + t) * 16 + 8 * h;
    const char* sb = a.sg.base + (long long)b * a.sg.ustride + ((long long)a.halo + t) * 16 + 8 * h;
    auto load_epi = [&](int m) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            zin[m][i] = *reinterpret_cast<const V4*>(zb + (long long)(4 * m + i) * ld * 16);
            sin_[m][i] = *reinterpret_cast<const V4*>(sb + (long long)(4 * m + i) * ld * 16);
        }
    };
    auto scheduled_loads = [&](auto x_c) {
        constexpr int X = decltype(x_c)::value;
        if constexpr (X > 0) {
#pragma unroll
            for (int j = kCRes; j < NF; ++j)
                if (frag_at(j) == X) load_frag(j);
        }
#pragma unroll
        for (int m = 0; m < NT; ++m)
            if (epi2_at(m, NST1, NST) == X) load_epi(m);
        __builtin_amdgcn_sched_barrier(0);
    };
    scheduled_loads(std::integral_constant<int, 0>{});
    __builtin_amdgcn_s_barrier();

    const char* a_rd = lds + (h * 128 + r) * 16;
    f32x16 acc[NT];
    auto zero_acc = [&]() {
#pragma unroll
        for (int m = 0; m < NT; ++m)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[m][q] = 0.0f;
    };
    zero_acc();

    auto stage = [&](auto s_c, const V8& b0, const V8& b1) {
        constexpr int S = decltype(s_c)::value;
        constexpr int AHEAD = (NST - 1 - S) < (kCD - 2) ? (NST - 1 - S) : (kCD - 2);
        constexpr int YOUNGER = kCPW * AHEAD + younger2(S, NF, NT, NST1, NST);
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
    unsigned ovf = 0;

    // ---- product 1: dx of the upper block ---------------------------------------------------------------------------------------
    [&]<int... S>(std::integer_sequence<int, S...>) {
        ([&] {
            if constexpr (S > 0) scheduled_loads(std::integral_constant<int, S>{});
            stage(std::integral_constant<int, S>{}, bfr[2 * S], bfr[2 * S + 1]);
        }(), ...);
    }(std::make_integer_sequence<int, NST1>{});
    // its result: stored (the lower block's weight gradients read it), and kept as the B operand of product 2 -- tile m = k-steps
    // 2 m, 2 m + 1 of the dx segment, k in accumulator order
    V8 dxf[NT][2];
    {
        const float osc = a.oscale1;
#pragma unroll
        for (int m = 0; m < NT; ++m) {
            unsigned pk[8];
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                const float v0 = acc[m][2 * d] * osc, v1 = acc[m][2 * d + 1] * osc;
                if constexpr (!BF) ovf |= (!(__builtin_fabsf(v0) <= 65504.0f) || !(__builtin_fabsf(v1) <= 65504.0f)) ? 1u : 0u;
                pk[d] = cpack2<BF>(v0, v1);
            }
            dxf[m][0] = __builtin_bit_cast(V8, u32x4{pk[0], pk[1], pk[2], pk[3]});
            dxf[m][1] = __builtin_bit_cast(V8, u32x4{pk[4], pk[5], pk[6], pk[7]});
            store_tile(a.dx, m, pk);
        }
    }
    zero_acc();

    // ---- product 2: dz of the lower block: K = [dS ; dx] ------------------------------------------------------------------------
    [&]<int... Q>(std::integer_sequence<int, Q...>) {
        ([&] {
            constexpr int S = NST1 + Q;
            scheduled_loads(std::integral_constant<int, S>{});
            stage(std::integral_constant<int, S>{}, bfr[2 * S], bfr[2 * S + 1]);
        }(), ...);
    }(std::make_integer_sequence<int, NT>{});
    [&]<int... Q>(std::integer_sequence<int, Q...>) {
        ([&] {
            constexpr int S = NST1 + NT + Q;
            scheduled_loads(std::integral_constant<int, S>{});
            stage(std::integral_constant<int, S>{}, dxf[Q][0], dxf[Q][1]);
        }(), ...);
    }(std::make_integer_sequence<int, NT>{});
    scheduled_loads(std::integral_constant<int, NST>{});

    {
        const float osc = a.oscale2;
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

hipError_t launch_hcol2(int prec, const HCol2Args& a_in, hipStream_t st) {
    if (a_in.nunit <= 0) return hipSuccess;
    if (prec != HP_BF16 && prec != HP_F16) return hipErrorInvalidValue;
    HCol2Args a = a_in;
    a.nwg = (a.nunit + 3) / 4;
    const unsigned grid = (unsigned)(((a.nwg + 7) / 8) * 8);
    const bool bf = prec == HP_BF16;
#define WN_LAUNCH_COL2(NT_)                                                                                               \
    do {                                                                                                                  \
        if (a.hasdr) {                                                                                                    \
            if (bf) hipLaunchKernelGGL((hcol2_kernel<true, NT_, true>), dim3(grid), dim3(256), 0, st, a);                 \
            else hipLaunchKernelGGL((hcol2_kernel<false, NT_, true>), dim3(grid), dim3(256), 0, st, a);                   \
        } else {                                                                                                          \
            if (bf) hipLaunchKernelGGL((hcol2_kernel<true, NT_, false>), dim3(grid), dim3(256), 0, st, a);                \
            else hipLaunchKernelGGL((hcol2_kernel<false, NT_, false>), dim3(grid), dim3(256), 0, st, a);                  \
        }                                                                                                                 \
        return hipGetLastError();                                                                                         \
    } while (0)
    switch (a.nt) {
        case 1: WN_LAUNCH_COL2(1);
        case 2: WN_LAUNCH_COL2(2);
        case 3: WN_LAUNCH_COL2(3);
        case 4: WN_LAUNCH_COL2(4);
    }
#undef WN_LAUNCH_COL2
    return hipErrorInvalidValue;
}

}  // namespace wn
