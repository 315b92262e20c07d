// Weight gradients on the half-precision MFMAs (f16x3 / f16 / bf16 modes):
//
//     out_p[m][n] = sum_{b, t} A_p[b][m][t] * B_p[b][n][t + off_p]          (p = "pair", e.g. dW_tanh_j = sum da[t] x[t+off_j]^T)
//
// The contraction index is TIME, but the half-series layout (wn_half.h) keeps 8 CHANNELS of one time step together.
// v_mfma_f32_16x16x32 wants, per lane, 8 consecutive k (= time steps) of one row (= channel): that transpose is done by
// the hardware on the way out of LDS with ds_read_b64_tr_b16 (4 time rows x 16 channel columns per 16-lane group,
// delivered column-major), so staging stays pure LDS-DMA of 16-byte units:
//  * one stage = one MFMA k-step = 32 time steps of the A tile (256 channels) and of the B tile (256 channels);
//    a 1 KiB DMA piece = 2 channel groups x 32 time steps = exactly one 16-channel MFMA operand tile of the stage.
//    Inside a piece the unit of (group g, step t = 8 kb + 4 ph + q) sits at 16 (2 (kb >> 1) + ph) + 8 g + 4 (kb & 1) + q:
//    the sixteen units one half-wave's transposed read touches (2 groups x 2 k-blocks x 4 consecutive steps) fall into 16
//    different 16-byte slots = all 64 banks once, the second four steps of a k-block are the same address + 256 B, and
//    four consecutive DMA lanes fetch 64 contiguous bytes.  (The DMA source address is per lane: the permutation is free.)
//  * the 16x16x32 shape, not 32x32x16: at equal cycles per FLOP the chip holds ~2.2 GHz under it against ~1.75 GHz
//    (tools/probes/hwgrad_loop.hip) -- these kernels run at the power limit, so that is +15 % in the loop model.
//    A stage is therefore 64 KiB at f16x3: a two-stage ring, the whole next stage issued during the first half of the
//    current one; five stages of 32 KiB in the one-plane modes.
//  * workgroup = 4 waves (2 x 2), each wave a 128 x 128 tile of fp32 accumulators (8 x 8 MFMA tiles, 256 registers);
//    split-K over time into slabs that wgrad_reduce_kernel (wn_wgrad.hip) sums in a fixed order -- deterministic, no
//    float atomics.
//  * bias gradients (row sums of A over time) come from the A fragments already in registers: v_dot2c_f32_f16 against
//    (1, 1) in the MFMAs' shadow, four of the wave's eight row tiles per wave column.
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "wn_half.h"

namespace wn {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef __bf16 b2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// LDS-DMA issued from inline asm, NOT through __builtin_amdgcn_global_load_lds: with the builtin hipcc (ROCm 7.2) knows an
// LDS write is pending on the vm counter and, because it cannot tell which LDS bytes __builtin_amdgcn_ds_read_tr16_b64 reads,
// puts `s_waitcnt vmcnt(0)` in front of the first transposed read of every k-step -- i.e. it waits for the prefetches just
// issued and the ring degenerates to synchronous staging (measured: SQ_WAIT_ANY 45 % of the wave cycles, 0.85 ms per launch).
// Hidden in asm the DMAs are counted by hand: one `s_waitcnt vmcnt(N)` + s_barrier per k-step (see the loop).  M0 (the LDS
// destination base) is compiler-reserved: it is saved, set and restored inside the one statement.
// Address form: wave-uniform 64-bit base in an SGPR pair + a 32-bit per-lane byte offset (one VGPR that never changes), so a
// piece costs no vector address arithmetic and no address registers.
__device__ __forceinline__ void glds16(const char* ubase, unsigned lane_off, const char* lds_dst) {
    const unsigned dst = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)lds_dst;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane_off), "s"(ubase), "s"(dst) : "memory");
}
#define WN_GLDS(ub, lo, lp) glds16((ub), (lo), (lp))

// 4 time rows x 16 channels of 16-bit elements per 16-lane group, transposed: lane i of the group gets channel i's four
// steps (the builtin lets hipcc count the read and place its wait; an inline-asm read would need both done by hand)
typedef short s4v __attribute__((__vector_size__(4 * sizeof(short))));
__device__ __forceinline__ u32x2 ds_read_tr16(const char* lds_ptr) {
    const s4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(lds_ptr));
    return __builtin_bit_cast(u32x2, v);
}

template <bool BF>
__device__ __forceinline__ float dot_ones(unsigned packed_pair, float acc) {
    if constexpr (BF) {
        const b2 one = {(__bf16)1.0f, (__bf16)1.0f};
        return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(b2, packed_pair), one, acc, false);
    } else {
        const h2 one = {(_Float16)1.0f, (_Float16)1.0f};
        return __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, packed_pair), one, acc, false);
    }
}

template <int P, bool BF>
__global__ __launch_bounds__(256, 1) void hwgrad_kernel(const HWgradArgs a) {
    constexpr int CH = 256;                     // channels of the A tile and of the B tile
    constexpr int KT = 32;                      // time steps per stage = the contraction depth of one 16x16x32 MFMA
    constexpr int T_PLANE = CH * KT * 2;        // bytes of one operand's plane per stage (16 KiB)
    constexpr int A_BYTES = P * T_PLANE, STAGE = 2 * A_BYTES;          // 64 KiB at f16x3, 32 KiB in the one-plane modes
    constexpr int D = (163840 / STAGE) > 5 ? 5 : (163840 / STAGE);     // 2 stages at f16x3, 5 in the one-plane modes
    constexpr int PW = STAGE / 4096;            // 1 KiB pieces per wave per stage (16 / 8)
    constexpr int INFLIGHT = (D - 2) * PW;
    constexpr int WT = 8, NPAIR = WT * WT;      // 8 x 8 accumulator tiles of 16 x 16 per wave
    static_assert(INFLIGHT < 64, "vmcnt is a 6-bit counter");
    __shared__ __attribute__((aligned(1024))) char lds[D * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    int split, tile;
    if (a.xcd_map & 1) {
        const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
        split = (local / a.ntile_total) * 8 + xcd;
        tile = local % a.ntile_total;
    } else {
        split = blockIdx.x / a.ntile_total;
        tile = blockIdx.x - split * a.ntile_total;
    }
    int p = 0;
    while (p + 1 < a.npair && tile >= a.pair[p + 1].tile0) ++p;
    const HWgradPair pr = a.pair[p];
    const int tl = tile - pr.tile0;
    const int tm = tl / pr.nt, tn = tl - tm * pr.nt;

    // this split's range of stages (32 time steps of one utterance each)
    int s_begin = (int)((long long)a.nstep * split / a.nsplit);
    int s_end = (int)((long long)a.nstep * (split + 1) / a.nsplit);
    const int nks = s_end - s_begin;
    if (a.xcd_map & 2) { s_begin = 0; s_end = nks; }   // measurement (WN_HWGRAD_DBG=1): every split re-reads the first range -> operands stay in L2

    // ---- staging: lane u of a piece fetches unit (group g, step t), see the header comment ---------------------------
    const int su_g = (lane >> 3) & 1;
    const int su_t = 8 * (2 * (lane >> 5) + ((lane >> 2) & 1)) + 4 * ((lane >> 4) & 1) + (lane & 3);
    const unsigned lane_src = (unsigned)((su_g * a.ld + su_t) * 16);   // < 2 groups x ld x 16 B: fits 32 bits
    // piece j of a plane = operand tile j = two channel groups; pieces 0..7 are the first HALF of the 256-channel tile, 8..15 the
    // second (another tensor in a composite pair, wn_half.h).  Wave w stages pieces w, w+4, w+8, w+12 of every (operand,
    // plane): PW pieces per wave and stage, in the order (jj, plane, A then B); jj >> 1 = the half, a compile-time constant.
    // A tile may reach past the operand's channels (outputs of those rows / columns are never read): stay inside the tensor
    // by re-reading its last two groups.
    long long goff_a[4], goff_b[4];                 // byte offset of this wave's piece jj inside its half's tensor (loop-invariant)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int hf = jj >> 1, pc = wave + 4 * (jj & 1);
        goff_a[jj] = (long long)min(tm * (CH / 8) + pr.a_gb[hf] + 2 * pc, pr.a_groups[hf] - 2) * a.ld * 16;
        goff_b[jj] = (long long)min(tn * (CH / 8) + pr.b_gb[hf] + 2 * pc, pr.b_groups[hf] - 2) * a.ld * 16;
    }
    int is_step = s_begin;
    const char* is_a[2] = {nullptr, nullptr};
    const char* is_b[2] = {nullptr, nullptr};
    auto stage_sources = [&]() {                    // wave-uniform source bases of the stage to issue next
        const int sb = is_step / a.steps_per_row;
        const int st = (is_step - sb * a.steps_per_row) * KT;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            is_a[hf] = pr.A[hf] + (long long)sb * pr.a_ustride[hf] + ((long long)a.halo + st) * 16;
            is_b[hf] = pr.Bm[hf] + (long long)sb * pr.b_ustride[hf] + ((long long)a.halo + st + pr.off[hf]) * 16;
        }
    };
    auto issue_piece = [&](char* stage, auto pic) {      // stage = LDS image of the stage being filled
        constexpr int PI = decltype(pic)::value;
        constexpr int jj = PI / (2 * P), pl = (PI % (2 * P)) / 2, hf = jj >> 1;
        constexpr bool isB = (PI & 1) != 0;
        const int piece = wave + 4 * jj;
        if constexpr (!isB) {
            WN_GLDS(is_a[hf] + pl * pr.a_pstride[hf] + goff_a[jj], lane_src, stage + pl * T_PLANE + piece * 1024);
        } else {
            WN_GLDS(is_b[hf] + pl * pr.b_pstride[hf] + goff_b[jj], lane_src, stage + A_BYTES + pl * T_PLANE + piece * 1024);
        }
    };
    auto issue_advance = [&]() { if (is_step + 1 < s_end) ++is_step; };   // past the end the last stage is staged again
    auto issue_stage = [&](int slot) {
        stage_sources();
        char* stage = lds + slot * STAGE;
        [&]<int... I>(std::integer_sequence<int, I...>) { (issue_piece(stage, std::integral_constant<int, I>{}), ...); }
        (std::make_integer_sequence<int, PW>{});
        issue_advance();
    };

    f32x4 acc[WT][WT];
#pragma unroll
    for (int m = 0; m < WT; ++m)
#pragma unroll
        for (int n = 0; n < WT; ++n)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[m][n][q] = 0.0f;
    float rs[WT / 2];
#pragma unroll
    for (int m = 0; m < WT / 2; ++m) rs[m] = 0.0f;

    // transposed-read address of this lane inside a piece: k-block kb = lane >> 4 (steps 8 kb ..), lane 4 q4 + pp of the
    // 16-lane group -> step q4 of the block's first / second four, channels 4 pp .. 4 pp + 3
    const int kb = lane >> 4, q4 = (lane >> 2) & 3, pp = lane & 3;
    const unsigned rd = (unsigned)((32 * (kb >> 1) + 8 * (pp >> 1) + 4 * (kb & 1) + q4) * 16 + 8 * (pp & 1));
    constexpr int hi_off = 256;                    // steps + 4: unit index + 16
    const bool do_rs = pr.rowsum != 0;
    // wave (wm, wn) owns quadrant (A half wm, B half wn) of the tile: a composite pair may not want all four.  An unwanted
    // quadrant is still computed (its wave stages, waits and sums bias rows with the others; skipping its MFMAs would not
    // shorten the workgroup, and a branch around the tile loop made hipcc spill the accumulators) but never stored.
    const bool wanted = (pr.quad_mask >> (2 * wm + wn)) & 1;

    if (nks > 0) {
#pragma unroll
        for (int s = 0; s < D - 1; ++s) issue_stage(s);
    }

    // ---- K loop ---------------------------------------------------------------------------------------------------
    // Per stage: counted wait + barrier, then 64 accumulator tiles x (1 or 3) MFMAs with the transposed fragment reads just in
    // time (two tiles ahead, sched_barrier-pinned) and the DMA pieces of the stage D - 1 ahead between the tiles.  With two
    // stages (f16x3) the whole next stage is issued in the first half of the current one -- it has the second half to land;
    // with five (one-plane modes) the pieces are spread evenly.
    int slot = 0;
    for (int ks = 0; ks < nks; ++ks) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT) : "memory");
        __builtin_amdgcn_s_barrier();
        const int wslot = slot == 0 ? D - 1 : slot - 1;   // the previous stage's slot is free now
        stage_sources();
        char* wst = lds + wslot * STAGE;
        const char* sa = lds + slot * STAGE + rd + (wm * WT) * 1024;
        const char* sbb = lds + slot * STAGE + A_BYTES + rd + (wn * WT) * 1024;
        u32x4 af[WT][P], bf[WT][P];
        auto read_for = [&](auto tc) {
            constexpr int t = decltype(tc)::value;
            if constexpr (t < NPAIR) {
                constexpr int m = t / WT, n = t % WT;
                if constexpr (n == 0) {
#pragma unroll
                    for (int pl = 0; pl < P; ++pl) {
                        const u32x2 lo = ds_read_tr16(sa + pl * T_PLANE + m * 1024), hi = ds_read_tr16(sa + pl * T_PLANE + m * 1024 + hi_off);
                        af[m][pl] = u32x4{lo[0], lo[1], hi[0], hi[1]};
                    }
                }
                if constexpr (m == 0) {
#pragma unroll
                    for (int pl = 0; pl < P; ++pl) {
                        const u32x2 lo = ds_read_tr16(sbb + pl * T_PLANE + n * 1024), hi = ds_read_tr16(sbb + pl * T_PLANE + n * 1024 + hi_off);
                        bf[n][pl] = u32x4{lo[0], lo[1], hi[0], hi[1]};
                    }
                }
            }
        };
        read_for(std::integral_constant<int, 0>{});
        read_for(std::integral_constant<int, 1>{});
        __builtin_amdgcn_sched_barrier(0);
        [&]<int... I>(std::integer_sequence<int, I...>) {
            ([&] {
                constexpr int idx = I, m = idx / WT, n = idx % WT;
                read_for(std::integral_constant<int, idx + 2>{});
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (BF) {
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8, af[m][0]), __builtin_bit_cast(b8, bf[n][0]), acc[m][n], 0, 0, 0);
                } else {
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, af[m][0]), __builtin_bit_cast(h8, bf[n][0]), acc[m][n], 0, 0, 0);
                    if constexpr (P == 2) {
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, af[m][0]), __builtin_bit_cast(h8, bf[n][1]), acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, af[m][1]), __builtin_bit_cast(h8, bf[n][0]), acc[m][n], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                constexpr bool front = (D == 2);   // two stages: piece p after tile 2 p; otherwise spread over the stage
                if constexpr (front ? (idx % 2 == 0 && idx / 2 < PW) : ((idx * PW) % NPAIR == 0)) {
                    if (!(a.xcd_map & 4)) issue_piece(wst, std::integral_constant<int, front ? idx / 2 : idx * PW / NPAIR>{});
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (n == WT - 1) {
                    // row sums of A for four of this wave's eight row tiles (the other wave column takes the other four)
                    if (do_rs && (m >> 2) == wn) {
#pragma unroll
                        for (int pl = 0; pl < P; ++pl)
#pragma unroll
                            for (int w = 0; w < 4; ++w) rs[m & 3] = dot_ones<BF>(af[m][pl][w], rs[m & 3]);
                    }
                }
            }(), ...);
        }(std::make_integer_sequence<int, NPAIR>{});
        issue_advance();
        slot = slot + 1 == D ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- write this split's partial tile -----------------------------------------------------------------------------
    // C/D layout of v_mfma_f32_16x16x32: column = lane & 15, rows 4 (lane >> 4) + q
    float* out = a.slab + (long long)split * a.slab_floats + pr.slab_off;
    if (wanted) {
#pragma unroll
        for (int m = 0; m < WT; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = tm * CH + wm * 16 * WT + 16 * m + 4 * kb + q;
                float* prow = out + (long long)row * pr.Np + tn * CH + wn * 16 * WT + (lane & 15);
#pragma unroll
                for (int n = 0; n < WT; ++n) prow[16 * n] = acc[m][n][q];
            }
    }
    if (do_rs && tn == 0) {
#pragma unroll
        for (int mm = 0; mm < WT / 2; ++mm) {
            float tot = rs[mm] + __shfl_xor(rs[mm], 16);   // the four k-blocks of a row live in lanes l, l+16, l+32, l+48
            tot += __shfl_xor(tot, 32);
            const int row = tm * CH + wm * 16 * WT + 16 * (wn * (WT / 2) + mm) + (lane & 15);
            if (lane < 16) a.rowsum[(long long)split * a.rs_floats + pr.rs_off + row] = tot;
        }
    }
}

static hipError_t launch_hw(int prec, const HWgradArgs& a, hipStream_t st) {
    const dim3 grid((unsigned)(a.ntile_total * a.nsplit)), block(256);
    if (prec == HP_F16X3) hipLaunchKernelGGL((hwgrad_kernel<2, false>), grid, block, 0, st, a);
    else if (prec == HP_F16) hipLaunchKernelGGL((hwgrad_kernel<1, false>), grid, block, 0, st, a);
    else if (prec == HP_BF16) hipLaunchKernelGGL((hwgrad_kernel<1, true>), grid, block, 0, st, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_hwgrad(int prec, const HWgradArgs& a, hipStream_t st) {
    if (a.ntile_total <= 0 || a.nsplit <= 0) return hipSuccess;
    static const int dbg = getenv("WN_HWGRAD_DBG") ? atoi(getenv("WN_HWGRAD_DBG")) : 0;
    if (dbg) {
        HWgradArgs b = a;
        b.xcd_map |= 2 * dbg;
        return launch_hw(prec, b, st);
    }
    return launch_hw(prec, a, st);
}

}  // namespace wn
