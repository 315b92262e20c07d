// Half-precision-MFMA series GEMM for gfx950 (MI355X): the LDS-tiled kernel behind the "f16x3", "f16" and "bf16" modes
// of the residual-block stack.
//
//   out[M x (B*L)] = Wpacked[M x K] * Bop[K x (B*L)]       on v_mfma_f32_32x32x16_{f16,bf16}, fp32 accumulate
//
// f16x3: every operand is a pair of fp16 planes (hi, lo) and every algorithmic product is three MFMAs
// (hi*hi + hi*lo + lo*hi; the lo*lo term is below fp32 rounding), i.e. fp32-equivalent results at 3/16 of the cost of
// v_mfma_f32_32x32x2_f32.  f16 / bf16: one plane, one MFMA.
//
// Why a different kernel from series_gemm_kernel: at 16x the MFMA rate a wave can no longer be fed from L2 directly
// (tools/f16x3_probe.hip: 43 B/clk/CU of fragments, 1.4x instead of ~5x), so operands are shared through LDS:
//  * workgroup = 4 waves (2 x 2) owning (64*MT) rows x 256 time steps; each wave keeps MT x 4 accumulator tiles of
//    32x32 (256 registers at MT=4, one wave per SIMD).
//  * one LDS stage = ONE MFMA k-step (16 channels): the weight tile [plane][k-group 2][row][8] and the activation tile
//    [plane][k-group 2][time 256][8], both as 16-byte units in exactly the order the fragments are read, so
//    ds_read_b128 is conflict-free (a 16-lane group reads 16 consecutive units) and staging is pure LDS-DMA
//    (global_load_lds_dwordx4, 1 KiB per wave-instruction, no VGPRs, no conversion: the half-series layout of
//    wn_half.h IS the fragment layout; weights are pre-packed in LDS-image order).
//  * a ring of D stages (128 KiB), D-1 k-steps of prefetch in flight, ONE raw s_barrier per k-step and a counted
//    s_waitcnt vmcnt(N) -- never 0 inside the loop.
//  * the same XCD-aware blockIdx mapping as the fp32 kernel: all row slabs of one time tile run on one XCD.
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "wn_half.h"
#include "wn_half_dev.h"

namespace wn {


// ---------------------------------------------------------------------------------------------------------------
// dense fp32 [B][C][L]  ->  half series
// ---------------------------------------------------------------------------------------------------------------
template <int P, bool BF>
__global__ __launch_bounds__(256) void hload_kernel(const HLoadArgs a) {
    // one thread = one 16-byte unit (8 channels of one time step); lanes run along time, so the eight dword loads of a
    // wave are 256 contiguous bytes each and the unit stores are 1 KiB contiguous
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long per_b = (long long)a.G * a.L;
    if (idx >= per_b * a.B) return;
    const int b = (int)(idx / per_b);
    const long long rem = idx - (long long)b * per_b;
    const int g = (int)(rem / a.L), t = (int)(rem - (long long)g * a.L);
    const float s = a.scale * (a.dyn_scale ? a.dyn_scale[0] : 1.0f);
    typedef typename HT<BF>::t T;
    typedef typename HT<BF>::v8 V8;
    V8 hi, lo;
    float v[8];
    unsigned ovf = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = 8 * g + j;
        v[j] = c < a.C ? a.src[((long long)b * a.C + c) * a.L + t] * s : 0.0f;
        hi[j] = (T)v[j];
        if (!BF) ovf |= (!(__builtin_fabsf(v[j]) <= 65504.0f)) ? 1u : 0u;
    }
    hi = pin(hi);   // see store4: the remainder must be taken against the stored bits
#pragma unroll
    for (int j = 0; j < 8; ++j) lo[j] = (T)(v[j] - (float)hi[j]);
    const long long pstride = (long long)a.G * a.ld * 16;
    char* d = a.dst + (long long)b * P * pstride + ((long long)g * a.ld + a.halo + t) * 16;
    *reinterpret_cast<V8*>(d) = hi;
    if (P == 2) *reinterpret_cast<V8*>(d + pstride) = lo;
    if (ovf && a.flag) atomicOr(a.flag, 1u);
}

hipError_t launch_hload(const HLoadArgs& a, hipStream_t st) {
    const long long n = (long long)a.B * a.G * a.L;
    if (n <= 0) return hipSuccess;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (a.planes == 2) hipLaunchKernelGGL((hload_kernel<2, false>), grid, block, 0, st, a);
    else if (a.bf16) hipLaunchKernelGGL((hload_kernel<1, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((hload_kernel<1, false>), grid, block, 0, st, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// weight packing:  PyTorch-layout fp32 parameters -> [slab][k-step][plane][k-group 2][row][8] (the LDS image of a stage)
// ---------------------------------------------------------------------------------------------------------------
template <bool TABLE>
__device__ __forceinline__ const float* hpack_src(const float* p, int dyn_id, const HPackDyn& d) {
    const int id = dyn_id & HPACK_DYN_MASK;
    if (!TABLE || id == 0) return p;   // (jobs passed as kernel arguments carry absolute pointers only)
    const char* b = id == 1 ? d.base[0] : (id == 2 ? d.base[1] : d.base[2]);   // (a dynamically indexed array would live in scratch)
    return reinterpret_cast<const float*>(b + reinterpret_cast<uintptr_t>(p));
}

// one thread = one 16-byte unit of plane 0 (+ plane 1); `a` lives in the kernel arguments (hpack_kernel) or in a device table
// (hpack_table_kernel, where wpacked / bias are offsets from d.out)
template <bool TABLE>
__device__ __forceinline__ void hpack_body(const HPackArgs& a, const HPackDyn& d, char* wpacked, float* bias, long long idx) {
    if (idx < a.total_units) {
        const int KG = a.kgroups;
        const long long plane_bytes = (long long)KG * a.rows * 16;     // one k-step of one plane
        const long long kstep_bytes = plane_bytes * a.planes;
        int slab = 0;
        while (slab + 1 < a.nslab && idx * 16 * a.planes >= a.slab_woff[slab + 1]) ++slab;
        const long long rel = idx - a.slab_woff[slab] / (16 * a.planes);   // unit index inside the slab, plane 0 numbering
        const int row = (int)(rel % a.rows);
        const long long kk = rel / a.rows;                             // KG * ks + kg
        const int kg = (int)(kk % KG);
        long long ks = kk / KG;
        const long long ks_abs = ks;
        int s = 0;
        while (ks >= a.seg_nks[s]) { ks -= a.seg_nks[s]; ++s; }
        const PackTile t = a.tile[slab * (a.rows / 32) + row / 32];
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.0f;
        if (t.row0 >= 0 && s < a.slab_nseg[slab]) {
            const HPackSrc src = a.set[t.set].seg[s];
            const int r = t.row0 + (row & 31);
            if (src.ptr != nullptr || (src.flags & HPACK_DYN_MASK)) {
                const float* sp = hpack_src<TABLE>(src.ptr, src.flags, d);
                if (r < src.rows) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        // accumulator order (KG == 2): slot (kg, j) of a 16-channel k-step holds channel 8 (j >> 2) + 4 kg + (j & 3)
                        const int c = (src.flags & HPACK_PERM) ? 16 * (int)ks + 8 * (j >> 2) + 4 * kg + (j & 3) : 8 * KG * (int)ks + 8 * kg + j;
                        if (c < src.cols) v[j] = sp[(long long)r * src.stride_r + (long long)c * src.stride_c] * src.scale;
                    }
                }
            }
        }
        char* dp = wpacked + a.slab_woff[slab] + ks_abs * kstep_bytes + ((long long)kg * a.rows + row) * 16;
        if (a.bf16) {
            b8 hi;
#pragma unroll
            for (int j = 0; j < 8; ++j) hi[j] = (__bf16)v[j];
            *reinterpret_cast<b8*>(dp) = hi;
        } else {
            // weights are packed as 256 * w / (input scale): |w| >= 16 (gate, proj, conv weights of 1/16-scaled inputs) or >= 256 leaves
            // fp16's range and would saturate the gate silently (tanh(inf) = 1) -- say so instead
            unsigned ovf = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) ovf |= (!(__builtin_fabsf(v[j]) <= 65504.0f)) ? 1u : 0u;
            unsigned* fl = TABLE ? d.flag : a.flag;
            if (ovf && fl) atomicOr(fl, 1u);
            h8 hi, lo;
#pragma unroll
            for (int j = 0; j < 8; ++j) hi[j] = (_Float16)v[j];
            hi = pin(hi);
#pragma unroll
            for (int j = 0; j < 8; ++j) lo[j] = (_Float16)(v[j] - (float)hi[j]);
            *reinterpret_cast<h8*>(dp) = hi;
            if (a.planes == 2) *reinterpret_cast<h8*>(dp + plane_bytes) = lo;
        }
    }
    const long long nb = (long long)a.nslab * a.rows;
    if (idx < nb && bias) {
        const int slab = (int)(idx / a.rows), rr = (int)(idx % a.rows);
        const PackTile t = a.tile[slab * (a.rows / 32) + rr / 32];
        float v = 0.0f;
        if (t.row0 >= 0) {
            const HPackSet& ps = a.set[t.set];
            const int r = t.row0 + (rr & 31);
            if (r < ps.bias_rows) {
                if (ps.bias0 != nullptr || ps.bias0_dyn) {
                    const float* b0 = hpack_src<TABLE>(ps.bias0, ps.bias0_dyn, d);
                    const int rep = ps.bias0_rep > 1 ? ps.bias0_rep : 1;
                    for (int i = 0; i < rep; ++i) v += b0[r + (long long)i * ps.bias0_stride];   // fixed order: bitwise reproducible
                }
                if (ps.bias1 != nullptr || ps.bias1_dyn) v += hpack_src<TABLE>(ps.bias1, ps.bias1_dyn, d)[r];
            }
            v *= ps.bias_scale;
        }
        bias[a.slab_boff[slab] + rr] = v;
    }
}

__global__ __launch_bounds__(256) void hpack_kernel(const HPackArgs a) {
    const HPackDyn d = {{nullptr, nullptr, nullptr}, nullptr, nullptr};
    hpack_body<false>(a, d, a.wpacked, a.bias, (long long)blockIdx.x * 256 + threadIdx.x);
}

__global__ __launch_bounds__(256) void hpack_table_kernel(const HPackArgs* jobs, const int* block0, int njobs, const HPackDyn d) {
    // the job of this thread block: block0 is ascending, njobs is a few hundred at most (binary search on wave-uniform values)
    int lo = 0, hi = njobs - 1;
    const int bid = (int)blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (block0[mid] <= bid) lo = mid; else hi = mid - 1;
    }
    const HPackArgs& a = jobs[lo];
    char* wp = d.out + reinterpret_cast<uintptr_t>(a.wpacked);
    float* bp = a.bias ? reinterpret_cast<float*>(d.out + reinterpret_cast<uintptr_t>(a.bias) - 1) : nullptr;   // offset + 1 (0 = no bias)
    hpack_body<true>(a, d, wp, bp, (long long)(bid - block0[lo]) * 256 + threadIdx.x);
}

hipError_t launch_hpack(const HPackArgs& a, hipStream_t st) {
    const long long n = hpack_threads(a);
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(hpack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_hpack_table(const HPackArgs* jobs, const int* block0, int njobs, int nblocks, const HPackDyn& dyn, hipStream_t st) {
    if (njobs <= 0 || nblocks <= 0) return hipSuccess;
    hipLaunchKernelGGL(hpack_table_kernel, dim3((unsigned)nblocks), dim3(256), 0, st, jobs, block0, njobs, dyn);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// the GEMM
// ---------------------------------------------------------------------------------------------------------------
// MT=4: 256-row tiles, one workgroup per CU (all 512 registers of a lane, the whole 160 KiB of LDS: a 5-deep ring at
// f16x3).  MT=2: 128-row tiles sized for TWO workgroups per CU (<= 256 registers, <= 80 KiB): the epilogue of one
// workgroup -- an HBM-bound burst of stores with no MFMA in it -- then overlaps the other's K loop.
template <int MT, int P, bool BF, int EPI>
__global__ __launch_bounds__(256, MT == 4 ? 1 : 2) void hgemm_kernel(const HGemmArgs a) {
    constexpr int ROWS = 64 * MT;
    constexpr int A_PLANE = 2 * ROWS * 16, B_PLANE = 2 * kHCol * 16;
    // a ring stage holds KPS k-steps: one at f16x3 (48 or 24 MFMAs per wave between barriers), two in the one-plane modes
    // (a single k-step is only 16 or 8 MFMAs there: the barrier, the counted wait and the fragment-read latency that each
    // stage pays once would weigh twice as much)
    constexpr int KPS = P == 1 ? 2 : 1;
    constexpr int A_BYTES = P * A_PLANE, B_BYTES = P * B_PLANE, SUB = A_BYTES + B_BYTES, STAGE = KPS * SUB;
    constexpr int LDS_BUDGET = MT == 4 ? 163840 : 81920;
    constexpr int D = (LDS_BUDGET / STAGE) > 8 ? 8 : (LDS_BUDGET / STAGE);   // ring depth: 5 (MT=4), 3 (MT=2)
    constexpr int A_PW = A_BYTES / 4096, B_PW = B_BYTES / 4096;       // 1 KiB pieces per wave per k-step
    constexpr int PW = A_PW + B_PW;
    constexpr int INFLIGHT = (D - 2) * KPS * PW;                      // pieces allowed to be outstanding at the wait
    constexpr int NPAIR = MT * 4;                                     // accumulator tiles per wave
    static_assert(INFLIGHT < 64, "vmcnt is a 6-bit counter");
    __shared__ __attribute__((aligned(1024))) char lds[D * STAGE];
    typedef typename HT<BF>::v8 V8;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;

    // ---- XCD-aware work mapping (as series_gemm_kernel) ------------------------------------------------------
    const int id = blockIdx.x;
    const int xcd = id & 7, local = id >> 3;
    const int slab_i = local % a.nslab;
    const int coltile = (local / a.nslab) * 8 + xcd;
    if (coltile >= a.ncol) return;
    const int b = coltile / a.tiles_per_row;
    const int t0 = (coltile - b * a.tiles_per_row) * kHCol;
    const HSlab sl = a.slab[slab_i];
    const int ld = a.ld;

    int nks = 0;
    for (int s = 0; s < sl.nseg; ++s) nks += a.seg[s].nks;
    if (a.dbg & 1) nks = 1;   // measurement: epilogue (and one k-step) only

    // ---- staging state (all wave-uniform scalars) --------------------------------------------------------------
    const char* a_src = a.wpacked + sl.woff;                 // stage ks of this slab: + ks * A_BYTES
    int is_ks = 0;                                           // next k-step to issue
    int is_seg = 0, is_left = a.seg[0].nks;
    long long is_pstride = a.seg[0].pstride;
    // (measurement, dbg & 8: every tile READS its B operand from the first 1024 columns of utterance 0 -- L2-resident: what the
    // launch would cost if its operands never came from HBM, e.g. the residual product inside a fused forward)
    // in the diagnostic build `make l2operands` (never loaded by the product path; the extra selects cost two instantiations a spill)
#ifdef WN_MEASURE_L2_OPERANDS
    const int b_ld = (a.dbg & 8) ? 0 : b;
    const long long unit0 = (long long)a.halo + ((a.dbg & 8) ? (t0 & 1023) : t0) + 64 * wave;
#else
    const int b_ld = b;
    const long long unit0 = (long long)a.halo + t0 + 64 * wave;   // this wave's first time unit of the tile (+ tap offset)
#endif
    const char* b_src = a.seg[0].base + (long long)b_ld * a.seg[0].ustride + (unit0 + a.seg[0].off) * 16;
    const unsigned lane16 = lane * 16u;

    // PW pieces per wave and stage: A_PW of the weight tile, then B_PW of the activation tile (plane, k-group)
    auto issue_piece = [&](char* stage, auto pic) {      // stage = LDS image of the k-step being staged
        constexpr int PI = decltype(pic)::value;
        if constexpr (PI < A_PW) {
            WN_GLDS(a_src + (long long)is_ks * A_BYTES + wave * 1024 + PI * 4096, lane16, stage + wave * 1024 + PI * 4096);
        } else {
            constexpr int p = (PI - A_PW) / 2, kg = (PI - A_PW) % 2;
            WN_GLDS(b_src + p * is_pstride + (long long)kg * ld * 16, lane16, stage + A_BYTES + ((p * 2 + kg) * kHCol + 64 * wave) * 16);
        }
    };
    auto issue_advance = [&]() {   // past the last k-step the state stops and the surplus issues re-stage the last step (never read)
        if (is_ks + 1 < nks) {
            ++is_ks;
            b_src += 2LL * ld * 16;
            if (--is_left == 0) {
                ++is_seg;
                const HSeg ns = a.seg[is_seg];
                is_left = ns.nks;
                is_pstride = ns.pstride;
                b_src = ns.base + (long long)b_ld * ns.ustride + (unit0 + ns.off) * 16;
            }
        }
    };
    auto issue = [&](int slot) {
#pragma unroll
        for (int kk = 0; kk < KPS; ++kk) {
            char* stage = lds + slot * STAGE + kk * SUB;
            [&]<int... I>(std::integer_sequence<int, I...>) { (issue_piece(stage, std::integral_constant<int, I>{}), ...); }
            (std::make_integer_sequence<int, PW>{});
            issue_advance();
        }
    };

    // ---- accumulators -----------------------------------------------------------------------------------------------
    f32x16 acc[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[m][n][q] = 0.0f;

    // ---- biases of this wave's rows, fetched before the K loop (128-row tiles) -------------------------------------------
    // A bias load inside the epilogue sits between stores: vmcnt counts loads and stores together and in order, so waiting
    // for it drains every store issued before it -- with one small load per row group the epilogue became a chain of
    // store round trips (8 to 32 full drains per tile).  Fetched here the values cost 32 registers and no wait at all.
    constexpr bool PRE = (MT == 2) && (EPI != HEPI_DGATE);
    f32x4 bvec[PRE ? MT : 1][4];
    if constexpr (PRE) {
        const float* bias_p = (a.bias ? a.bias + sl.boff : reinterpret_cast<const float*>(a.wpacked + sl.woff)) + wm * 32 * MT + 4 * h;
        const bool has_bias = a.bias != nullptr;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(bias_p + 32 * m + 8 * i);   // rows 32 m + 8 i + 4 h + (0..3)
                bvec[m][i] = has_bias ? v : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            }
    }

    // ---- prologue: D-1 stages in flight -------------------------------------------------------------------------------
#pragma unroll
    for (int s = 0; s < D - 1; ++s) issue(s);

    const unsigned a_rd = (unsigned)((h * ROWS + wm * 32 * MT + r) * 16);
    const unsigned b_rd = (unsigned)(A_BYTES + (h * kHCol + wn * 128 + r) * 16);   // inside one k-step's image

    int slot = 0;
    for (int ks = 0; ks < nks; ks += KPS) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT) : "memory");   // this wave's pieces of this stage have landed
        __builtin_amdgcn_s_barrier();                                      // ... everyone's have; the previous slot is free
        const int wslot = slot == 0 ? D - 1 : slot - 1;                    // the stage D - 1 ahead goes into the slot just freed
#pragma unroll
        for (int kk = 0; kk < KPS; ++kk) {
            const char* st = lds + slot * STAGE + kk * SUB;
            char* wst = lds + wslot * STAGE + kk * SUB;
            {
                // Fragments are read just in time, two accumulator tiles ahead of their first use (tile (m, n) first needs
                // af[m] when n == 0 and bf[n] when m == 0), and every step is pinned by sched_barrier: left alone hipcc hoists all
                // (MT + 4) * P reads to the top and the k-step's first MFMA waits for most of them.  The DMA pieces of the stage
                // D - 1 ahead are issued one at a time between the tiles (piece p after tile floor(p * NPAIR / PW)), so their
                // issue cost hides behind MFMAs too.  (K is a whole number of stages: every segment has an even k-step count.)
                V8 af[MT][P], bf[4][P];
                auto read_for = [&](auto tc) {
                    constexpr int t = decltype(tc)::value;
                    if constexpr (t < NPAIR) {
                        constexpr int m = t / 4, n = t % 4;
                        if constexpr (n == 0) {
#pragma unroll
                            for (int p = 0; p < P; ++p) af[m][p] = *reinterpret_cast<const V8*>(st + a_rd + p * A_PLANE + m * 512);
                        }
                        if constexpr (m == 0) {
#pragma unroll
                            for (int p = 0; p < P; ++p) bf[n][p] = *reinterpret_cast<const V8*>(st + b_rd + p * B_PLANE + n * 512);
                        }
                    }
                };
                read_for(std::integral_constant<int, 0>{});
                read_for(std::integral_constant<int, 1>{});
                __builtin_amdgcn_sched_barrier(0);
                [&]<int... I>(std::integer_sequence<int, I...>) {
                    ([&] {
                        constexpr int idx = I, m = idx / 4, n = idx % 4;
                        read_for(std::integral_constant<int, idx + 2>{});
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (BF) {
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m][0], bf[n][0], acc[m][n], 0, 0, 0);
                        } else {
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[m][0], bf[n][0], acc[m][n], 0, 0, 0);
                            if constexpr (P == 2) {
                                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[m][0], bf[n][1], acc[m][n], 0, 0, 0);
                                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[m][1], bf[n][0], acc[m][n], 0, 0, 0);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        [&]<int... Q>(std::integer_sequence<int, Q...>) {
                            ([&] {
                                if constexpr ((Q * NPAIR) / PW == idx) {
                                    issue_piece(wst, std::integral_constant<int, Q>{});
                                    __builtin_amdgcn_sched_barrier(0);
                                }
                            }(), ...);
                        }(std::make_integer_sequence<int, PW>{});
                    }(), ...);
                }(std::make_integer_sequence<int, NPAIR>{});
            }
            issue_advance();
        }
        slot = slot + 1 == D ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the surplus stages before the epilogue's own loads/stores
    if (a.dbg & 2) {          // measurement: K loop only (the accumulators stay live through a store that never happens)
        float sum = 0.0f;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int q = 0; q < 16; ++q) sum += acc[m][n][q];
        if (sum == 1.2345e-30f && a.flag) a.flag[0] = 7;
        return;
    }

    // ---- epilogue ---------------------------------------------------------------------------------------------------
    // C/D layout of the 32x32 tile: column = lane & 31, rows (q & 3) + 8 (q >> 2) + 4 h: registers 4i..4i+3 are four
    // CONSECUTIVE channels 8i + 4h .. +3 of the tile = one 8-byte piece of the half-series unit (group, t).
    const float osc = a.oscale;
    unsigned ovf = 0;
    const int rowbase = wm * 32 * MT;                      // first row of this wave inside the slab
    // FAST PATH: a tile that lies wholly inside the tensor (all 256 columns < L, every row group a real channel group) runs
    // straight-line code -- no branch, no load between the stores (the dgate epilogue: loads one row group AHEAD of the
    // stores) -- so hipcc's waits are exact counts and the stores leave back to back.  Every conditional in the general path
    // below is a control-flow join at which hipcc falls back to `s_waitcnt vmcnt(0)` = a full drain of the stores before it.
    const bool full_cols = t0 + kHCol <= a.L;
    if constexpr (PRE && EPI == HEPI_STORE) {
        const HDst d = a.dst[sl.dst];
        if (full_cols && sl.row0 + ROWS <= d.cp) {
            // row groups in pairs (i, i + 1): 16-byte stores, lanes 0-31 the unit of group i, lanes 32-63 that of group i + 1 (store8)
            char* dbase = d.base + (long long)b * d.ustride + ((long long)a.halo + t0 + wn * 128 + r) * 16;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int i = 0; i < 4; i += 2) {
                    char* prow = dbase + (long long)(((sl.row0 + rowbase + 32 * m + 8 * i) >> 3) + h) * ld * 16;
#pragma unroll
                    for (int n = 0; n < 4; ++n) {
                        float va[4], vb[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            va[q] = acc[m][n][4 * i + q] * osc + bvec[m][i][q];
                            vb[q] = acc[m][n][4 * i + 4 + q] * osc + bvec[m][i + 1][q];
                        }
                        store8<P, BF>(prow + n * 512, d.pstride, va, vb, ovf);
                    }
                }
            if constexpr (!BF) {
                if (ovf && a.flag) atomicOr(a.flag, 1u);
            }
            return;
        }
    } else if constexpr (PRE && EPI == HEPI_GATE) {
        if (full_cols && sl.row0 + ROWS / 2 <= a.gate_rows) {
            long long col = ((long long)a.halo + t0 + wn * 128 + r) * 16;
            int bq = b;
            if (a.dbg & 4) { bq = 0; col = ((long long)a.halo + (t0 & 1023) + wn * 128 + r) * 16; }   // measurement: every tile stores into the first 1024 columns (L2-resident)
            char* zb = a.z.base + (long long)bq * a.z.ustride + col;
            const bool keep = a.sg.base != nullptr;            // training: the sigmoid is kept for the backward pass (tanh = z / sigmoid)
            char* sb = keep ? a.sg.base + (long long)bq * a.sg.ustride + col : zb;
#pragma unroll
            for (int j = 0; j < MT / 2; ++j)
#pragma unroll
                for (int i = 0; i < 4; i += 2) {       // row groups in pairs: 16-byte stores (store8)
                    const long long o = (long long)(((sl.row0 + wm * 16 * MT + 32 * j + 8 * i) >> 3) + h) * ld * 16;
#pragma unroll
                    for (int n = 0; n < 4; ++n) {
                        float vsa[4], vza[4], vsb[4], vzb[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float ta = h_tanh(acc[2 * j][n][4 * i + q] * osc + bvec[2 * j][i][q]);
                            vsa[q] = h_sigmoid(acc[2 * j + 1][n][4 * i + q] * osc + bvec[2 * j + 1][i][q]);
                            vza[q] = ta * vsa[q];
                            const float tb = h_tanh(acc[2 * j][n][4 * i + 4 + q] * osc + bvec[2 * j][i + 1][q]);
                            vsb[q] = h_sigmoid(acc[2 * j + 1][n][4 * i + 4 + q] * osc + bvec[2 * j + 1][i + 1][q]);
                            vzb[q] = tb * vsb[q];
                        }
                        const long long on = o + n * 512;
                        store8<P, BF, false>(zb + on, a.z.pstride, vza, vzb, ovf);
                        if (keep) store8<P, BF, false>(sb + on, a.sg.pstride, vsa, vsb, ovf);   // wave-uniform; no load is pending
                    }
                }
            if constexpr (!BF) {
                if (ovf && a.flag) atomicOr(a.flag, 1u);
            }
            return;
        }
    } else if constexpr (MT == 2 && EPI == HEPI_DGATE && P == 1) {
        // one-plane modes: row groups in pairs, 16-byte stores (two planes would need 128 more registers for the pair prefetch: spills)
        if (full_cols && sl.row0 + ROWS <= a.da.cp) {
            typedef typename HT<BF>::v4 V4;
            const long long col = ((long long)a.halo + t0 + wn * 128 + r) * 16;
            const char* tab = a.z.base + (long long)b * a.z.ustride + col + 8 * h;          // z = tanh * sigmoid (the tanh itself is not stored)
            const char* sgb = a.sg.base + (long long)b * a.sg.ustride + col + 8 * h;
            char* dab = a.da.base + (long long)b * a.da.ustride + col;                     // (stores: whole units, see store8)
            char* dgb = a.dg.base + (long long)b * a.dg.ustride + col;
            constexpr int NG = MT * 4;                         // row groups of 8 channels per wave, taken in pairs
            V4 rt[2][2][4][P], rs[2][2][4][P];                 // raw z / sigmoid of a PAIR of row groups, one pair ahead of the stores
            auto fetch = [&](int g, V4 (&ft)[2][4][P], V4 (&fs)[2][4][P]) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const long long o = (long long)((sl.row0 + rowbase + 8 * (g + e)) >> 3) * ld * 16;
#pragma unroll
                    for (int n = 0; n < 4; ++n)
#pragma unroll
                        for (int p = 0; p < P; ++p) {
                            ft[e][n][p] = *reinterpret_cast<const V4*>(tab + o + n * 512 + p * a.z.pstride);
                            fs[e][n][p] = *reinterpret_cast<const V4*>(sgb + o + n * 512 + p * a.sg.pstride);
                        }
                }
            };
            fetch(0, rt[0], rs[0]);
#pragma unroll
            for (int g = 0; g < NG; g += 2) {
                const int cur = (g >> 1) & 1;
                if (g + 2 < NG) fetch(g + 2, rt[cur ^ 1], rs[cur ^ 1]);
                __builtin_amdgcn_sched_barrier(0);
                const int m = g >> 2, i = g & 3;               // groups g, g + 1 = registers 4 i .., 4 i + 4 .. of row tile m
                const long long o = (long long)(((sl.row0 + rowbase + 8 * g) >> 3) + h) * ld * 16;
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    float va[2][4], vg[2][4];
#pragma unroll
                    for (int e = 0; e < 2; ++e)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            float z_ = (float)rt[cur][e][n][0][q], s_ = (float)rs[cur][e][n][0][q];
                            if constexpr (P == 2) { z_ += (float)rt[cur][e][n][1][q]; s_ += (float)rs[cur][e][n][1][q]; }
                            const float dz = acc[m][n][4 * (i + e) + q] * osc;
                            dgate(dz, z_, s_, va[e][q], vg[e][q]);
                        }
                    store8<P, BF>(dab + o + n * 512, a.da.pstride, va[0], va[1], ovf);
                    store8<P, BF>(dgb + o + n * 512, a.dg.pstride, vg[0], vg[1], ovf);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (!BF) {
                if (ovf && a.flag) atomicOr(a.flag, 1u);
            }
            return;
        }
    } else if constexpr (MT == 2 && EPI == HEPI_DGATE) {
        if (full_cols && sl.row0 + ROWS <= a.da.cp) {
            typedef typename HT<BF>::v4 V4;
            const long long col = ((long long)a.halo + t0 + wn * 128 + r) * 16 + 8 * h;
            const char* tab = a.z.base + (long long)b * a.z.ustride + col;          // z = tanh * sigmoid (the tanh itself is not stored)
            const char* sgb = a.sg.base + (long long)b * a.sg.ustride + col;
            char* dab = a.da.base + (long long)b * a.da.ustride + col;
            char* dgb = a.dg.base + (long long)b * a.dg.ustride + col;
            constexpr int NG = MT * 4;                         // row groups of 8 channels per wave
            V4 rt[2][4][P], rs[2][4][P];                       // raw tanh / sigmoid of a row group, one group ahead
            auto fetch = [&](int g, V4 (&ft)[4][P], V4 (&fs)[4][P]) {
                const long long o = (long long)((sl.row0 + rowbase + 8 * g) >> 3) * ld * 16;
#pragma unroll
                for (int n = 0; n < 4; ++n)
#pragma unroll
                    for (int p = 0; p < P; ++p) {
                        ft[n][p] = *reinterpret_cast<const V4*>(tab + o + n * 512 + p * a.z.pstride);
                        fs[n][p] = *reinterpret_cast<const V4*>(sgb + o + n * 512 + p * a.sg.pstride);
                    }
            };
            fetch(0, rt[0], rs[0]);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + 1 < NG) fetch(g + 1, rt[(g + 1) & 1], rs[(g + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                const int m = g >> 2, i = g & 3;
                const long long o = (long long)((sl.row0 + rowbase + 8 * g) >> 3) * ld * 16;
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    float va[4], vg[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float z_ = (float)rt[g & 1][n][0][q], s_ = (float)rs[g & 1][n][0][q];
                        if constexpr (P == 2) { z_ += (float)rt[g & 1][n][1][q]; s_ += (float)rs[g & 1][n][1][q]; }
                        const float dz = acc[m][n][4 * i + q] * osc;
                        dgate(dz, z_, s_, va[q], vg[q]);
                    }
                    store4<P, BF>(dab + o + n * 512, a.da.pstride, va, ovf);
                    store4<P, BF>(dgb + o + n * 512, a.dg.pstride, vg, ovf);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (!BF) {
                if (ovf && a.flag) atomicOr(a.flag, 1u);
            }
            return;
        }
    } else if constexpr (PRE && EPI == HEPI_F32) {
        if (full_cols && !a.out32_accum && sl.row0 + ROWS <= a.out32_rows) {
            const float dsc = osc * (a.dyn_inv ? a.dyn_inv[0] : 1.0f);
            float* pbase = a.out32 + ((long long)b * a.out32_rows + sl.row0 + rowbase + 4 * h) * a.L + t0 + wn * 128 + r;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    float* prow = pbase + (long long)(32 * m + (q & 3) + 8 * (q >> 2)) * a.L;
#pragma unroll
                    for (int n = 0; n < 4; ++n) prow[32 * n] = acc[m][n][q] * dsc + bvec[m][q >> 2][q & 3];
                }
            return;
        }
    }
    // ---- general path: ragged tiles (last columns of an utterance, channel counts that end inside the slab) ----------------
    if constexpr (EPI == HEPI_STORE) {
        const HDst d = a.dst[sl.dst];
        char* dbase = d.base + (long long)b * d.ustride + ((long long)a.halo + t0 + wn * 128 + r) * 16 + 8 * h;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int lrow = rowbase + 32 * m + 8 * i;     // local row of the unit (multiple of 8)
                const int ch = sl.row0 + lrow;
                if (ch < d.cp) {
                    float bv[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) bv[q] = a.bias ? a.bias[sl.boff + lrow + 4 * h + q] : 0.0f;
                    char* prow = dbase + (long long)(ch >> 3) * ld * 16;
#pragma unroll
                    for (int n = 0; n < 4; ++n) {
                        if (t0 + wn * 128 + 32 * n + r < a.L) {
                            float v[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q) v[q] = acc[m][n][4 * i + q] * osc + bv[q];
                            store4<P, BF>(prow + n * 512, d.pstride, v, ovf);
                        }
                    }
                }
            }
    } else if constexpr (EPI == HEPI_LEAKY) {
        const HDst d = a.dst[sl.dst];
        const bool masked = a.z.base != nullptr;
        const float slope = a.leaky, osc2 = a.oscale2;
        const long long colo = ((long long)a.halo + t0 + wn * 128 + r) * 16 + 8 * h;
        char* dbase = d.base + (long long)b * d.ustride + colo;
        const char* mbase = masked ? a.z.base + (long long)b * a.z.ustride + colo : nullptr;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int lrow = rowbase + 32 * m + 8 * i;
                const int ch = sl.row0 + lrow;
                if (ch < d.cp) {
                    float bv[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) bv[q] = a.bias ? a.bias[sl.boff + lrow + 4 * h + q] : 0.0f;
                    const long long ro = (long long)(ch >> 3) * ld * 16;
#pragma unroll
                    for (int n = 0; n < 4; ++n) {
                        if (t0 + wn * 128 + 32 * n + r < a.L) {
                            float v[4];
                            if (masked) {
                                float mv[4];
                                load4<P, BF>(mbase + ro + n * 512, a.z.pstride, mv);
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    const float g = acc[m][n][4 * i + q] * osc;
                                    v[q] = __builtin_signbitf(mv[q]) ? g * slope : g;      // the SIGN BIT is the mask: +0 = a positive that underflowed
                                }
                            } else {
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    const float y = (acc[m][n][4 * i + q] * osc + bv[q]) * osc2;
                                    v[q] = y > 0.0f ? y : -__builtin_fabsf(y * slope);     // y <= 0 is stored with the sign bit set (-0 for 0: torch's x > 0 rule)
                                }
                            }
                            store4<P, BF>(dbase + ro + n * 512, d.pstride, v, ovf);
                        }
                    }
                }
            }
    } else if constexpr (EPI == HEPI_GATE) {
        // tiles (2j, 2j+1) of a wave hold a and g of the same 32 channels
        const long long col = ((long long)a.halo + t0 + wn * 128 + r) * 16 + 8 * h;
#pragma unroll
        for (int j = 0; j < MT / 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int lch = wm * 16 * MT + 32 * j + 8 * i;          // local channel of the unit
                const int ch = sl.row0 + lch;
                if (ch < a.z.cp) {
                    float ba[4], bg[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        ba[q] = a.bias ? a.bias[sl.boff + rowbase + 32 * (2 * j) + 8 * i + 4 * h + q] : 0.0f;
                        bg[q] = a.bias ? a.bias[sl.boff + rowbase + 32 * (2 * j + 1) + 8 * i + 4 * h + q] : 0.0f;
                    }
                    const long long o = (long long)(ch >> 3) * ld * 16 + col;
#pragma unroll
                    for (int n = 0; n < 4; ++n) {
                        if (t0 + wn * 128 + 32 * n + r < a.L) {
                            float vt[4], vs[4], vz[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const bool valid = ch + 4 * h + q < a.gate_rows;   // pad channels stay exactly zero
                                vt[q] = valid ? h_tanh(acc[2 * j][n][4 * i + q] * osc + ba[q]) : 0.0f;
                                vs[q] = valid ? h_sigmoid(acc[2 * j + 1][n][4 * i + q] * osc + bg[q]) : 0.0f;
                                vz[q] = vt[q] * vs[q];
                            }
                            const long long on = o + n * 512;
                            store4<P, BF, false>(a.z.base + (long long)b * a.z.ustride + on, a.z.pstride, vz, ovf);
                            if (a.sg.base) store4<P, BF, false>(a.sg.base + (long long)b * a.sg.ustride + on, a.sg.pstride, vs, ovf);
                        }
                    }
                }
            }
    } else if constexpr (EPI == HEPI_DGATE) {
        const long long col = ((long long)a.halo + t0 + wn * 128 + r) * 16 + 8 * h;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ch = sl.row0 + rowbase + 32 * m + 8 * i;
                if (ch < a.da.cp) {
                    const long long o = (long long)(ch >> 3) * ld * 16 + col;
                    float ta_[4][4], sg_[4][4];
#pragma unroll
                    for (int n = 0; n < 4; ++n) {     // all eight (sixteen with lo planes) loads of the row group in flight together
                        load4<P, BF>(a.z.base + (long long)b * a.z.ustride + o + n * 512, a.z.pstride, ta_[n]);   // z, not tanh
                        load4<P, BF>(a.sg.base + (long long)b * a.sg.ustride + o + n * 512, a.sg.pstride, sg_[n]);
                    }
#pragma unroll
                    for (int n = 0; n < 4; ++n) {
                        if (t0 + wn * 128 + 32 * n + r < a.L) {
                            float va[4], vg[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const float dz = acc[m][n][4 * i + q] * osc;
                                dgate(dz, ta_[n][q], sg_[n][q], va[q], vg[q]);
                            }
                            store4<P, BF>(a.da.base + (long long)b * a.da.ustride + o + n * 512, a.da.pstride, va, ovf);
                            store4<P, BF>(a.dg.base + (long long)b * a.dg.ustride + o + n * 512, a.dg.pstride, vg, ovf);
                        }
                    }
                }
            }
    } else {   // HEPI_F32: dense fp32 [B][rows][L]
        const float dsc = osc * (a.dyn_inv ? a.dyn_inv[0] : 1.0f);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int lrow = rowbase + 32 * m + (q & 3) + 8 * (q >> 2) + 4 * h;
                const int row = sl.row0 + lrow;
                if (row < a.out32_rows) {
                    const float bv = a.bias ? a.bias[sl.boff + lrow] : 0.0f;
                    float* prow = a.out32 + ((long long)b * a.out32_rows + row) * a.L + t0 + wn * 128 + r;
#pragma unroll
                    for (int n = 0; n < 4; ++n) {
                        if (t0 + wn * 128 + 32 * n + r < a.L) {
                            float v = acc[m][n][q] * dsc + bv;
                            if (a.out32_accum) v += prow[32 * n];
                            prow[32 * n] = v;
                        }
                    }
                }
            }
    }
    if constexpr (!BF) {
        if (ovf && a.flag) atomicOr(a.flag, 1u);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The same GEMM on v_mfma_f32_16x16x32: 256 x 128 tiles, EIGHT waves, one workgroup per CU
// ---------------------------------------------------------------------------------------------------------------
// The half path runs at the chip's power limit and the 16x16x32 shape sustains a higher clock than 32x32x16 at equal cycles per
// FLOP (tools/probes/hwgrad_loop.hip: 2.2 vs 1.75 GHz).  Its contraction depth is 32 channels, so a stage holds 32 channels:
// 48 KiB at f16x3 for a 256 x 128 tile -- a three-stage ring takes most of the LDS and there is no room for a second workgroup.
// Instead ONE workgroup of eight waves (4 x 2, each 64 rows x 64 columns = 4 x 4 accumulator tiles of 16 x 16, 64 registers)
// keeps two waves on every SIMD.  Used for FULL tiles only (rows a multiple of 256, L a multiple of 128: the plan decides,
// HPlan::init); everything else stays on hgemm_kernel.  Weights are packed [k-step of 32][plane][k-group 4][row 256][8].
constexpr int kHCol8 = 128;

// NT = accumulator tiles per wave along time.  8 -> 256 x 256 workgroup tiles, 64 x 128 per wave, TWO 64 KiB stages with the
// whole next stage issued at the start of the current one (f16x3: the variant in use); 4 -> 256 x 128 tiles, 64 x 64 per wave,
// three stages (the first version; a third more LDS read traffic and half again the staging per FLOP: gate 0.407 vs 0.350 ms).
// WR = wave rows: 4 -> eight waves on 256 rows, one workgroup per CU (f16x3); 2 -> four waves on 128 rows and half the LDS, two
// workgroups per CU: the form of the one-plane modes, whose 32-channel stage is only 24 KiB at 128 x 256.
template <int P, bool BF, int EPI, int NT, int WR>
__global__ __launch_bounds__(128 * WR, WR == 4 ? 1 : 2) void hgemm8_kernel(const HGemmArgs a) {
    constexpr int ROWS = 64 * WR, COLS = 32 * NT, NWAVE = 2 * WR;
    constexpr int A_PLANE = 4 * ROWS * 16, B_PLANE = 4 * COLS * 16;
    constexpr int A_BYTES = P * A_PLANE, B_BYTES = P * B_PLANE, STAGE = A_BYTES + B_BYTES;   // 48 KiB at f16x3, 24 KiB one plane
    constexpr int LDS_BUDGET = WR == 4 ? 163840 : 81920;
    constexpr int D = (LDS_BUDGET / STAGE) > 6 ? 6 : (LDS_BUDGET / STAGE);
    constexpr int A_PW = A_BYTES / (1024 * NWAVE), B_PW = B_BYTES / (1024 * NWAVE);           // 1 KiB pieces per wave and stage
    static_assert(A_PW * 1024 * NWAVE == A_BYTES && B_PW * 1024 * NWAVE == B_BYTES && D >= 2, "piece split / ring depth");
    constexpr int PW = A_PW + B_PW;
    constexpr int INFLIGHT = (D - 2) * PW;
    constexpr int NPAIR = 4 * NT;
    static_assert(INFLIGHT < 64, "vmcnt is a 6-bit counter");
    static_assert(B_PW >= 1, "piece split");
    __shared__ __attribute__((aligned(1024))) char lds[D * STAGE];
    typedef typename HT<BF>::v8 V8;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int c16 = lane & 15, rq = lane >> 4;

    const int id = blockIdx.x;
    const int xcd = id & 7, local = id >> 3;
    const int slab_i = local % a.nslab;
    const int coltile = (local / a.nslab) * 8 + xcd;
    if (coltile >= a.ncol) return;
    const int b = coltile / a.tiles_per_row;
    const int t0 = (coltile - b * a.tiles_per_row) * COLS;
    const HSlab sl = a.slab[slab_i];
    const int ld = a.ld;

    int nks = 0;
    for (int s = 0; s < sl.nseg; ++s) nks += a.seg[s].nks >> 1;   // segments are whole numbers of 32-channel k-steps
    if (a.dbg & 1) nks = 1;

    const char* a_src = a.wpacked + sl.woff;
    int is_ks = 0;
    int is_seg = 0, is_left = a.seg[0].nks >> 1;
    long long is_pstride = a.seg[0].pstride;
    const long long unit0 = (long long)a.halo + t0;
    const char* b_src = a.seg[0].base + (long long)b * a.seg[0].ustride + (unit0 + a.seg[0].off) * 16;
    const unsigned lane16 = lane * 16u;

    auto issue_piece = [&](char* stage, auto pic) {
        constexpr int PI = decltype(pic)::value;
        if constexpr (PI < A_PW) {
            const int piece = wave + NWAVE * PI;                              // the weight image of a stage is contiguous
            WN_GLDS(a_src + (long long)is_ks * A_BYTES + piece * 1024, lane16, stage + piece * 1024);
        } else {
            constexpr int HC = COLS / 64;                                     // 64-column pieces per (plane, k-group)
            const int q = wave + NWAVE * (PI - A_PW);                         // (plane, k-group, column piece)
            const int hc = q % HC, kg = (q / HC) & 3, p = q / (4 * HC);
            WN_GLDS(b_src + p * is_pstride + (long long)kg * ld * 16 + hc * 1024, lane16,
                    stage + A_BYTES + ((p * 4 + kg) * COLS + 64 * hc) * 16);
        }
    };
    auto issue_advance = [&]() {
        if (is_ks + 1 < nks) {
            ++is_ks;
            b_src += 4LL * ld * 16;
            if (--is_left == 0) {
                ++is_seg;
                const HSeg ns = a.seg[is_seg];
                is_left = ns.nks >> 1;
                is_pstride = ns.pstride;
                b_src = ns.base + (long long)b * ns.ustride + (unit0 + ns.off) * 16;
            }
        }
    };
    auto issue = [&](int slot) {
        char* stage = lds + slot * STAGE;
        [&]<int... I>(std::integer_sequence<int, I...>) { (issue_piece(stage, std::integral_constant<int, I>{}), ...); }
        (std::make_integer_sequence<int, PW>{});
        issue_advance();
    };

    f32x4 acc[4][NT];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[m][n][q] = 0.0f;

    // biases of this lane's rows (16 m + 4 rq + 0..3), fetched before the K loop (see hgemm_kernel)
    f32x4 bvec[4];
    if constexpr (EPI != HEPI_DGATE) {
        const float* bias_p = (a.bias ? a.bias + sl.boff : reinterpret_cast<const float*>(a.wpacked + sl.woff)) + wm * 64 + 4 * rq;
        const bool has_bias = a.bias != nullptr;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(bias_p + 16 * m);
            bvec[m] = has_bias ? v : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
    }

#pragma unroll
    for (int s = 0; s < D - 1; ++s) issue(s);

    // fragment of v_mfma_f32_16x16x32: lane (c16, rq) holds row / column c16, k = 8 rq .. 8 rq + 7 = k-group rq of the stage
    const unsigned a_rd = (unsigned)((rq * ROWS + wm * 64 + c16) * 16);
    const unsigned b_rd = (unsigned)(A_BYTES + (rq * COLS + wn * 16 * NT + c16) * 16);

    int slot = 0;
    for (int ks = 0; ks < nks; ++ks) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT) : "memory");
        __builtin_amdgcn_s_barrier();
        const int wslot = slot == 0 ? D - 1 : slot - 1;
        const char* st = lds + slot * STAGE;
        char* wst = lds + wslot * STAGE;
        V8 af[4][P], bf[NT][P];
        auto read_for = [&](auto tc) {
            constexpr int t = decltype(tc)::value;
            if constexpr (t < NPAIR) {
                constexpr int m = t / NT, n = t % NT;
                if constexpr (n == 0) {
#pragma unroll
                    for (int p = 0; p < P; ++p) af[m][p] = *reinterpret_cast<const V8*>(st + a_rd + p * A_PLANE + m * 256);
                }
                if constexpr (m == 0) {
#pragma unroll
                    for (int p = 0; p < P; ++p) bf[n][p] = *reinterpret_cast<const V8*>(st + b_rd + p * B_PLANE + n * 256);
                }
            }
        };
        read_for(std::integral_constant<int, 0>{});
        read_for(std::integral_constant<int, 1>{});
        __builtin_amdgcn_sched_barrier(0);
        [&]<int... I>(std::integer_sequence<int, I...>) {
            ([&] {
                constexpr int idx = I, m = idx / NT, n = idx % NT;
                read_for(std::integral_constant<int, idx + 2>{});
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (BF) {
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m][0], bf[n][0], acc[m][n], 0, 0, 0);
                } else {
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[m][0], bf[n][0], acc[m][n], 0, 0, 0);
                    if constexpr (P == 2) {
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[m][0], bf[n][1], acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[m][1], bf[n][0], acc[m][n], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                [&]<int... Q>(std::integer_sequence<int, Q...>) {
                    ([&] {
                        // two stages: the whole next stage is issued at the start of this one (piece Q after tile Q); else spread
                        if constexpr (D == 2 ? (Q == idx) : ((Q * NPAIR) / PW == idx)) {
                            issue_piece(wst, std::integral_constant<int, Q>{});
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }(), ...);
                }(std::make_integer_sequence<int, PW>{});
            }(), ...);
        }(std::make_integer_sequence<int, NPAIR>{});
        issue_advance();
        slot = slot + 1 == D ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (a.dbg & 2) {
        float sum = 0.0f;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int q = 0; q < 4; ++q) sum += acc[m][n][q];
        if (sum == 1.2345e-30f && a.flag) a.flag[0] = 7;
        return;
    }

    // ---- epilogue ----------------------------------------------------------------------------------------------------
    // C/D layout of the 16x16 tile: column = lane & 15, rows 4 (lane >> 4) + q: four consecutive channels = one 8-byte piece
    // of the unit (group 2 m + (rq >> 1) of the wave's eight, half rq & 1).  Rows are always whole tiles (the plan sees to it);
    // along time L is a multiple of 16, so a 16-column accumulator tile is either wholly inside the utterance or wholly
    // outside: `nok(n)` is wave-uniform and the only condition in here (it guards stores; loads past L stay inside the row).
    const float osc = a.oscale;
    unsigned ovf = 0;
    const long long col = ((long long)a.halo + t0 + wn * 16 * NT + c16) * 16 + 8 * (rq & 1);
    const int c_first = t0 + wn * 16 * NT;
    auto nok = [&](int n) { return c_first + 16 * n < a.L; };
    if constexpr (EPI == HEPI_STORE) {
        const HDst d = a.dst[sl.dst];
        char* dbase = d.base + (long long)b * d.ustride + col + (long long)(((sl.row0 + wm * 64) >> 3) + (rq >> 1)) * ld * 16;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            char* prow = dbase + (long long)(2 * m) * ld * 16;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = acc[m][n][q] * osc + bvec[m][q];
                if (nok(n)) store4<P, BF>(prow + n * 256, d.pstride, v, ovf);
            }
        }
    } else if constexpr (EPI == HEPI_LEAKY) {
        const HDst d = a.dst[sl.dst];
        const bool masked = a.z.base != nullptr;
        const float slope = a.leaky, osc2 = a.oscale2;
        const long long o0 = col + (long long)(((sl.row0 + wm * 64) >> 3) + (rq >> 1)) * ld * 16;
        char* dbase = d.base + (long long)b * d.ustride + o0;
        const char* mbase = masked ? a.z.base + (long long)b * a.z.ustride + o0 : nullptr;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const long long ro = (long long)(2 * m) * ld * 16;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                if (nok(n)) {
                    float v[4];
                    if (masked) {
                        float mv[4];
                        load4<P, BF>(mbase + ro + n * 256, a.z.pstride, mv);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float g = acc[m][n][q] * osc;
                            v[q] = __builtin_signbitf(mv[q]) ? g * slope : g;      // the SIGN BIT is the mask: +0 = a positive that underflowed
                        }
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float y = (acc[m][n][q] * osc + bvec[m][q]) * osc2;
                            v[q] = y > 0.0f ? y : -__builtin_fabsf(y * slope);     // y <= 0 is stored with the sign bit set (-0 for 0: torch's x > 0 rule)
                        }
                    }
                    store4<P, BF>(dbase + ro + n * 256, d.pstride, v, ovf);
                }
            }
        }
    } else if constexpr (EPI == HEPI_GATE) {
        // tile rows 0..31 of a wave are a, 32..63 are g of the same 32 channels: pairs (m, m + 2)
        const long long o0 = col + (long long)(((sl.row0 + wm * 32) >> 3) + (rq >> 1)) * ld * 16;
        char* zb = a.z.base + (long long)b * a.z.ustride + o0;
        const bool keep = a.sg.base != nullptr;
        char* sb = keep ? a.sg.base + (long long)b * a.sg.ustride + o0 : zb;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                float vs[4], vz[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float vt = h_tanh(acc[m][n][q] * osc + bvec[m][q]);
                    vs[q] = h_sigmoid(acc[m + 2][n][q] * osc + bvec[m + 2][q]);
                    vz[q] = vt * vs[q];
                }
                const long long on = (long long)(2 * m) * ld * 16 + n * 256;
                if (nok(n)) {
                    store4<P, BF, false>(zb + on, a.z.pstride, vz, ovf);
                    if (keep) store4<P, BF, false>(sb + on, a.sg.pstride, vs, ovf);
                }
            }
    } else if constexpr (EPI == HEPI_DGATE) {
        typedef typename HT<BF>::v4 V4;
        const long long o0 = col + (long long)(((sl.row0 + wm * 64) >> 3) + (rq >> 1)) * ld * 16;
        const char* zb = a.z.base + (long long)b * a.z.ustride + o0;          // z = tanh * sigmoid (the tanh itself is not stored)
        const char* sgb = a.sg.base + (long long)b * a.sg.ustride + o0;
        char* dab = a.da.base + (long long)b * a.da.ustride + o0;
        char* dgb = a.dg.base + (long long)b * a.dg.ustride + o0;
        constexpr int NH = NT / 4;                         // groups of four column tiles per 16-row tile
        constexpr int NG = 4 * NH;
        V4 rz[2][4][P], rs[2][4][P];                       // raw z / sigmoid of one group, one group ahead of the stores
        auto fetch = [&](int g, V4 (&fz)[4][P], V4 (&fs)[4][P]) {
            const long long o = (long long)(2 * (g / NH)) * ld * 16 + (g % NH) * 1024;
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    fz[n][p] = *reinterpret_cast<const V4*>(zb + o + n * 256 + p * a.z.pstride);
                    fs[n][p] = *reinterpret_cast<const V4*>(sgb + o + n * 256 + p * a.sg.pstride);
                }
        };
        fetch(0, rz[0], rs[0]);
        [&]<int... G>(std::integer_sequence<int, G...>) {
            ([&] {
                constexpr int g = G, m = g / NH, nh = g % NH;
                if constexpr (g + 1 < NG) fetch(g + 1, rz[(g + 1) & 1], rs[(g + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                const long long o = (long long)(2 * m) * ld * 16 + nh * 1024;
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    float va[4], vg[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float z_ = (float)rz[g & 1][n][0][q], s_ = (float)rs[g & 1][n][0][q];
                        if constexpr (P == 2) { z_ += (float)rz[g & 1][n][1][q]; s_ += (float)rs[g & 1][n][1][q]; }
                        dgate(acc[m][4 * nh + n][q] * osc, z_, s_, va[q], vg[q]);
                    }
                    if (nok(4 * nh + n)) {
                        store4<P, BF>(dab + o + n * 256, a.da.pstride, va, ovf);
                        store4<P, BF>(dgb + o + n * 256, a.dg.pstride, vg, ovf);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }(), ...);
        }(std::make_integer_sequence<int, NG>{});
    } else {   // HEPI_F32: dense fp32 [B][rows][L]
        const float dsc = osc * (a.dyn_inv ? a.dyn_inv[0] : 1.0f);
        float* pbase = a.out32 + ((long long)b * a.out32_rows + sl.row0 + wm * 64 + 4 * rq) * a.L + c_first + c16;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float* prow = pbase + (long long)(16 * m + q) * a.L;
                if (a.out32_accum) {   // the second skips_sum group of a > 32-block stack: a row's reads in flight, then add and store
                    float old[NT];
#pragma unroll
                    for (int n = 0; n < NT; ++n) old[n] = nok(n) ? prow[16 * n] : 0.0f;
#pragma unroll
                    for (int n = 0; n < NT; ++n)
                        if (nok(n)) prow[16 * n] = acc[m][n][q] * dsc + bvec[m][q] + old[n];
                } else {
#pragma unroll
                    for (int n = 0; n < NT; ++n)
                        if (nok(n)) prow[16 * n] = acc[m][n][q] * dsc + bvec[m][q];
                }
            }
    }
    if constexpr (!BF) {
        if (ovf && a.flag) atomicOr(a.flag, 1u);
    }
}

template <int P, bool BF, int NT, int WR = 4>
static hipError_t launch_h8(int epi, const HGemmArgs& a, unsigned grid, hipStream_t st) {
    switch (epi) {
        case HEPI_STORE: hipLaunchKernelGGL((hgemm8_kernel<P, BF, HEPI_STORE, NT, WR>), dim3(grid), dim3(128 * WR), 0, st, a); break;
        case HEPI_GATE: hipLaunchKernelGGL((hgemm8_kernel<P, BF, HEPI_GATE, NT, WR>), dim3(grid), dim3(128 * WR), 0, st, a); break;
        case HEPI_DGATE: hipLaunchKernelGGL((hgemm8_kernel<P, BF, HEPI_DGATE, NT, WR>), dim3(grid), dim3(128 * WR), 0, st, a); break;
        case HEPI_F32: hipLaunchKernelGGL((hgemm8_kernel<P, BF, HEPI_F32, NT, WR>), dim3(grid), dim3(128 * WR), 0, st, a); break;
        case HEPI_LEAKY: hipLaunchKernelGGL((hgemm8_kernel<P, BF, HEPI_LEAKY, NT, WR>), dim3(grid), dim3(128 * WR), 0, st, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

template <int MT, int P, bool BF>
static hipError_t launch_h(int epi, const HGemmArgs& a, unsigned grid, hipStream_t st) {
    static const bool report = getenv("WN_HGEMM_OCCUPANCY") != nullptr;   // measurement: print the occupancy API's answer once
    if (report) {
        static bool done = false;
        if (!done) {
            done = true;
            int nb = -1;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, hgemm_kernel<MT, P, BF, HEPI_F32>, 256, 0);
            fprintf(stderr, "[wn] hgemm_kernel<MT=%d, P=%d, BF=%d>: %d workgroup(s) per CU by the occupancy API\n", MT, P, (int)BF, nb);
        }
    }
    if constexpr (MT == 4) {
        // 256-row tiles are used by the long-K skips_sum product only (HPlan::init): dense fp32 out, or the activated half series
        if (epi == HEPI_F32) hipLaunchKernelGGL((hgemm_kernel<MT, P, BF, HEPI_F32>), dim3(grid), dim3(256), 0, st, a);
        else if (epi == HEPI_LEAKY) hipLaunchKernelGGL((hgemm_kernel<MT, P, BF, HEPI_LEAKY>), dim3(grid), dim3(256), 0, st, a);
        else return hipErrorInvalidValue;
        return hipGetLastError();
    } else {
        switch (epi) {
            case HEPI_STORE: hipLaunchKernelGGL((hgemm_kernel<MT, P, BF, HEPI_STORE>), dim3(grid), dim3(256), 0, st, a); break;
            case HEPI_GATE: hipLaunchKernelGGL((hgemm_kernel<MT, P, BF, HEPI_GATE>), dim3(grid), dim3(256), 0, st, a); break;
            case HEPI_DGATE: hipLaunchKernelGGL((hgemm_kernel<MT, P, BF, HEPI_DGATE>), dim3(grid), dim3(256), 0, st, a); break;
            case HEPI_F32: hipLaunchKernelGGL((hgemm_kernel<MT, P, BF, HEPI_F32>), dim3(grid), dim3(256), 0, st, a); break;
            case HEPI_LEAKY: hipLaunchKernelGGL((hgemm_kernel<MT, P, BF, HEPI_LEAKY>), dim3(grid), dim3(256), 0, st, a); break;
            default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
}

hipError_t launch_hgemm(int prec, int MT, int epi, const HGemmArgs& a_in, hipStream_t st) {
    if (a_in.nslab <= 0 || a_in.B <= 0 || a_in.L <= 0) return hipSuccess;
    HGemmArgs a = a_in;
    static const int dbg = getenv("WN_HGEMM_DBG") ? atoi(getenv("WN_HGEMM_DBG")) : 0;
    static const int dbg_epi = getenv("WN_HGEMM_DBG_EPI") ? atoi(getenv("WN_HGEMM_DBG_EPI")) : -1;   // restrict the knob to one epilogue class
    a.dbg = (dbg_epi < 0 || dbg_epi == epi) ? dbg : 0;
    if (MT == 10) {  // hgemm8_kernel, four waves on 128 x 256 tiles, two workgroups per CU (one-plane modes)
        if (a.L % 16 != 0 || prec == HP_F16X3) return hipErrorInvalidValue;
        a.tiles_per_row = (a.L + 255) / 256;
        a.ncol = a.B * a.tiles_per_row;
        const unsigned g10 = (unsigned)(a.nslab * ((a.ncol + 7) / 8 * 8));
        if (prec == HP_F16) return launch_h8<1, false, 8, 2>(epi, a, g10, st);
        return launch_h8<1, true, 8, 2>(epi, a, g10, st);
    }
    if (MT == 9) {   // hgemm8_kernel, 256 x 256 tiles (f16x3): whole 16-column accumulator tiles, the last tile of a row may be partial
        if (a.L % 16 != 0 || prec != HP_F16X3) return hipErrorInvalidValue;
        a.tiles_per_row = (a.L + 255) / 256;
        a.ncol = a.B * a.tiles_per_row;
        return launch_h8<2, false, 8>(epi, a, (unsigned)(a.nslab * ((a.ncol + 7) / 8 * 8)), st);
    }
    if (MT == 8) {   // hgemm8_kernel, 256 x 128 tiles: full tiles only
        if (a.L % kHCol8 != 0) return hipErrorInvalidValue;
        a.tiles_per_row = a.L / kHCol8;
        a.ncol = a.B * a.tiles_per_row;
        const unsigned grid8 = (unsigned)(a.nslab * ((a.ncol + 7) / 8 * 8));
        if (prec == HP_F16X3) return launch_h8<2, false, 4>(epi, a, grid8, st);
        if (prec == HP_F16) return launch_h8<1, false, 4>(epi, a, grid8, st);
        if (prec == HP_BF16) return launch_h8<1, true, 4>(epi, a, grid8, st);
        return hipErrorInvalidValue;
    }
    a.tiles_per_row = (a.L + kHCol - 1) / kHCol;
    a.ncol = a.B * a.tiles_per_row;
    const unsigned grid = (unsigned)(a.nslab * ((a.ncol + 7) / 8 * 8));
    if (MT == 4) {
        if (prec == HP_F16X3) return launch_h<4, 2, false>(epi, a, grid, st);
        if (prec == HP_F16) return launch_h<4, 1, false>(epi, a, grid, st);
        if (prec == HP_BF16) return launch_h<4, 1, true>(epi, a, grid, st);
    } else if (MT == 2) {
        if (prec == HP_F16X3) return launch_h<2, 2, false>(epi, a, grid, st);
        if (prec == HP_F16) return launch_h<2, 1, false>(epi, a, grid, st);
        if (prec == HP_BF16) return launch_h<2, 1, true>(epi, a, grid, st);
    }
    return hipErrorInvalidValue;
}

}  // namespace wn
