// Series GEMM for gfx950 (MI355X): the one MFMA kernel behind the dilated convs, the 1x1
// residual/skip/projection products and the backward-data products of the WaveNet block.
//
//   out[M x (B*L)] = Wpacked[M x K] * Bop[K x (B*L)]   (+ bias, + fused epilogue)
//
// Design (CDNA4-first, see DESIGN.md "kernels"):
//  * one wavefront per workgroup owns MT*32 output rows x 128 time steps and keeps the whole
//    MT x 4 grid of 32x32 fp32 accumulator tiles in AGPRs (256 registers at MT=4); nothing is
//    staged through LDS and there are no barriers -- v_mfma_f32_32x32x2_f32 runs at the fp32
//    vector rate, so operand traffic per MFMA is tiny (1/2 VGPR per MFMA at MT=4) and is fed
//    straight from L2 with 16-byte loads.
//  * time is the contiguous axis of every activation tensor: lane n of a half-wave reads the
//    four consecutive steps 4n..4n+3 of one channel row with ONE global_load_dwordx4, so a
//    half-wave covers 512 contiguous bytes of x[c][t0+off .. ] -- the dilated tap is just a
//    different start column (`off`), the zero halo of the padded series layout makes it
//    mask-free.  The four floats of that load are the B operands of the four N-tiles.
//  * weights are pre-packed in fragment order (wn_pack.hip): one global_load_dwordx4 per lane
//    gives the A operands of four consecutive k-steps of one 32-row tile, fully coalesced.
//  * register prefetch: weights one k-block (8 channels = 4 MFMA k-steps x MT x 4 tiles) ahead, activations three.
//  * blockIdx -> (column tile, slab) mapping keeps all slabs of one column tile on one XCD
//    (blocks b and b+8 share an XCD/L2), so the activation tile is fetched into one L2 only.
#include <cstdlib>

#include "wn_kernels.h"

namespace wn {

__device__ __forceinline__ int rowof(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// sigmoid / tanh from one v_exp_f32 + one v_rcp_f32 each (both 1 ulp).  Absolute error <= ~2e-7 (measured against
// torch in tests/test_gpu_parity.py::test_gate_activation_accuracy); the ocml tanhf/expf pair cost 24 us of VALU per
// 128x128 tile, i.e. 15 % of the gate GEMM's wave time (tools/block_stamps.py).
__device__ __forceinline__ float sigmoid_f(float x) {
    const float e = __builtin_amdgcn_exp2f(-1.44269504088896341f * x);   // exp(-x); inf for x << 0 -> result 0
    return __builtin_amdgcn_rcpf(1.0f + e);
}
__device__ __forceinline__ float tanh_f(float x) {
    const float ax = __builtin_fabsf(x);
    const float e = __builtin_amdgcn_exp2f(-2.88539008177792681f * ax);  // exp(-2|x|) in (0, 1]
    const float big = (1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e);
    const float x2 = x * x;                                                // |x| < 0.125: odd Taylor series, rel. err < 1e-8
    const float small = ax * (1.0f + x2 * (-0.333333333f + x2 * (0.133333333f + x2 * -0.0539682540f)));
    return __builtin_copysignf(ax < 0.125f ? small : big, x);
}

// B operand of one k-step: NT consecutive time steps of one channel row (one 16- or 8-byte load per lane)
template <int NT> struct BVec;
template <> struct BVec<4> { typedef f32x4 reg_t; };
template <> struct BVec<2> { typedef float __attribute__((ext_vector_type(2))) reg_t; };

// MT x NT accumulator tiles per wave (rows x 32*NT time steps); PFB = how many k-blocks ahead the activation
// (B) operand is prefetched (weights are always one k-block ahead: they are L2 hits); WPS = waves per SIMD the
// register budget is sized for.
template <int MT, int NT, int EPI, int PFB, int WPS>
__global__ __launch_bounds__(64, WPS) void series_gemm_kernel(const GemmArgs a) {
    typedef typename BVec<NT>::reg_t breg_t;
    constexpr int COLS = 32 * NT;
    const int lane = threadIdx.x;
    const int n0 = lane & 31, h0 = lane >> 5;   // prologue / K-loop copies (the epilogue recomputes them)
#ifdef WN_STAMPS
    const unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long st_t1 = 0, st_t2 = 0;
#endif

    // ---- XCD-aware work mapping -------------------------------------------------------------
    // grid = nslab * ncol8 (ncol8 = ncol rounded up to 8).  Workgroups id and id+8 share an XCD:
    // give each XCD every 8th column tile and run that tile's slabs back to back on it.
    const int id = blockIdx.x;
    const int xcd = id & 7, local = id >> 3;
    const int slab_i = local % a.nslab;
    const int coltile = (local / a.nslab) * 8 + xcd;
    if (coltile >= a.ncol) return;
    const int b = coltile / a.tiles_per_row;
    const int t0 = (coltile - b * a.tiles_per_row) * COLS;
    const GemmSlab sl = a.slab[slab_i];
    const int ld = a.ld;

    // ---- accumulators ---------------------------------------------------------------------------
    // Every epilogue starts from the row's bias.  (EPI_ACCUM, the stack's running skips_sum += ..., used to preload the
    // destination tile into the accumulators here; hipcc kept the 256 initial values in VGPRs until the first MFMA and
    // spilled 79-200 registers.  It is a batched read-modify-write epilogue now, see below.)
    f32x16 acc[MT][NT];
    // bias: an UNCONDITIONAL load through a pointer that is always valid, times 0 or 1 -- `a.bias ? a.bias[i] : 0` made hipcc
    // branch around every one of the 64 loads (and, in the EPI_ACCUM preload, spill 79 registers across those branches)
    const float* bias_p = (a.bias ? a.bias : a.wpacked) + sl.boff;
    const float bias_on = a.bias ? 1.0f : 0.0f;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float bv = bias_p[32 * m + rowof(r, h0)] * bias_on;
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[m][t][r] = bv;
        }
    }

    // ---- K loop over the slab's segments, one k-block (8 channel rows) per step ---------------
    int nkb = 0;
    for (int s = 0; s < sl.nseg; ++s) nkb += a.seg[s].nkb;

    // Addressing: operands are fetched with BUFFER loads -- a wave-uniform resource (SGPRs: the tensor of the current
    // segment at this utterance), a wave-uniform running byte offset (one SGPR, advanced by one s_add per k-block)
    // and a per-lane byte offset that never changes (one VGPR) -- so the loop spends no vector instructions on
    // pointers.  (Per-lane 64-bit pointers cost ~7 v_lshl_add_u64 per k-block, ~100 of its 4096 MFMA cycles.)
    constexpr unsigned kRsrcFlags = 0x00020000u;   // gfx9 raw buffer, 32-bit data format
    auto resource = [&](const float* base) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0xffffffffu, kRsrcFlags);
    };
    const __amdgpu_buffer_rsrc_t wrs = resource(a.wpacked + sl.woff);
    const unsigned a_off = lane * 16u;
    unsigned a_soff = 0;
    const long tilebase = (long)a.halo + t0;       // wave-uniform column of the tile
    unsigned b_off[4];   // rows 4h..4h+3 of each 8-row k-block belong to this half-wave
#pragma unroll
    for (int q = 0; q < 4; ++q) b_off[q] = 4u * ((unsigned)(4 * h0 + q) * (unsigned)ld + (unsigned)(NT * n0));
    int seg = 0;
    int seg_left = a.seg[0].nkb;
    __amdgpu_buffer_rsrc_t brs = resource(a.seg[0].base + (long)b * a.seg[0].cp * ld);
    unsigned b_soff = 4u * (unsigned)(tilebase + a.seg[0].off);   // halo >= |off|: never negative
    int b_left = nkb;   // k-blocks not yet fetched
    const unsigned a_last = (unsigned)(nkb > 0 ? nkb - 1 : 0) * (MT * 1024u);

    f32x4 A[2][MT];    // A[slot][m][q]: A operand of k-step q for row-tile m
    breg_t B[4][4];    // B[slot][q][t]: B operand of k-step q for column-tile t
    // The fetch helpers are BRANCH-FREE: with control flow around the loads hipcc loses track of the outstanding
    // loads and waits vmcnt(~0) for the prefetch it has just issued, every k-block (measured: 72 instead of 64
    // cycles per MFMA).  Past the last k-block the offsets simply stop advancing, so the (PFB + 1) surplus
    // fetches re-read the last block.
    auto loadA = [&](f32x4 (&dst)[MT]) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
            dst[m] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, a_off + m * 1024u, a_soff, 0));
        a_soff = min(a_soff + MT * 1024u, a_last);   // stops at the last k-block (s_add + s_min: no vector instruction)
    };
    auto loadB = [&](breg_t (&dst)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if constexpr (NT == 4) dst[q] = __builtin_bit_cast(breg_t, __builtin_amdgcn_raw_buffer_load_b128(brs, b_off[q], b_soff, 0));
            else dst[q] = __builtin_bit_cast(breg_t, __builtin_amdgcn_raw_buffer_load_b64(brs, b_off[q], b_soff, 0));
        }
        const int more = min(b_left - 1, 1);   // b_left >= 1 always: 1 while another k-block follows, else 0
        b_left -= more;
        b_soff += (unsigned)more * (32u * (unsigned)ld);
        seg_left -= more;
        if (b_left > 1 - more && seg_left == 0) {   // (more == 1 && seg_left == 0), written without a bool -> int conversion   // wave-uniform and rare; no vector memory op inside, so the counters stay exact
            ++seg;
            const GemmSeg ns = a.seg[seg];
            seg_left = ns.nkb;
            brs = resource(ns.base + (long)b * ns.cp * ld);
            b_soff = 4u * (unsigned)(tilebase + ns.off);
        }
    };
    auto compute = [&](const f32x4 (&fa)[MT], const breg_t (&fb)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[m][q], fb[q][t], acc[m][t], 0, 0, 0);
    };

    loadA(A[0]);
#pragma unroll
    for (int p = 0; p < PFB; ++p) loadB(B[p]);
#ifdef WN_STAMPS
    __builtin_amdgcn_sched_barrier(0);
    st_t1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
#endif
    int g0 = 0;
    for (; g0 + 4 <= nkb; g0 += 4) {   // straight-line body: 4 k-blocks = 4 x (MT + 4 loads, MT*NT*4 MFMAs)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            loadA(A[(j + 1) & 1]);
            loadB(B[(j + PFB) & 3]);
            __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of the MFMA block
            compute(A[j & 1], B[j & 3]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {       // tail: nkb % 4 blocks (the ring continues at slot 0: g0 is a multiple of 4)
        if (g0 + j < nkb) {
            loadA(A[(j + 1) & 1]);
            loadB(B[(j + PFB) & 3]);
            __builtin_amdgcn_sched_barrier(0);
            compute(A[j & 1], B[j & 3]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

#ifdef WN_STAMPS
    __builtin_amdgcn_sched_barrier(0);
    st_t2 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
    struct StampOut {
        unsigned long long* p; unsigned long long t0, r0, t1, t2; int lane;
        __device__ ~StampOut() {
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long t_issued = __builtin_amdgcn_s_memtime();   // every epilogue store has been issued
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // include the drain of the epilogue's stores
            if (p && lane == 0) {
                p[0] = t0; p[1] = t1; p[2] = t2; p[3] = __builtin_amdgcn_s_memtime();
                p[4] = r0; p[5] = __builtin_amdgcn_s_memrealtime();
                p[6] = __builtin_amdgcn_s_getreg(63492);   // HW_REG_HW_ID
                // HW_REG_XCC_ID in the low half, cycles from the end of the K loop to the last store's issue in the high half
                p[7] = (unsigned long long)__builtin_amdgcn_s_getreg(63508) | ((t_issued - t2) << 32);
            }
        }
    } stamp_out{a.stamps ? a.stamps + 8ull * blockIdx.x : nullptr, st_t0, st_r0, st_t1, st_t2, lane};
#endif
    // ---- epilogue -----------------------------------------------------------------------------
    // the lane index is recomputed here (v_mbcnt: one wave per workgroup, so lane == threadIdx.x) instead of being kept
    // alive across the K loop, which owns the whole register file: hipcc otherwise spills it to scratch and reloads it
    const int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int n = lane_e & 31, h = lane_e >> 5;
    const long colbase = tilebase + NT * n;        // this lane's first column
    const int c0 = t0 + NT * n;  // first of this lane's NT columns
    if (c0 >= a.L) return;
    auto clip = [&](breg_t v) {  // columns >= L stay zero (the layout's zero tail)
#pragma unroll
        for (int t = 1; t < NT; ++t)
            if (c0 + t >= a.L) v[t] = 0.0f;
        return v;
    };

    if constexpr (EPI == EPI_LINEAR) {
        const GemmDst d = a.dst[sl.dst];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = sl.row0 + 32 * m + rowof(r, h);
                if (row < d.rows) {
                    float* p = d.base + ((long)b * d.cp + row) * ld + colbase;
                    breg_t v;
#pragma unroll
                    for (int t = 0; t < NT; ++t) v[t] = acc[m][t][r];
                    *reinterpret_cast<breg_t*>(p) = clip(v);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    } else if constexpr (EPI == EPI_ACCUM) {
        // dst += acc: the old values are fetched a whole batch of rows ahead (8 x 16 B per lane in flight) before any is
        // used -- row-by-row load->add->store chains cost 45 us of a 112 us wave (tools/block_stamps.py)
        const GemmDst d = a.dst[sl.dst];
        constexpr int RB = 8, BPT = 16 / RB, NBATCH = MT * BPT;
        breg_t old[2][RB];
        float* const tile = d.base + (long)b * d.cp * ld + colbase;
        auto fetch = [&](int bi, breg_t (&dst)[RB]) {
            const int m = bi / BPT, r0 = (bi % BPT) * RB;
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                int row = sl.row0 + 32 * m + rowof(r0 + i, h);
                row = row < d.rows ? row : d.rows - 1;   // rows past the end are never stored: clamp, don't branch
                dst[i] = *reinterpret_cast<const breg_t*>(tile + (long)row * ld);
            }
        };
        fetch(0, old[0]);
#pragma unroll
        for (int bi = 0; bi < NBATCH; ++bi) {
            if (bi + 1 < NBATCH) fetch(bi + 1, old[(bi + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            const int m = bi / BPT, r0 = (bi % BPT) * RB;
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const int row = sl.row0 + 32 * m + rowof(r0 + i, h);
                if (row < d.rows) {
                    breg_t v;
#pragma unroll
                    for (int t = 0; t < NT; ++t) v[t] = acc[m][t][r0 + i] + old[bi & 1][i][t];
                    *reinterpret_cast<breg_t*>(tile + (long)row * ld) = clip(v);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    } else if constexpr (EPI == EPI_GATE) {
        // tiles (2i, 2i+1) hold a and g of the same 32 channels
#pragma unroll
        for (int i = 0; i < MT / 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ch = sl.row0 + 32 * i + rowof(r, h);
                if (ch < a.gate_rows) {
                    breg_t vs, vz;
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const float tt = tanh_f(acc[2 * i][t][r]);
                        const float ss = sigmoid_f(acc[2 * i + 1][t][r]);
                        vs[t] = ss;
                        vz[t] = tt * ss;
                    }
                    const long o = ((long)b * a.gate_cp + ch) * ld + colbase;
                    *reinterpret_cast<breg_t*>(a.z + o) = clip(vz);
                    // training keeps sigmoid(g) beside z for the backward pass; tanh(a) is recovered there as z / sigmoid(g)
                    if (a.sg) *reinterpret_cast<breg_t*>(a.sg + o) = clip(vs);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {  // EPI_DGATE: acc = dz ; da = dz*sg*(1-ta^2), dg = dz*ta*sg*(1-sg) with ta = z / sg
        // z/sg come from HBM: issue a whole batch of rows (up to 32 x 16 B per lane in flight) before using any of
        // them -- row-by-row load->use chains cost 55 us of a 186 us wave (tools/block_stamps.py).
        constexpr int RB = (MT == 4) ? 8 : 4;   // rows per batch: 2 x RB x 16 B per lane in flight, one batch ahead
        constexpr int BPT = 16 / RB;            // batches per 32-row tile
        constexpr int NBATCH = MT * BPT;
        breg_t vt[2][RB], vs[2][RB];
        auto fetch = [&](int bi, breg_t (&ft)[RB], breg_t (&fs)[RB]) {
            const int m = bi / BPT, r0 = (bi % BPT) * RB;
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                int ch = sl.row0 + 32 * m + rowof(r0 + i, h);
                ch = ch < a.gate_rows ? ch : a.gate_rows - 1;   // clamp (never stored), no branch
                const long o = ((long)b * a.gate_cp + ch) * ld + colbase;
                ft[i] = *reinterpret_cast<const breg_t*>(a.z + o);      // z = tanh(a) sigmoid(g)
                fs[i] = *reinterpret_cast<const breg_t*>(a.sg + o);
            }
        };
        fetch(0, vt[0], vs[0]);
#pragma unroll
        for (int bi = 0; bi < NBATCH; ++bi) {
            if (bi + 1 < NBATCH) fetch(bi + 1, vt[(bi + 1) & 1], vs[(bi + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            const int m = bi / BPT, r0 = (bi % BPT) * RB;
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const int r = r0 + i;
                const int ch = sl.row0 + 32 * m + rowof(r, h);
                if (ch < a.gate_rows) {
                    const long o = ((long)b * a.gate_cp + ch) * ld + colbase;
                    const breg_t z_ = vt[bi & 1][i], sg_ = vs[bi & 1][i];
                    breg_t va, vg;
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        // tanh = z / sigmoid (not stored): da = dz s (1 - t^2) = dz (s - z t), dg = dz t s (1 - s) = dz z (1 - s);
                        // s = 0 (sigmoid underflow) has z = 0 and both gradients 0
                        const float dz = acc[m][t][r];
                        const float tt = sg_[t] > 0.0f ? z_[t] / sg_[t] : 0.0f;
                        va[t] = dz * (sg_[t] - z_[t] * tt);
                        vg[t] = dz * z_[t] * (1.0f - sg_[t]);
                    }
                    *reinterpret_cast<breg_t*>(a.da + o) = clip(va);
                    *reinterpret_cast<breg_t*>(a.dg + o) = clip(vg);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

#ifdef WN_STAMPS
static unsigned long long* g_stamp_buf = nullptr;
static int g_stamp_kc = -1;          // only launches of this kernel class write stamps (-1: all)
int g_debug_kc = -1;                 // set by the API layer before each launch
extern "C" void wn_debug_set_stamp_buffer(void* p) { g_stamp_buf = (unsigned long long*)p; }
extern "C" void wn_debug_set_stamp_class(int kc) { g_stamp_kc = kc; }
extern "C" unsigned wn_debug_last_grid = 0;
#endif

template <int MT, int NT, int EPI, int PFB, int WPS>
static hipError_t launch_one(GemmArgs a, hipStream_t st) {
#ifdef WN_STAMPS
    const bool stamp_this = g_stamp_buf && (g_stamp_kc < 0 || g_stamp_kc == g_debug_kc);
    a.stamps = stamp_this ? g_stamp_buf : nullptr;
#endif
    a.tiles_per_row = (a.L + 32 * NT - 1) / (32 * NT);
    a.ncol = a.B * a.tiles_per_row;
    const int ncol8 = (a.ncol + 7) / 8 * 8;
    const unsigned grid = (unsigned)(a.nslab * ncol8);
#ifdef WN_STAMPS
    if (stamp_this) wn_debug_last_grid = grid;
#endif
    hipLaunchKernelGGL((series_gemm_kernel<MT, NT, EPI, PFB, WPS>), dim3(grid), dim3(64), 0, st, a);
    return hipGetLastError();
}

// Variant 0 (default): MT x 4 tiles, B prefetched 3 k-blocks ahead, one wave per SIMD.
// WN_GEMM_VARIANT=1 selects prefetch depth 1 for A/B measurements (tools/gemm_sweep.py).
static int gemm_variant() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("WN_GEMM_VARIANT");
        v = e ? atoi(e) : 0;
    }
    return v;
}

template <int EPI>
static hipError_t launch_epi(int MT, const GemmArgs& a, hipStream_t st) {
    const int v = gemm_variant();
    if (MT == 4) {
        // measured on MI355X (tools/gemm_sweep.py, profiles/r01/gemm_variants.txt): prefetch depth 1 vs 3 is a
        // tie; 64-column tiles at two waves per SIMD (NT=2) were 20-30 % slower and are not instantiated.
        if (v == 1) return launch_one<4, 4, EPI, 1, 1>(a, st);
        return launch_one<4, 4, EPI, 3, 1>(a, st);
    }
    if (MT == 2) return launch_one<2, 4, EPI, 3, 2>(a, st);
    if constexpr (EPI != EPI_GATE) {
        if (MT == 1) return launch_one<1, 4, EPI, 3, 2>(a, st);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_gemm(int MT, int epi, const GemmArgs& a, hipStream_t st) {
    if (a.nslab <= 0 || a.B <= 0 || a.L <= 0) return hipSuccess;
    switch (epi) {
        case EPI_LINEAR: return launch_epi<EPI_LINEAR>(MT, a, st);
        case EPI_GATE: return launch_epi<EPI_GATE>(MT, a, st);
        case EPI_DGATE: return launch_epi<EPI_DGATE>(MT, a, st);
        case EPI_ACCUM: return launch_epi<EPI_ACCUM>(MT, a, st);
    }
    return hipErrorInvalidValue;
}

}  // namespace wn
