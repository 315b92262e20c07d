// Weight-gradient kernel for gfx950: time/batch-contracted outer products
//
//     out_p[m][n] = sum_{b, t} A_p[b][m][t] * B_p[b][n][t + off_p]            (p = "pair")
//
// e.g. dW_tanh[:, :, j] = sum da[t] x[t + off_j]^T.  The contraction index is TIME, which is the
// contiguous axis in memory, while v_mfma_f32_32x32x2_f32 wants the contraction index across
// k-steps with a channel on each lane -- so tiles are read from HBM time-coalesced (8 lanes x
// 16 B per channel row), transposed through LDS (row pitch 36 floats: conflict-free 16-byte
// fragment reads), and each lane then reads its channel's 16 consecutive time steps.
//
// The LDS image is double-buffered (2 x 72 KB at WT=4) and the chunk loop is software-pipelined by hand: the next
// chunk's ds_writes, the HBM loads of the chunk after it and every operand-fragment read sit in the shadow of the
// MFMAs (sched_group_barrier pins the interleave), one barrier per chunk.
//
// Workgroup = 4 waves (2 x 2), each wave owns a (WT*32) x (WT*32) output tile in AGPRs
// (256 accumulator registers at WT=4, i.e. a 256 x 256 tile per workgroup = a whole C x C
// matrix at C=256).  Split-K over time: workgroup `split` handles a contiguous range of
// 32-step chunks and writes its partial tile to its own slab; wgrad_reduce_kernel sums the
// slabs in a fixed order (deterministic; no float atomics) and scatters into PyTorch layouts.
// Bias gradients (row sums of A) are accumulated by the staging threads on the fly.
#include "wn_kernels.h"
#include <type_traits>

namespace wn {

constexpr int kChunk = 32;   // time steps per LDS stage
constexpr int kPitch = 36;   // LDS row pitch in floats (32 + 4: 16-byte aligned, bank-conflict free)

__device__ __forceinline__ int rowof_w(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// A pair's row count need not be a multiple of the workgroup tile: staging rows past the end re-read row 0 of the
// operand (always valid memory) and the outputs they produce -- rows / columns >= M / N of the partial tile, and their
// row sums -- are never read by wgrad_reduce_kernel.  (They used to be replaced by zeros: 4 v_cndmask per staged float4,
// 65 of the ~150 vector instructions of a chunk and 11 spilled registers in a separate "ragged" instantiation.)
template <int WT>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradArgs a) {
    constexpr int ROWS = 2 * WT * 32;   // rows of the A tile and of the B tile held by the workgroup
    constexpr int PASSES = ROWS / 32;   // staging passes: 32 rows x 32 steps per pass (256 threads x float4)
    constexpr int STAGE = 2 * ROWS * kPitch;   // floats of one LDS stage (A tile + B tile)
    // two stages: while the MFMAs consume stage s, the next chunk is written into stage s^1 (one barrier per chunk)
    __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int i = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn_ = wave & 1;

    // ---- which (pair, tile, split) -------------------------------------------------------------
    // XCD-aware: workgroups id and id+8 share an XCD (and its L2).  All tiles of one time-split read the same
    // operand chunks (da, dg, dr, x, z are each used by 2-3 pairs), so they are placed on ONE XCD, back to back:
    // split = (id/8 / ntile) * 8 + id%8, tile = (id/8) % ntile.  nsplit is a multiple of 8 when this mapping is on.
    int split, tile;
    if (a.xcd_map) {
        const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
        split = (local / a.ntile_total) * 8 + xcd;
        tile = local % a.ntile_total;
    } else {
        split = blockIdx.x / a.ntile_total;
        tile = blockIdx.x - split * a.ntile_total;
    }
    int p = 0;
    while (p + 1 < a.npair && tile >= a.pair[p + 1].tile0) ++p;
    const WgradPair pr = a.pair[p];
    const int tl = tile - pr.tile0;
    const int tm = tl / pr.nt, tn = tl - tm * pr.nt;

    const int c_begin = (int)((long long)a.nchunk * split / a.nsplit);
    const int c_end = (int)((long long)a.nchunk * (split + 1) / a.nsplit);

    // ---- staging geometry ---------------------------------------------------------------------
    // thread (trow, tc) stages row q*32+trow, steps 4tc..4tc+3 of every pass q.  Its byte offsets inside a chunk are
    // constant, so a load is  wave-uniform chunk base (SGPRs) + 32-bit per-thread offset : no VALU per load.
    const int trow = tid >> 3, tc = tid & 7;
    const int ld = a.ld;
    unsigned offA[PASSES], offB[PASSES];
#pragma unroll
    for (int q = 0; q < PASSES; ++q) {
        const int ar = tm * ROWS + q * 32 + trow, br = tn * ROWS + q * 32 + trow;
        offA[q] = (unsigned)(((ar < pr.a_cp ? ar : 0) * ld + 4 * tc) * 4);
        offB[q] = (unsigned)(((br < pr.b_cp ? br : 0) * ld + 4 * tc) * 4);
    }
    f32x4 ra[PASSES], rb[PASSES];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 rs[PASSES];   // row sums of A, kept four-wide (two v_pk_fma per staged float4) and folded once at the end
#pragma unroll
    for (int q = 0; q < PASSES; ++q) rs[q] = zero4;

    // The chunk being loaded is addressed as buffer resource (SGPRs: the pair's A / B tensor at one utterance) + scalar byte
    // offset of the chunk's first column + the constant per-thread offset: no vector instruction per load.
    constexpr unsigned kRsrcFlags = 0x00020000u;   // gfx9 raw buffer, 32-bit data format
    auto resource = [&](const float* base) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0xffffffffu, kRsrcFlags);
    };
    __amdgpu_buffer_rsrc_t rsA = resource(pr.A), rsB = resource(pr.Bm);
    unsigned soA = 0, soB = 0;
    // (utterance, chunk-in-row) of the next chunk to fetch, advanced incrementally (no division in the loop)
    int nb = c_begin / a.chunks_per_row, ncc = c_begin - nb * a.chunks_per_row;
    auto next_chunk = [&](bool advance) {   // branch-free: past the end the last chunk is simply fetched again
        const unsigned col = (unsigned)(a.halo + ncc * kChunk);
        rsA = resource(pr.A + (long)nb * pr.a_cp * ld);
        rsB = resource(pr.Bm + (long)nb * pr.b_cp * ld);
        soA = 4u * col;
        soB = 4u * (unsigned)((int)col + pr.off);   // halo >= |off|: never negative
        const int n1 = ncc + 1;
        const bool wrap = n1 == a.chunks_per_row;
        ncc = advance ? (wrap ? 0 : n1) : ncc;
        nb = advance ? (wrap ? nb + 1 : nb) : nb;
    };
    auto load_a = [&](int q) { ra[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, offA[q], soA, 0)); };
    auto load_b = [&](int q) { rb[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, offB[q], soB, 0)); };
    const bool do_rs = pr.rowsum && tn == 0;   // wave-uniform; the branch below holds no memory instruction
    // registers -> LDS stage st ([row][t], pitch 36), + row sums of A (`real` = 0 for the duplicate chunk that the
    // branch-free tail iteration stages, so that it is not counted twice)
    auto write_a = [&](int st, int q, float real) {
        const f32x4 v = ra[q];
        *reinterpret_cast<f32x4*>(&lds[st * STAGE + (q * 32 + trow) * kPitch + 4 * tc]) = v;
        rs[q] += real * v;   // `real` is 0 for the duplicate tail chunk and for workgroups that emit no row sums
    };
    auto write_b = [&](int st, int q) {
        const f32x4 v = rb[q];
        *reinterpret_cast<f32x4*>(&lds[st * STAGE + ROWS * kPitch + (q * 32 + trow) * kPitch + 4 * tc]) = v;
    };

    f32x16 acc[WT][WT];
#pragma unroll
    for (int m = 0; m < WT; ++m)
#pragma unroll
        for (int nn = 0; nn < WT; ++nn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][nn][r] = 0.0f;

    // lane (i, h) contracts time steps 16h .. 16h+15 of a chunk, four (one "sub-step") at a time.
    // fragment piece j of a set: j < WT -> A rows of row-tile j, else B rows of column-tile j-WT
    auto load_frag = [&](int st, int sub, int j, f32x4 (&fa)[WT], f32x4 (&fb)[WT]) {
        const float* As = lds + st * STAGE;
        const float* Bs = As + ROWS * kPitch;
        if (j < WT)
            fa[j] = *reinterpret_cast<const f32x4*>(&As[((wm * WT + j) * 32 + i) * kPitch + 16 * h + 4 * sub]);
        else
            fb[j - WT] = *reinterpret_cast<const f32x4*>(&Bs[((wn_ * WT + (j - WT)) * 32 + i) * kPitch + 16 * h + 4 * sub]);
    };

    // ---- software pipeline -----------------------------------------------------------------------
    // Per 32-step chunk each wave issues 4 regions of 4*WT*WT MFMAs (one per 4-step sub-step).  A region is cut into
    // G = 4*WT groups of WT MFMAs; each group is fenced with sched_barrier and given at most one memory instruction,
    // which therefore issues in the shadow of the previous group's MFMAs (an fp32 MFMA holds the pipe for 64 cycles):
    //   region 0: MFMA(sub 0) || group g: ds_write piece g of chunk c+1 -> stage st^1 ; ds_read frags sub 1
    //   region 1: MFMA(sub 1) || group g: HBM load piece g of chunk c+2 -> registers  ; ds_read frags sub 2
    //   region 2: MFMA(sub 2) || ds_read frags sub 3 ;  BARRIER (stage st consumed by all, stage st^1 complete)
    //   region 3: MFMA(sub 3) || ds_read frags sub 0 of chunk c+1 (stage st^1)
    constexpr int G = 4 * WT;
    f32x4 fa0[WT], fb0[WT], fa1[WT], fb1[WT];
    auto region = [&](const f32x4 (&fa)[WT], const f32x4 (&fb)[WT], auto&& side) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int m = 0; m < WT; ++m) {
                side(s * WT + m);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nn = 0; nn < WT; ++nn)
                    acc[m][nn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[m][s], fb[nn][s], acc[m][nn], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
    };

    if (c_begin < c_end) {
        next_chunk(c_begin + 1 < c_end);
#pragma unroll
        for (int q = 0; q < PASSES; ++q) { load_a(q); load_b(q); }
#pragma unroll
        for (int q = 0; q < PASSES; ++q) { write_a(0, q, do_rs ? 1.0f : 0.0f); write_b(0, q); }
        next_chunk(c_begin + 2 < c_end);
#pragma unroll
        for (int q = 0; q < PASSES; ++q) { load_a(q); load_b(q); }
    }
    __syncthreads();
    if (c_begin < c_end) {
#pragma unroll
        for (int j = 0; j < 2 * WT; ++j) load_frag(0, 0, j, fa0, fb0);
    }
    // The loop body is straight-line code (no branches): with control flow around the loads hipcc falls back to
    // s_waitcnt vmcnt(0) before every load and LDS write, which serialises the HBM loads (measured: 2.5 ms vs 1.7 ms).
    // Past the last chunk the same chunk is fetched again and written to a stage nobody reads.
    // The LDS stage is a compile-time constant of the body (two chunks per loop iteration), so every LDS address is a
    // loop-invariant register plus an immediate: a vector instruction in this stream costs ~10 cycles of MFMA issue.
    auto chunk = [&](auto stage_tag, int c) {
        constexpr int st = decltype(stage_tag)::value;
        const float real = (c + 1 < c_end && do_rs) ? 1.0f : 0.0f;
        // region 0: registers hold chunk c+1 (loaded during chunk c-1's MFMAs) -> stage st^1
        region(fa0, fb0, [&](int g) {
            if (g & 1) write_b(st ^ 1, g >> 1); else write_a(st ^ 1, g >> 1, real);
            if (g >= G - 2 * WT) load_frag(st, 1, g - (G - 2 * WT), fa1, fb1);
        });
        // region 1: HBM loads of chunk c+2 fly under the MFMAs and across the barrier
        next_chunk(c + 3 < c_end);
        region(fa1, fb1, [&](int g) {
            if (g & 1) load_b(g >> 1); else load_a(g >> 1);
            if (g < 2 * WT) load_frag(st, 2, g, fa0, fb0);
        });
        region(fa0, fb0, [&](int g) {
            if (g < 2 * WT) load_frag(st, 3, g, fa1, fb1);
        });
        __syncthreads();   // every wave has read all of stage st and written all of stage st^1
        region(fa1, fb1, [&](int g) {
            if (g < 2 * WT) load_frag(st ^ 1, 0, g, fa0, fb0);
        });
    };
    int c = c_begin;
    for (; c + 2 <= c_end; c += 2) {
        chunk(std::integral_constant<int, 0>{}, c);
        chunk(std::integral_constant<int, 1>{}, c + 1);
    }
    if (c < c_end) chunk(std::integral_constant<int, 0>{}, c);   // c - c_begin is even here

    // ---- write the partial tile to this split's slab --------------------------------------------
    float* out = a.slab + (long long)split * a.slab_floats + pr.slab_off;
#pragma unroll
    for (int m = 0; m < WT; ++m)
#pragma unroll
        for (int nn = 0; nn < WT; ++nn)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = tm * ROWS + (wm * WT + m) * 32 + rowof_w(r, h);
                const int col = tn * ROWS + (wn_ * WT + nn) * 32 + i;
                out[(long long)row * pr.Np + col] = acc[m][nn][r];
            }

    // ---- row sums of A (bias gradients): reduce the 8 threads that share a row ------------------
    if (do_rs) {
        float* rsout = a.rowsum + (long long)split * a.rs_floats + pr.rs_off;
#pragma unroll
        for (int q = 0; q < PASSES; ++q) {
            float v = (rs[q][0] + rs[q][1]) + (rs[q][2] + rs[q][3]);
            v += __shfl_xor(v, 1);
            v += __shfl_xor(v, 2);
            v += __shfl_xor(v, 4);
            if (tc == 0) rsout[tm * ROWS + q * 32 + trow] = v;
        }
    }
}

// Sum the split-K slabs in split order and scatter into the PyTorch-layout gradient tensors.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const ReduceArgs a) {
    const int p = blockIdx.y;
    const ReduceDst d = a.d[p];
    const float dyn = a.dyn_inv ? a.dyn_inv[0] : 1.0f;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const int nq = (d.N + 3) >> 2;   // each thread sums four consecutive columns (16-byte loads; the slab is padded to Np)
    if (idx < (long long)d.M * nq && d.w) {
        const int m = (int)(idx / nq), n = 4 * (int)(idx - (long long)m * nq);
        const float* src = a.slab + d.slab_off + (long long)m * d.Np + n;
        // fixed summation order (split 0, 1, 2, ...: bitwise reproducible), but eight loads in flight at a time
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        int s = 0;
        for (; s + 24 <= a.nsplit; s += 24) {   // the kernel is a chain of dependent round trips: keep 24 loads in flight
            f32x4 t[24];
#pragma unroll
            for (int u = 0; u < 24; ++u) t[u] = *reinterpret_cast<const f32x4*>(src + (long long)(s + u) * a.slab_floats);
#pragma unroll
            for (int u = 0; u < 24; ++u) v += t[u];
        }
        for (; s + 8 <= a.nsplit; s += 8) {
            f32x4 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = *reinterpret_cast<const f32x4*>(src + (long long)(s + u) * a.slab_floats);
#pragma unroll
            for (int u = 0; u < 8; ++u) v += t[u];
        }
        for (; s < a.nsplit; ++s) v += *reinterpret_cast<const f32x4*>(src + (long long)s * a.slab_floats);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (n + j < d.N) d.w[(long long)m * d.sm + (long long)(n + j) * d.sn] = v[j] * (d.post * dyn);
    }
    if (idx < d.M && (d.b0 || d.b1)) {
        const float* src = a.rowsum + d.rs_off + idx;
        float v = 0.0f;
        int s = 0;
        for (; s + 24 <= a.nsplit; s += 24) {   // same fixed order, 24 loads in flight (a serial chain here was the
            float t[24];                        // whole kernel's critical path: 72 dependent round trips)
#pragma unroll
            for (int u = 0; u < 24; ++u) t[u] = src[(long long)(s + u) * a.rs_floats];
#pragma unroll
            for (int u = 0; u < 24; ++u) v += t[u];
        }
        for (; s < a.nsplit; ++s) v += src[(long long)s * a.rs_floats];
        v *= dyn;
        if (d.b0) d.b0[idx] = v;
        if (d.b1) d.b1[idx] = v;
    }
}

hipError_t launch_wgrad(int WT, const WgradArgs& a, hipStream_t st) {
    const unsigned grid = (unsigned)(a.ntile_total * a.nsplit);
    if (grid == 0) return hipSuccess;
#define WN_LAUNCH_WGRAD(W) hipLaunchKernelGGL((wgrad_kernel<W>), dim3(grid), dim3(256), 0, st, a)
    switch (WT) {
        case 1: WN_LAUNCH_WGRAD(1); break;
        case 2: WN_LAUNCH_WGRAD(2); break;
        case 4: WN_LAUNCH_WGRAD(4); break;
        default: return hipErrorInvalidValue;
    }
#undef WN_LAUNCH_WGRAD
    return hipGetLastError();
}

hipError_t launch_wgrad_reduce(const ReduceArgs& a, hipStream_t st) {
    long long mx = 1;
    for (int p = 0; p < a.npair; ++p) {
        const long long mn = (long long)a.d[p].M * ((a.d[p].N + 3) / 4);
        if (mn > mx) mx = mn;
        if (a.d[p].M > mx) mx = a.d[p].M;
    }
    if (a.npair == 0) return hipSuccess;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((mx + 255) / 256), (unsigned)a.npair), dim3(256), 0, st, a);
    return hipGetLastError();
}

}  // namespace wn
