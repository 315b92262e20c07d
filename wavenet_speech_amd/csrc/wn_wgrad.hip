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
// Workgroup = 4 waves (2 x 2), each wave owns a (WT*32) x (WT*32) output tile in AGPRs
// (256 accumulator registers at WT=4, i.e. a 256 x 256 tile per workgroup = a whole C x C
// matrix at C=256).  Split-K over time: workgroup `split` handles a contiguous range of
// 32-step chunks and writes its partial tile to its own slab; wgrad_reduce_kernel sums the
// slabs in a fixed order (deterministic; no float atomics) and scatters into PyTorch layouts.
// Bias gradients (row sums of A) are accumulated by the staging threads on the fly.
#include "wn_kernels.h"

namespace wn {

constexpr int kChunk = 32;   // time steps per LDS stage
constexpr int kPitch = 36;   // LDS row pitch in floats (32 + 4: 16-byte aligned, bank-conflict free)

__device__ __forceinline__ int rowof_w(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

template <int WT>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradArgs a) {
    constexpr int ROWS = 2 * WT * 32;   // rows of the A tile and of the B tile held by the workgroup
    constexpr int PASSES = ROWS / 32;   // staging passes: 32 rows x 32 steps per pass (256 threads x float4)
    __shared__ __attribute__((aligned(16))) float lds[2 * ROWS * kPitch];
    float* As = lds;
    float* Bs = lds + ROWS * kPitch;

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int i = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn_ = wave & 1;

    // ---- which (pair, tile, split) -------------------------------------------------------------
    const int split = blockIdx.x / a.ntile_total;
    const int tile = blockIdx.x - split * a.ntile_total;
    int p = 0;
    while (p + 1 < a.npair && tile >= a.pair[p + 1].tile0) ++p;
    const WgradPair pr = a.pair[p];
    const int tl = tile - pr.tile0;
    const int tm = tl / pr.nt, tn = tl - tm * pr.nt;

    const int c_begin = (int)((long long)a.nchunk * split / a.nsplit);
    const int c_end = (int)((long long)a.nchunk * (split + 1) / a.nsplit);

    // ---- staging geometry ---------------------------------------------------------------------
    const int trow = tid >> 3, tc = tid & 7;
    const int ld = a.ld;
    f32x4 ra[PASSES], rb[PASSES];
    float rs[PASSES];
#pragma unroll
    for (int q = 0; q < PASSES; ++q) rs[q] = 0.0f;

    auto issue = [&](int c) {
        const int b = c / a.chunks_per_row;
        const long col = (long)a.halo + (long)(c - b * a.chunks_per_row) * kChunk + 4 * tc;
#pragma unroll
        for (int q = 0; q < PASSES; ++q) {
            const int ar = tm * ROWS + q * 32 + trow;
            const int br = tn * ROWS + q * 32 + trow;
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            ra[q] = ar < pr.a_cp ? *reinterpret_cast<const f32x4*>(pr.A + ((long)b * pr.a_cp + ar) * ld + col) : zero;
            rb[q] = br < pr.b_cp ? *reinterpret_cast<const f32x4u*>(pr.Bm + ((long)b * pr.b_cp + br) * ld + col + pr.off) : zero;
        }
    };

    f32x16 acc[WT][WT];
#pragma unroll
    for (int m = 0; m < WT; ++m)
#pragma unroll
        for (int nn = 0; nn < WT; ++nn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][nn][r] = 0.0f;

    if (c_begin < c_end) issue(c_begin);
    for (int c = c_begin; c < c_end; ++c) {
        __syncthreads();  // everyone finished reading the previous stage
#pragma unroll
        for (int q = 0; q < PASSES; ++q) {
            *reinterpret_cast<f32x4*>(&As[(q * 32 + trow) * kPitch + 4 * tc]) = ra[q];
            *reinterpret_cast<f32x4*>(&Bs[(q * 32 + trow) * kPitch + 4 * tc]) = rb[q];
            rs[q] += (ra[q][0] + ra[q][1]) + (ra[q][2] + ra[q][3]);
        }
        __syncthreads();
        if (c + 1 < c_end) issue(c + 1);  // next stage's HBM loads fly under this stage's MFMAs

        // lane (i, h) contracts time steps 16h .. 16h+15 of the chunk, four at a time
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
            f32x4 fa[WT], fb[WT];
#pragma unroll
            for (int m = 0; m < WT; ++m)
                fa[m] = *reinterpret_cast<const f32x4*>(&As[((wm * WT + m) * 32 + i) * kPitch + 16 * h + 4 * sub]);
#pragma unroll
            for (int nn = 0; nn < WT; ++nn)
                fb[nn] = *reinterpret_cast<const f32x4*>(&Bs[((wn_ * WT + nn) * 32 + i) * kPitch + 16 * h + 4 * sub]);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int m = 0; m < WT; ++m)
#pragma unroll
                    for (int nn = 0; nn < WT; ++nn)
                        acc[m][nn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[m][s], fb[nn][s], acc[m][nn], 0, 0, 0);
        }
    }

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
    if (pr.rowsum && tn == 0) {
        float* rsout = a.rowsum + (long long)split * a.rs_floats + pr.rs_off;
#pragma unroll
        for (int q = 0; q < PASSES; ++q) {
            float v = rs[q];
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
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long mn = (long long)d.M * d.N;
    if (idx < mn && d.w) {
        const int m = (int)(idx / d.N), n = (int)(idx - (long long)m * d.N);
        const float* src = a.slab + d.slab_off + (long long)m * d.Np + n;
        float v = 0.0f;
        for (int s = 0; s < a.nsplit; ++s) v += src[(long long)s * a.slab_floats];
        d.w[(long long)m * d.sm + (long long)n * d.sn] = v;
    }
    if (idx < d.M && (d.b0 || d.b1)) {
        const float* src = a.rowsum + d.rs_off + idx;
        float v = 0.0f;
        for (int s = 0; s < a.nsplit; ++s) v += src[(long long)s * a.rs_floats];
        if (d.b0) d.b0[idx] = v;
        if (d.b1) d.b1[idx] = v;
    }
}

hipError_t launch_wgrad(int WT, const WgradArgs& a, hipStream_t st) {
    const unsigned grid = (unsigned)(a.ntile_total * a.nsplit);
    if (grid == 0) return hipSuccess;
    switch (WT) {
        case 1: hipLaunchKernelGGL(wgrad_kernel<1>, dim3(grid), dim3(256), 0, st, a); break;
        case 2: hipLaunchKernelGGL(wgrad_kernel<2>, dim3(grid), dim3(256), 0, st, a); break;
        case 4: hipLaunchKernelGGL(wgrad_kernel<4>, dim3(grid), dim3(256), 0, st, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_wgrad_reduce(const ReduceArgs& a, hipStream_t st) {
    long long mx = 1;
    for (int p = 0; p < a.npair; ++p) {
        const long long mn = (long long)a.d[p].M * a.d[p].N;
        if (mn > mx) mx = mn;
        if (a.d[p].M > mx) mx = a.d[p].M;
    }
    if (a.npair == 0) return hipSuccess;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((mx + 255) / 256), (unsigned)a.npair), dim3(256), 0, st, a);
    return hipGetLastError();
}

}  // namespace wn
