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
//  * register double-buffering one k-block (8 channels = 4 MFMA k-steps x MT x 4 tiles) ahead.
//  * blockIdx -> (column tile, slab) mapping keeps all slabs of one column tile on one XCD
//    (blocks b and b+8 share an XCD/L2), so the activation tile is fetched into one L2 only.
#include "wn_kernels.h"

namespace wn {

__device__ __forceinline__ int rowof(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

template <int MT>
struct Frag {
    f32x4 a[MT];  // a[m][q]: A operand of k-step q for row-tile m
    f32x4 b[4];   // b[q][t]: B operand of k-step q for column-tile t
};

template <int MT, int EPI>
__global__ __launch_bounds__(64) void series_gemm_kernel(const GemmArgs a) {
    const int lane = threadIdx.x;
    const int n = lane & 31, h = lane >> 5;

    // ---- XCD-aware work mapping -------------------------------------------------------------
    // grid = nslab * ncol8 (ncol8 = ncol rounded up to 8).  Workgroups id and id+8 share an XCD:
    // give each XCD every 8th column tile and run that tile's slabs back to back on it.
    const int id = blockIdx.x;
    const int xcd = id & 7, local = id >> 3;
    const int slab_i = local % a.nslab;
    const int coltile = (local / a.nslab) * 8 + xcd;
    if (coltile >= a.ncol) return;
    const int b = coltile / a.tiles_per_row;
    const int t0 = (coltile - b * a.tiles_per_row) * kColTile;
    const GemmSlab sl = a.slab[slab_i];
    const int ld = a.ld;

    // ---- accumulators, initialised with the bias of their row --------------------------------
    f32x16 acc[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float bv = a.bias ? a.bias[sl.boff + 32 * m + rowof(r, h)] : 0.0f;
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[m][t][r] = bv;
        }
    }

    // ---- K loop over the slab's segments, one k-block (8 channel rows) per step ---------------
    int nkb_total = 0;
    for (int s = 0; s < sl.nseg; ++s) nkb_total += a.seg[s].nkb;

    const float* wp = a.wpacked + sl.woff + lane * 4;
    const long colbase = (long)a.halo + t0 + 4 * n;
    int seg = 0;
    int seg_left = a.seg[0].nkb;
    const float* bp = a.seg[0].base + ((long)b * a.seg[0].cp + 4 * h) * ld + colbase + a.seg[0].off;

    auto load = [&](Frag<MT>& f) {
#pragma unroll
        for (int m = 0; m < MT; ++m) f.a[m] = *reinterpret_cast<const f32x4*>(wp + m * 256);
#pragma unroll
        for (int q = 0; q < 4; ++q) f.b[q] = *reinterpret_cast<const f32x4u*>(bp + (long)q * ld);
        wp += MT * 256;
        bp += 8 * (long)ld;
        if (--seg_left == 0) {
            ++seg;
            if (seg < sl.nseg) {
                seg_left = a.seg[seg].nkb;
                bp = a.seg[seg].base + ((long)b * a.seg[seg].cp + 4 * h) * ld + colbase + a.seg[seg].off;
            }
        }
    };
    auto compute = [&](const Frag<MT>& f) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[m][q], f.b[q][t], acc[m][t], 0, 0, 0);
    };

    Frag<MT> f0, f1;
    load(f0);
    for (int g = 0; g < nkb_total; g += 2) {
        const bool has1 = g + 1 < nkb_total;
        if (has1) load(f1);
        __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of the MFMA block
        compute(f0);
        __builtin_amdgcn_sched_barrier(0);
        if (has1) {
            if (g + 2 < nkb_total) load(f0);
            __builtin_amdgcn_sched_barrier(0);
            compute(f1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- epilogue -----------------------------------------------------------------------------
    const int c0 = t0 + 4 * n;  // first of this lane's four columns
    if (c0 >= a.L) return;
    const bool v1 = c0 + 1 < a.L, v2 = c0 + 2 < a.L, v3 = c0 + 3 < a.L;
    auto clip = [&](f32x4 v) {  // columns >= L stay zero (the layout's zero tail)
        if (!v1) v[1] = 0.0f;
        if (!v2) v[2] = 0.0f;
        if (!v3) v[3] = 0.0f;
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
                    f32x4 v = {acc[m][0][r], acc[m][1][r], acc[m][2][r], acc[m][3][r]};
                    if (d.accumulate) v += *reinterpret_cast<const f32x4*>(p);
                    *reinterpret_cast<f32x4*>(p) = clip(v);
                }
            }
        }
    } else if constexpr (EPI == EPI_GATE) {
        // tiles (2i, 2i+1) hold a and g of the same 32 channels
#pragma unroll
        for (int i = 0; i < MT / 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ch = sl.row0 + 32 * i + rowof(r, h);
                if (ch < a.gate_rows) {
                    f32x4 vt, vs, vz;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float tt = tanhf(acc[2 * i][t][r]);
                        const float ss = sigmoid_f(acc[2 * i + 1][t][r]);
                        vt[t] = tt;
                        vs[t] = ss;
                        vz[t] = tt * ss;
                    }
                    const long o = ((long)b * a.gate_cp + ch) * ld + colbase;
                    *reinterpret_cast<f32x4*>(a.z + o) = clip(vz);
                    if (a.ta) {
                        *reinterpret_cast<f32x4*>(a.ta + o) = clip(vt);
                        *reinterpret_cast<f32x4*>(a.sg + o) = clip(vs);
                    }
                }
            }
        }
    } else {  // EPI_DGATE: acc = dz ; da = dz*sg*(1-ta^2), dg = dz*ta*sg*(1-sg)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ch = sl.row0 + 32 * m + rowof(r, h);
                if (ch < a.gate_rows) {
                    const long o = ((long)b * a.gate_cp + ch) * ld + colbase;
                    const f32x4 vt = *reinterpret_cast<const f32x4*>(a.ta + o);
                    const f32x4 vs = *reinterpret_cast<const f32x4*>(a.sg + o);
                    f32x4 va, vg;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float dz = acc[m][t][r];
                        va[t] = dz * vs[t] * (1.0f - vt[t] * vt[t]);
                        vg[t] = dz * vt[t] * vs[t] * (1.0f - vs[t]);
                    }
                    *reinterpret_cast<f32x4*>(a.da + o) = clip(va);
                    *reinterpret_cast<f32x4*>(a.dg + o) = clip(vg);
                }
            }
        }
    }
}

template <int MT, int EPI>
static hipError_t launch_one(const GemmArgs& a, hipStream_t st) {
    const int ncol8 = (a.ncol + 7) / 8 * 8;
    const unsigned grid = (unsigned)(a.nslab * ncol8);
    hipLaunchKernelGGL((series_gemm_kernel<MT, EPI>), dim3(grid), dim3(64), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_gemm(int MT, int epi, const GemmArgs& a, hipStream_t st) {
    if (a.nslab <= 0 || a.ncol <= 0) return hipSuccess;
    switch (epi * 8 + MT) {
        case EPI_LINEAR * 8 + 1: return launch_one<1, EPI_LINEAR>(a, st);
        case EPI_LINEAR * 8 + 2: return launch_one<2, EPI_LINEAR>(a, st);
        case EPI_LINEAR * 8 + 4: return launch_one<4, EPI_LINEAR>(a, st);
        case EPI_GATE * 8 + 2: return launch_one<2, EPI_GATE>(a, st);
        case EPI_GATE * 8 + 4: return launch_one<4, EPI_GATE>(a, st);
        case EPI_DGATE * 8 + 1: return launch_one<1, EPI_DGATE>(a, st);
        case EPI_DGATE * 8 + 2: return launch_one<2, EPI_DGATE>(a, st);
        case EPI_DGATE * 8 + 4: return launch_one<4, EPI_DGATE>(a, st);
    }
    return hipErrorInvalidValue;
}

}  // namespace wn
