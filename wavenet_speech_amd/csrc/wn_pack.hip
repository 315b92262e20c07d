// Weight repacking: PyTorch-layout parameters -> MFMA-fragment order for series_gemm_kernel.
//
// Packed layout of one slab:  [k-block g][row-tile m (MT)][lane (64)][q (4)]  floats, where
//   lane = 32*h + i  holds  W[row = 32*m + i][k = 8*g + 4*h + q]
// i.e. exactly the A operand of v_mfma_f32_32x32x2_f32 (lane l: A[i = l&31][k = l>>5]) for the
// four k-steps q of k-block g, so one coalesced global_load_dwordx4 per lane feeds four MFMAs.
// Runs once per optimizer step per block (a few MB), so one thread per output float is plenty.
#include "wn_kernels.h"

namespace wn {

__global__ __launch_bounds__(256) void pack_kernel(const PackArgs a) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx < a.total) {
        // locate the slab
        int slab = 0;
        while (slab + 1 < a.nslab && idx >= a.slab_woff[slab + 1]) ++slab;
        const long long rel = idx - a.slab_woff[slab];
        const int q = (int)(rel & 3);
        const int lane = (int)((rel >> 2) & 63);
        const long long gm = rel >> 8;  // g*MT + m
        const int m = (int)(gm % a.MT);
        int g = (int)(gm / a.MT);
        int s = 0;
        while (g >= a.seg_nkb[s]) { g -= a.seg_nkb[s]; ++s; }
        const int c = 8 * g + 4 * (lane >> 5) + q;
        const PackTile t = a.tile[slab * a.MT + m];
        float v = 0.0f;
        if (t.row0 >= 0) {
            const PackSrc src = a.set[t.set].seg[s];
            const int r = t.row0 + (lane & 31);
            if (src.ptr && r < src.rows && c < src.cols) v = src.ptr[(long long)r * src.stride_r + (long long)c * src.stride_c];
        }
        a.wpacked[idx] = v;
    }
    // bias: [slab][MT*32]
    const long long nb = (long long)a.nslab * a.MT * 32;
    if (idx < nb && a.bias) {
        const int slab = (int)(idx / (a.MT * 32));
        const int rr = (int)(idx % (a.MT * 32));
        const PackTile t = a.tile[slab * a.MT + rr / 32];
        float v = 0.0f;
        if (t.row0 >= 0) {
            const PackSet& ps = a.set[t.set];
            const int r = t.row0 + (rr & 31);
            if (r < ps.bias_rows) {
                if (ps.bias0) v += ps.bias0[r];
                if (ps.bias1) v += ps.bias1[r];
            }
        }
        a.bias[a.slab_boff[slab] + rr] = v;
    }
}

hipError_t launch_pack(const PackArgs& a, hipStream_t st) {
    long long n = a.total;
    const long long nb = (long long)a.nslab * a.MT * 32;
    if (nb > n) n = nb;
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a);
    return hipGetLastError();
}

}  // namespace wn
