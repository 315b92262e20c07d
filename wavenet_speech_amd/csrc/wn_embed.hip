// Entry conv of WaveNet on quantised levels.  The reference feeds entry_conv1d a dense one-hot
// [B][classes][L] built by modules/fns.py:6-15 (262 MB of zeros and ones at 16 x 256 x 16000) and multiplies it
// (modules/wavenet.py:54,93); with the level indices q[b][t] as input the same causal conv is a gather of weight columns:
//
//     y[b][co][t] = bias[co] + sum_j W[co][ q[b][t + j - (k-1)] ][j]          (taps before t = 0 contribute nothing)
//
// and its weight gradient a per-class segmented sum   dW[co][c][j] = sum_{b,t : q[b][t + j - (k-1)] = c} dy[b][co][t].
// Both are HBM-bound byte work (dy / y are read / written once; the weight table is 2 KB per output channel and stays
// in L1/LDS), so no MFMA here: time-coalesced 16-byte accesses and a deterministic reduction -- no float atomics.
#include "../../include/wavenet_amd.h"
#include "wn_kernels.h"

namespace wn {

constexpr int kEmbCo = 16;     // output channels per thread in the forward kernel (the level indices are read once per 16)
constexpr int kEmbTile = 64;   // output channels per workgroup in the backward kernel
constexpr int kEmbParts = 256 / kEmbTile;   // class ranges: thread = (channel, class range)
constexpr int kEmbSlabs = 64;  // position slabs of the backward kernel: partial tables are reduced in slab order

// ---- forward: one thread = 4 consecutive time steps x kEmbCo channels ----------------------------------------------
__global__ __launch_bounds__(256) void embed_forward_kernel(const long long* __restrict__ q, const float* __restrict__ W,
                                                            const float* __restrict__ bias, float* __restrict__ y, int B, int L,
                                                            int classes, int Co, int k, int* __restrict__ bad) {
    const int L4 = (L + 3) / 4;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const int cg = blockIdx.y;                          // channel group
    if (gid >= (long long)B * L4) return;
    const int b = (int)(gid / L4), t0 = (int)(gid - (long long)b * L4) * 4;
    // levels t0-(k-1) .. t0+3
    int lev[4 + WN_MAX_TAPS];
    bool any_bad = false;
#pragma unroll
    for (int i = 0; i < 4 + WN_MAX_TAPS - 1; ++i) {
        const int t = t0 - (k - 1) + i;
        long long v = -1;
        if (i < 3 + k && t >= 0 && t < L) {
            v = q[(long long)b * L + t];
            if (v < 0 || v >= classes) { any_bad = true; v = -1; }   // never index the table with an unchecked level
        }
        lev[i] = (int)v;
    }
    if (any_bad && bad) atomicAdd(bad, 1);
    const bool vec = (L % 4 == 0);
    for (int c = 0; c < kEmbCo; ++c) {
        const int co = cg * kEmbCo + c;
        if (co >= Co) break;
        const float* w = W + (long long)co * classes * k;
        const float bv = bias ? bias[co] : 0.0f;
        float out[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float acc = bv;
#pragma unroll
            for (int j = 0; j < WN_MAX_TAPS; ++j) {      // unrolled to the maximum so that lev[] stays in registers
                if (j < k) {
                    const int l = lev[i + j];
                    if (l >= 0) acc += w[l * k + j];
                }
            }
            out[i] = acc;
        }
        float* dst = y + ((long long)b * Co + co) * L + t0;
        if (vec) {
            *reinterpret_cast<f32x4*>(dst) = f32x4{out[0], out[1], out[2], out[3]};
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (t0 + i < L) dst[i] = out[i];
        }
    }
}

// ---- backward: workgroup = (position slab, 64-channel tile); LDS holds the [tap][class][64] partial table -----------
// thread = (channel = tid & 63, class quarter = tid >> 6): every thread walks ALL positions of the slab in order but adds
// only the positions whose class falls in its quarter, so each table cell has exactly one writer and a fixed summation
// order (bitwise reproducible).  dy is staged through LDS in 64-position chunks so the global reads stay time-coalesced.
// Cost: the walk is instruction-bound -- positions x threads iterations, three quarters of them idle -- 1.95 ms at
// 16 x 256 x 16000 (tools/embed_bench.py); 16-channel tiles with four workgroups per CU were slower still (2.5 ms: 16x the
// redundancy).  The host side therefore uses this kernel only when asked to (WN_EMBED_BACKWARD=gather) and otherwise forms
// dW with the fp32 wgrad GEMM on a one-hot that lives only for the duration of the backward call (0.55 ms).
__global__ __launch_bounds__(256) void embed_backward_kernel(const long long* __restrict__ q, const float* __restrict__ dy,
                                                             float* __restrict__ partial, int B, int L, int classes, int Co,
                                                             int k, int npos_per_slab) {
    extern __shared__ float lds[];
    const int slab = blockIdx.x, ct = blockIdx.y;
    const int tid = threadIdx.x, col = tid & (kEmbTile - 1), quarter = tid / kEmbTile;
    const int cq = (classes + kEmbParts - 1) / kEmbParts;    // classes per range
    float* table = lds;                                      // [k][classes + 1][T]   (row `classes` = bias gradient, tap 0 only)
    float* stage = lds + (long long)k * (classes + 1) * kEmbTile;  // [T channels][65]
    int* slev = reinterpret_cast<int*>(stage + kEmbTile * 65);     // [64 + k - 1] levels of the chunk
    for (int i = tid; i < k * (classes + 1) * kEmbTile; i += 256) table[i] = 0.0f;
    // positions are flat indices b*L + t < 2^31 (checked by the host); all index arithmetic below is 32-bit and the
    // (utterance, step) of a position is tracked incrementally -- the first version spent its time in 64-bit div/mod
    const int npos = B * L;
    const int p0 = slab * npos_per_slab;
    const int p1 = min(p0 + npos_per_slab, npos);
    for (int c0 = p0; c0 < p1; c0 += 64) {                   // chunks of 64 consecutive positions (may cross utterances)
        __syncthreads();
        const int b0 = c0 / L, t0 = c0 - b0 * L;             // wave-uniform, once per chunk
        // stage dy[ct*64 + r][pos] for the chunk: lanes run along positions (time-coalesced)
        for (int e = tid; e < kEmbTile * 64; e += 256) {
            const int r = e >> 6, pp = e & 63;
            float v = 0.0f;
            if (c0 + pp < p1) {
                int b = b0, t = t0 + pp;
                while (t >= L) { t -= L; ++b; }              // at most a few iterations (only when L < 64)
                const int co = ct * kEmbTile + r;
                if (co < Co) v = dy[((long long)b * Co + co) * L + t];
            }
            stage[r * 65 + pp] = v;
        }
        for (int e = tid; e < 64 + k - 1; e += 256) {        // level feeding tap j of position pp: index pp + j
            const int pos = c0 + e - (k - 1);
            int lv = -1;
            if (pos >= 0 && pos < npos) lv = (int)q[pos];
            slev[e] = (lv >= 0 && lv < classes) ? lv : -1;
        }
        __syncthreads();
        const int n = min(64, p1 - c0);
        int t = t0;
        for (int pp = 0; pp < n; ++pp) {
            const float g = stage[col * 65 + pp];
            if (quarter == 0) table[classes * kEmbTile + col] += g;                       // bias gradient row (tap 0 block)
            for (int j = 0; j < k; ++j) {
                if (t + j - (k - 1) < 0) continue;                                       // tap reaches before the utterance
                const int lv = slev[pp + j];
                if (lv >= 0 && lv / cq == quarter) table[(j * (classes + 1) + lv) * kEmbTile + col] += g;
            }
            if (++t == L) t = 0;
        }
    }
    __syncthreads();
    float* dst = partial + ((long long)slab * gridDim.y + ct) * k * (classes + 1) * kEmbTile;
    for (int i = tid; i < k * (classes + 1) * kEmbTile; i += 256) dst[i] = table[i];
}

__global__ __launch_bounds__(256) void embed_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dW,
                                                           float* __restrict__ db, int nslab, int ntile, int classes, int Co,
                                                           int k) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;   // over [tile][tap][class + 1][T]
    const long long per_tile = (long long)k * (classes + 1) * kEmbTile;
    if (idx >= per_tile * ntile) return;
    const int ct = (int)(idx / per_tile);
    const long long rem = idx - ct * per_tile;
    const int j = (int)(rem / ((classes + 1) * kEmbTile));
    const int cl = (int)((rem / kEmbTile) % (classes + 1)), col = (int)(rem % kEmbTile);
    const int co = ct * kEmbTile + col;
    float v = 0.0f;
    for (int s = 0; s < nslab; ++s) v += partial[((long long)s * ntile + ct) * per_tile + rem];   // fixed order
    if (co >= Co) return;
    if (cl < classes) dW[((long long)co * classes + cl) * k + j] = v;
    else if (j == 0 && db) db[co] = v;
}

}  // namespace wn

namespace wn {
int hip_fail_shared(hipError_t e, const char* what);
struct ProfScopeShared { void* impl; ProfScopeShared(int kc, double flops, hipStream_t st); ~ProfScopeShared(); };
}
using namespace wn;
static const int KC_EMBED = 17;   // index into wn_api.hip's kernel-class table

static int check_embed(int batch, int length, int classes, int out_channels, int k) {
    if (batch <= 0 || length <= 0 || classes <= 0 || out_channels <= 0 || k < 1) return WN_ERR_BAD_SHAPE;
    if (k > WN_MAX_TAPS || classes > 512 || out_channels > WN_MAX_CHANNELS) return WN_ERR_UNSUPPORTED;
    if ((size_t)k * (classes + 1) * kEmbTile * 4 + kEmbTile * 65 * 4 + (64 + WN_MAX_TAPS) * 4 > 160 * 1024) return WN_ERR_UNSUPPORTED;
    if ((double)batch * (double)length >= 2147483648.0) return WN_ERR_UNSUPPORTED;   // flat positions are 32-bit in the kernels
    return WN_OK;
}

int wn_embed_forward(const long long* levels, const float* weight, const float* bias, float* y, int batch, int length,
                     int classes, int out_channels, int kernel_width, int* bad_levels, wn_stream_t stream) {
    int rc = check_embed(batch, length, classes, out_channels, kernel_width);
    if (rc != WN_OK) return rc;
    if (!levels || !weight || !y) return WN_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    const long long n = (long long)batch * ((length + 3) / 4);
    const dim3 grid((unsigned)((n + 255) / 256), (unsigned)((out_channels + kEmbCo - 1) / kEmbCo));
    ProfScopeShared prof(KC_EMBED, 0.0, st);
    hipLaunchKernelGGL(embed_forward_kernel, grid, dim3(256), 0, st, levels, weight, bias, y, batch, length, classes, out_channels,
                       kernel_width, bad_levels);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail_shared(e, "embed_forward");
    return WN_OK;
}

size_t wn_embed_workspace_bytes(int batch, int length, int classes, int out_channels, int kernel_width) {
    if (check_embed(batch, length, classes, out_channels, kernel_width) != WN_OK) return 0;
    const int ntile = (out_channels + kEmbTile - 1) / kEmbTile;
    return (size_t)kEmbSlabs * ntile * kernel_width * (classes + 1) * kEmbTile * 4;
}

int wn_embed_backward(const long long* levels, const float* dy, float* dweight, float* dbias, void* workspace,
                      size_t workspace_bytes, int batch, int length, int classes, int out_channels, int kernel_width,
                      wn_stream_t stream) {
    int rc = check_embed(batch, length, classes, out_channels, kernel_width);
    if (rc != WN_OK) return rc;
    if (!levels || !dy || !dweight || !workspace) return WN_ERR_NULL;
    if (workspace_bytes < wn_embed_workspace_bytes(batch, length, classes, out_channels, kernel_width)) return WN_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int ntile = (out_channels + kEmbTile - 1) / kEmbTile;
    const long long npos = (long long)batch * length;
    const int per_slab = (int)((npos + kEmbSlabs - 1) / kEmbSlabs);
    const size_t lds_bytes = (size_t)kernel_width * (classes + 1) * kEmbTile * 4 + kEmbTile * 65 * 4 + (64 + WN_MAX_TAPS) * 4;
    ProfScopeShared prof(KC_EMBED, 0.0, st);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(embed_backward_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds_bytes);
    if (e != hipSuccess) return hip_fail_shared(e, "embed_backward attribute");
    hipLaunchKernelGGL(embed_backward_kernel, dim3(kEmbSlabs, ntile), dim3(256), lds_bytes, st, levels, dy, (float*)workspace, batch,
                       length, classes, out_channels, kernel_width, per_slab);
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail_shared(e, "embed_backward");
    const long long nred = (long long)ntile * kernel_width * (classes + 1) * kEmbTile;
    hipLaunchKernelGGL(embed_reduce_kernel, dim3((unsigned)((nred + 255) / 256)), dim3(256), 0, st, (const float*)workspace, dweight,
                       dbias, kEmbSlabs, ntile, classes, out_channels, kernel_width);
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail_shared(e, "embed_reduce");
    return WN_OK;
}
