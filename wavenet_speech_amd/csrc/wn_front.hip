// The front-ends either side of the residual stack that are NOT GEMM-shaped, as plain HBM-bound kernels
// that read the caller's dense tensors and write the stack's own layouts -- no intermediate dense tensor, no layout copy:
//
//  * RawCTCNet.feature_layer[0..1] (reference modules/raw_ctcnet.py:57-61,128): Conv1d(1 -> F, k, padding k - 1) + LeakyReLU on the
//    raw one-channel signal.  With one input channel the conv is k multiply-adds per output element: elementwise work (the series
//    GEMM ran it with 31 of its 32 K rows zero).  hfeature_fwd_kernel writes leaky(conv(x)) straight into the half series of
//    length L + k - 1; hfeature_wgrad_kernel forms dW [F][k] and db [F] from the series gradient and the raw signal
//    (deterministic: per-slab partial sums, reduced in slab order).
//  * WaveNetClassifier.mean_pool (reference modules/classifier.py:53,102): AvgPool1d(pool) in front of the stack, fused into the
//    load of the stack's input series (pool_load kernels: dense [B][C][L] -> series of length L / pool) and its backward into
//    the scatter of the stack's input gradient (each pooled gradient divided by pool, broadcast to its pool columns).
#include <cstring>

#include "../../include/wavenet_amd.h"
#include "wn_half.h"
#include "wn_half_dev.h"

namespace wn {

constexpr int kFeatSlab = 4096;     // positions (utterance x time) per partial-sum slab of the feature wgrad

struct HFeatArgs {
    const float* x;         // dense [B][L] (one input channel)
    const float* w;         // [F][1][k]
    const float* b;         // [F]
    char* y;                // half series [B][P][G][ld][8], valid length L + k - 1
    unsigned* flag;
    float scale, slope;
    int B, L, Lout, F, G, k, ld, halo;
};

template <int P, bool BF>
__global__ __launch_bounds__(256) void hfeature_fwd_kernel(const HFeatArgs a) {
    // one thread = one 16-byte unit (8 features of one time step); lanes run along time
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long per_b = (long long)a.G * a.Lout;
    if (idx >= per_b * a.B) return;
    const int b = (int)(idx / per_b);
    const long long rem = idx - (long long)b * per_b;
    const int g = (int)(rem / a.Lout), t = (int)(rem - (long long)g * a.Lout);
    typedef typename HT<BF>::t T;
    typedef typename HT<BF>::v8 V8;
    float xs[WN_MAX_TAPS];
#pragma unroll
    for (int j = 0; j < WN_MAX_TAPS; ++j) {
        const int ti = t + j - (a.k - 1);
        xs[j] = (j < a.k && ti >= 0 && ti < a.L) ? a.x[(long long)b * a.L + ti] : 0.0f;
    }
    float v[8];
    unsigned ovf = 0;
#pragma unroll
    for (int c8 = 0; c8 < 8; ++c8) {
        const int c = 8 * g + c8;
        float acc = 0.0f;
        if (c < a.F) {
            acc = a.b ? a.b[c] : 0.0f;
#pragma unroll
            for (int j = 0; j < WN_MAX_TAPS; ++j)
                if (j < a.k) acc = __builtin_fmaf(a.w[(long long)c * a.k + j], xs[j], acc);
        }
        const float y = acc * a.scale;
        // y <= 0 is stored with the sign bit set (-0 for 0): the stored activation is the LeakyReLU mask of the backward pass (wn_half.h).
        // Pad channels stay exactly +0.
        v[c8] = c < a.F ? (y > 0.0f ? y : -__builtin_fabsf(y * a.slope)) : 0.0f;
        if (!BF) ovf |= (!(__builtin_fabsf(v[c8]) <= 65504.0f)) ? 1u : 0u;
    }
    V8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) hi[j] = (T)v[j];
    hi = pin(hi);
#pragma unroll
    for (int j = 0; j < 8; ++j) lo[j] = (T)(v[j] - (float)hi[j]);
    const long long pstride = (long long)a.G * a.ld * 16;
    char* d = a.y + (long long)b * P * pstride + ((long long)g * a.ld + a.halo + t) * 16;
    *reinterpret_cast<V8*>(d) = hi;
    if (P == 2) *reinterpret_cast<V8*>(d + pstride) = lo;
    if (ovf && a.flag) atomicOr(a.flag, 1u);
}

struct HFeatWgradArgs {
    const float* x;         // dense [B][L]
    const char* dy;         // half series gradient, valid length Lout
    float* partial;         // [nslab][G][8][k + 1]   (row k = bias gradient)
    int B, L, Lout, G, k, ld, halo, nslab;
};

// workgroup = (slab of kFeatSlab positions, feature group of 8); thread = positions tid, tid + 256, ...; per thread 8 x (k + 1) sums,
// folded across the workgroup through LDS in a fixed order
template <int P, bool BF>
__global__ __launch_bounds__(256) void hfeature_wgrad_kernel(const HFeatWgradArgs a) {
    typedef typename HT<BF>::v8 V8;
    const int slab = blockIdx.x, g = blockIdx.y, tid = threadIdx.x;
    const long long npos = (long long)a.B * a.Lout;
    const long long p0 = (long long)slab * kFeatSlab;
    const long long p1 = p0 + kFeatSlab < npos ? p0 + kFeatSlab : npos;
    const long long pstride = (long long)a.G * a.ld * 16;
    float s[8][WN_MAX_TAPS + 1];
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int j = 0; j <= WN_MAX_TAPS; ++j) s[c][j] = 0.0f;
    for (long long p = p0 + tid; p < p1; p += 256) {
        const int b = (int)(p / a.Lout), t = (int)(p - (long long)b * a.Lout);
        const char* src = a.dy + (long long)b * P * pstride + ((long long)g * a.ld + a.halo + t) * 16;
        const V8 hi = *reinterpret_cast<const V8*>(src);
        float dv[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) dv[c] = (float)hi[c];
        if (P == 2) {
            const V8 lo = *reinterpret_cast<const V8*>(src + pstride);
#pragma unroll
            for (int c = 0; c < 8; ++c) dv[c] += (float)lo[c];
        }
#pragma unroll
        for (int j = 0; j < WN_MAX_TAPS; ++j) {
            if (j < a.k) {
                const int ti = t + j - (a.k - 1);
                const float xv = (ti >= 0 && ti < a.L) ? a.x[(long long)b * a.L + ti] : 0.0f;
#pragma unroll
                for (int c = 0; c < 8; ++c) s[c][j] = __builtin_fmaf(dv[c], xv, s[c][j]);
            }
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) s[c][WN_MAX_TAPS] += dv[c];
    }
    // fixed-order fold: lanes of a wave by shuffles, the four waves through LDS
    __shared__ float red[4][8][WN_MAX_TAPS + 1];
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int j = 0; j <= WN_MAX_TAPS; ++j) {
            float v = s[c][j];
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
            if ((tid & 63) == 0) red[tid >> 6][c][j] = v;
        }
    __syncthreads();
    if (tid < 8 * (WN_MAX_TAPS + 1)) {
        const int c = tid / (WN_MAX_TAPS + 1), j = tid % (WN_MAX_TAPS + 1);
        const float v = ((red[0][c][j] + red[1][c][j]) + red[2][c][j]) + red[3][c][j];
        const int jj = j == WN_MAX_TAPS ? a.k : j;                 // bias row is stored right after the k taps
        if (j == WN_MAX_TAPS || j < a.k)
            a.partial[(((long long)slab * a.G + g) * 8 + c) * (a.k + 1) + jj] = v;
    }
}

__global__ __launch_bounds__(256) void hfeature_wgrad_reduce_kernel(const float* partial, float* dw, float* db, const float* dyn_inv,
                                                                     float post, int nslab, int G, int F, int k) {
    const int idx = blockIdx.x * 256 + threadIdx.x;                // over [F][k + 1]
    if (idx >= F * (k + 1)) return;
    const int c = idx / (k + 1), j = idx % (k + 1);
    const int g = c >> 3, c8 = c & 7;
    float v = 0.0f;
    for (int s = 0; s < nslab; ++s) v += partial[(((long long)s * G + g) * 8 + c8) * (k + 1) + j];   // slab order: reproducible
    v *= post * (dyn_inv ? dyn_inv[0] : 1.0f);
    if (j < k) dw[(long long)c * k + j] = v;
    else if (db) db[c] = v;
}

// ---- AvgPool1d fused into the load of the stack's input ------------------------------------------------------------------------
struct PoolArgs {
    const float* src;       // dense [B][C][L]
    char* dst;              // half series (planes P) or fp32 padded series
    float* dense;           // backward: dense gradient [B][C][L]
    unsigned* flag;
    const float* dyn;
    float scale;
    int B, C, L, Lp, pool, G, Cp, ld, halo;
};

template <int P, bool BF>
__global__ __launch_bounds__(256) void hpool_load_kernel(const PoolArgs a) {
    // one thread = one unit (8 channels of one POOLED time step): pool strided reads per channel, lanes along pooled time
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long per_b = (long long)a.G * a.Lp;
    if (idx >= per_b * a.B) return;
    const int b = (int)(idx / per_b);
    const long long rem = idx - (long long)b * per_b;
    const int g = (int)(rem / a.Lp), t = (int)(rem - (long long)g * a.Lp);
    typedef typename HT<BF>::t T;
    typedef typename HT<BF>::v8 V8;
    const float s = a.scale * (a.dyn ? a.dyn[0] : 1.0f) / (float)a.pool;
    V8 hi, lo;
    float v[8];
    unsigned ovf = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = 8 * g + j;
        float acc = 0.0f;
        if (c < a.C) {
            const float* p = a.src + ((long long)b * a.C + c) * a.L + (long long)t * a.pool;
            for (int q = 0; q < a.pool; ++q) acc += p[q];
        }
        v[j] = acc * s;
        hi[j] = (T)v[j];
        if (!BF) ovf |= (!(__builtin_fabsf(v[j]) <= 65504.0f)) ? 1u : 0u;
    }
    hi = pin(hi);
#pragma unroll
    for (int j = 0; j < 8; ++j) lo[j] = (T)(v[j] - (float)hi[j]);
    const long long pstride = (long long)a.G * a.ld * 16;
    char* d = a.dst + (long long)b * P * pstride + ((long long)g * a.ld + a.halo + t) * 16;
    *reinterpret_cast<V8*>(d) = hi;
    if (P == 2) *reinterpret_cast<V8*>(d + pstride) = lo;
    if (ovf && a.flag) atomicOr(a.flag, 1u);
}

// fp32 padded series [B][Cp][ld]: thread = (b, c, pooled t), lanes along pooled time
__global__ __launch_bounds__(256) void pool_load_kernel(const PoolArgs a) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long per_b = (long long)a.C * a.Lp;
    if (idx >= per_b * a.B) return;
    const int b = (int)(idx / per_b);
    const long long rem = idx - (long long)b * per_b;
    const int c = (int)(rem / a.Lp), t = (int)(rem - (long long)c * a.Lp);
    const float* p = a.src + ((long long)b * a.C + c) * a.L + (long long)t * a.pool;
    float acc = 0.0f;
    for (int q = 0; q < a.pool; ++q) acc += p[q];
    reinterpret_cast<float*>(a.dst)[((long long)b * a.Cp + c) * a.ld + a.halo + t] = acc / (float)a.pool;
}

// backward of the pooled load: dense dx[b][c][t] = dpooled[b][c][t / pool] / pool for t < Lp * pool, 0 for the dropped tail.
// src = the pooled gradient as DENSE fp32 [B][C][Lp] (what both stack paths hand back for their input)
__global__ __launch_bounds__(256) void pool_unload_kernel(const PoolArgs a) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long per_b = (long long)a.C * a.L;
    if (idx >= per_b * a.B) return;
    const int b = (int)(idx / per_b);
    const long long rem = idx - (long long)b * per_b;
    const int c = (int)(rem / a.L), t = (int)(rem - (long long)c * a.L);
    const int tp = t / a.pool;
    a.dense[idx] = tp < a.Lp ? a.src[((long long)b * a.C + c) * a.Lp + tp] / (float)a.pool : 0.0f;
}

// ---- dynamic power-of-two gradient scale of the fp16 modes ------------------------------------------------------------------------
// scale = 2^floor(log2(target / max|x|)), clamped to 2^+-100, and its reciprocal, as two device floats: one read of x (the torch
// expression it replaces -- abs, amax, clamp, div, log2, floor, clamp, exp2, reciprocal -- was nine launches and wrote |x| out).
// max is order-independent: an integer atomicMax on the bit pattern of |x| (monotonic for non-negative floats) is deterministic.
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long long n, unsigned* __restrict__ acc) {
    const long long n4 = n >> 2;
    float m = 0.0f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
        m = __builtin_fmaxf(m, __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(v[0]), __builtin_fabsf(v[1])),
                                               __builtin_fmaxf(__builtin_fabsf(v[2]), __builtin_fabsf(v[3]))));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = __builtin_fmaxf(m, __builtin_fabsf(x[(n4 << 2) + threadIdx.x]));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = __builtin_fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(acc, __builtin_bit_cast(unsigned, m));   // NaN bit patterns order above inf: they win, as in torch
}

__global__ void grad_scale_kernel(unsigned* acc, float target, float* out) {
    const float amax = __builtin_fmaxf(__builtin_bit_cast(float, acc[0]), 1e-30f);
    float e = __builtin_floorf(__builtin_log2f(target / amax));
    e = __builtin_fminf(__builtin_fmaxf(e, -100.0f), 100.0f);
    const float s = __builtin_exp2f(e);
    out[0] = s;
    out[1] = 1.0f / s;
    acc[0] = 0;                                                   // ready for the next call
}

}  // namespace wn

namespace wn {
int hip_fail_shared(hipError_t e, const char* what);
struct ProfScopeShared { void* impl; ProfScopeShared(int kc, double flops, hipStream_t st); ~ProfScopeShared(); };
}
using namespace wn;
static const int KC_FRONT = 10;   // timed with the layout loads (hload_kernel class of wn_api.hip's table)

namespace {
inline int cp32i(int c) { return (c + 31) / 32 * 32; }
bool halfp(int p) { return p == WN_F16X3 || p == WN_F16 || p == WN_BF16; }
int check_feat(int precision, int batch, int length, int features, int k, int ld, int halo) {
    if (!halfp(precision)) return WN_ERR_UNSUPPORTED;
    if (batch <= 0 || length <= 0 || features <= 0 || k < 1) return WN_ERR_BAD_SHAPE;
    if (k > WN_MAX_TAPS || features > WN_MAX_CHANNELS) return WN_ERR_UNSUPPORTED;
    const int lout = length + k - 1;
    if (halo < 0 || ld < 2 * halo + (lout + 255) / 256 * 256) return WN_ERR_BAD_SHAPE;
    return WN_OK;
}
}  // namespace

int wn_hfeature_forward(int precision, const float* x, const float* weight, const float* bias, void* y_series, int batch, int length,
                        int features, int kernel_width, int ld, int halo, float out_scale, float leaky_slope, unsigned* overflow_flag,
                        wn_stream_t stream) {
    int rc = check_feat(precision, batch, length, features, kernel_width, ld, halo);
    if (rc != WN_OK) return rc;
    if (!x || !weight || !y_series) return WN_ERR_NULL;
    if (!(out_scale > 0.0f)) return WN_ERR_BAD_SHAPE;
    HFeatArgs a;
    std::memset(&a, 0, sizeof(a));
    a.x = x; a.w = weight; a.b = bias; a.y = (char*)y_series; a.flag = overflow_flag; a.scale = out_scale; a.slope = leaky_slope;
    a.B = batch; a.L = length; a.Lout = length + kernel_width - 1; a.F = features; a.G = cp32i(features) / 8; a.k = kernel_width;
    a.ld = ld; a.halo = halo;
    hipStream_t st = (hipStream_t)stream;
    const long long n = (long long)a.B * a.G * a.Lout;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    ProfScopeShared prof(KC_FRONT, 0.0, st);
    if (precision == WN_F16X3) hipLaunchKernelGGL((hfeature_fwd_kernel<2, false>), grid, block, 0, st, a);
    else if (precision == WN_BF16) hipLaunchKernelGGL((hfeature_fwd_kernel<1, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((hfeature_fwd_kernel<1, false>), grid, block, 0, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail_shared(e, "hfeature_fwd");
    return WN_OK;
}

size_t wn_hfeature_wgrad_workspace_bytes(int batch, int length, int features, int kernel_width) {
    if (batch <= 0 || length <= 0 || features <= 0 || kernel_width < 1 || kernel_width > WN_MAX_TAPS) return 0;
    const long long npos = (long long)batch * (length + kernel_width - 1);
    const long long nslab = (npos + kFeatSlab - 1) / kFeatSlab;
    return (size_t)nslab * (cp32i(features) / 8) * 8 * (kernel_width + 1) * sizeof(float);
}

int wn_hfeature_backward_weights(int precision, const float* x, const void* dy_series, float dy_scale, float* dweight, float* dbias,
                                 int batch, int length, int features, int kernel_width, int ld, int halo, const float* dyn_inv_scale,
                                 void* workspace, size_t workspace_bytes, wn_stream_t stream) {
    int rc = check_feat(precision, batch, length, features, kernel_width, ld, halo);
    if (rc != WN_OK) return rc;
    if (!x || !dy_series || !dweight || !workspace) return WN_ERR_NULL;
    if (!(dy_scale > 0.0f)) return WN_ERR_BAD_SHAPE;
    if (workspace_bytes < wn_hfeature_wgrad_workspace_bytes(batch, length, features, kernel_width)) return WN_ERR_WORKSPACE;
    HFeatWgradArgs a;
    std::memset(&a, 0, sizeof(a));
    a.x = x; a.dy = (const char*)dy_series; a.partial = (float*)workspace;
    a.B = batch; a.L = length; a.Lout = length + kernel_width - 1; a.G = cp32i(features) / 8; a.k = kernel_width; a.ld = ld; a.halo = halo;
    const long long npos = (long long)a.B * a.Lout;
    a.nslab = (int)((npos + kFeatSlab - 1) / kFeatSlab);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)a.nslab, (unsigned)a.G), block(256);
    ProfScopeShared prof(KC_FRONT, 0.0, st);
    if (precision == WN_F16X3) hipLaunchKernelGGL((hfeature_wgrad_kernel<2, false>), grid, block, 0, st, a);
    else if (precision == WN_BF16) hipLaunchKernelGGL((hfeature_wgrad_kernel<1, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((hfeature_wgrad_kernel<1, false>), grid, block, 0, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail_shared(e, "hfeature_wgrad");
    const int n = features * (kernel_width + 1);
    hipLaunchKernelGGL(hfeature_wgrad_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float*)workspace, dweight,
                       dbias, dyn_inv_scale, 1.0f / dy_scale, a.nslab, a.G, features, kernel_width);
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail_shared(e, "hfeature_wgrad_reduce");
    return WN_OK;
}

// ---- pooled loads ---------------------------------------------------------------------------------------------------------------
int wn_hseries_load_pooled(int precision, const float* dense, void* series, int batch, int channels, int length, int pool, int ld, int halo,
                           float scale, const float* dyn_scale, unsigned* overflow_flag, wn_stream_t stream) {
    if (!halfp(precision)) return WN_ERR_UNSUPPORTED;
    if (!dense || !series) return WN_ERR_NULL;
    if (batch <= 0 || channels <= 0 || length <= 0 || pool < 1 || length / pool < 1 || channels > WN_MAX_CHANNELS) return WN_ERR_BAD_SHAPE;
    const int lp = length / pool;
    if (halo < 0 || ld < 2 * halo + (lp + 255) / 256 * 256) return WN_ERR_BAD_SHAPE;
    PoolArgs a;
    std::memset(&a, 0, sizeof(a));
    a.src = dense; a.dst = (char*)series; a.flag = overflow_flag; a.dyn = dyn_scale; a.scale = scale;
    a.B = batch; a.C = channels; a.L = length; a.Lp = lp; a.pool = pool; a.G = cp32i(channels) / 8; a.ld = ld; a.halo = halo;
    hipStream_t st = (hipStream_t)stream;
    const long long n = (long long)a.B * a.G * a.Lp;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    ProfScopeShared prof(KC_FRONT, 0.0, st);
    if (precision == WN_F16X3) hipLaunchKernelGGL((hpool_load_kernel<2, false>), grid, block, 0, st, a);
    else if (precision == WN_BF16) hipLaunchKernelGGL((hpool_load_kernel<1, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((hpool_load_kernel<1, false>), grid, block, 0, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail_shared(e, "hpool_load");
    return WN_OK;
}

int wn_series_load_pooled(const float* dense, float* series, int batch, int channels, int length, int pool, int ld, int halo,
                          wn_stream_t stream) {
    if (!dense || !series) return WN_ERR_NULL;
    if (batch <= 0 || channels <= 0 || length <= 0 || pool < 1 || length / pool < 1) return WN_ERR_BAD_SHAPE;
    const int lp = length / pool;
    if (halo < 0 || (halo & 3) || (ld & 3) || ld < 2 * halo + (lp + 127) / 128 * 128) return WN_ERR_BAD_SHAPE;
    PoolArgs a;
    std::memset(&a, 0, sizeof(a));
    a.src = dense; a.dst = (char*)series;
    a.B = batch; a.C = channels; a.L = length; a.Lp = lp; a.pool = pool; a.Cp = (channels + 7) / 8 * 8; a.ld = ld; a.halo = halo;
    hipStream_t st = (hipStream_t)stream;
    const long long n = (long long)a.B * a.C * a.Lp;
    ProfScopeShared prof(KC_FRONT, 0.0, st);
    hipLaunchKernelGGL(pool_load_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail_shared(e, "pool_load");
    return WN_OK;
}

int wn_pool_backward(const float* dpooled, float* dx, int batch, int channels, int length, int pool, wn_stream_t stream) {
    if (!dpooled || !dx) return WN_ERR_NULL;
    if (batch <= 0 || channels <= 0 || length <= 0 || pool < 1 || length / pool < 1) return WN_ERR_BAD_SHAPE;
    PoolArgs a;
    std::memset(&a, 0, sizeof(a));
    a.src = dpooled; a.dense = dx;
    a.B = batch; a.C = channels; a.L = length; a.Lp = length / pool; a.pool = pool;
    hipStream_t st = (hipStream_t)stream;
    const long long n = (long long)a.B * a.C * a.L;
    ProfScopeShared prof(KC_FRONT, 0.0, st);
    hipLaunchKernelGGL(pool_unload_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail_shared(e, "pool_unload");
    return WN_OK;
}

int wn_grad_scale(const float* x, long long count, float target, float* scale_and_inverse, unsigned* accumulator, wn_stream_t stream) {
    if (!x || !scale_and_inverse || !accumulator) return WN_ERR_NULL;
    if (count <= 0 || !(target > 0.0f) || (reinterpret_cast<uintptr_t>(x) & 15)) return WN_ERR_BAD_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const long long n4 = count / 4;
    const unsigned grid = (unsigned)(n4 / 256 < 1 ? 1 : (n4 / 256 > 2048 ? 2048 : n4 / 256));
    ProfScopeShared prof(KC_FRONT, 0.0, st);
    hipLaunchKernelGGL(absmax_kernel, dim3(grid), dim3(256), 0, st, x, count, accumulator);
    hipLaunchKernelGGL(grad_scale_kernel, dim3(1), dim3(1), 0, st, accumulator, target, scale_and_inverse);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail_shared(e, "grad_scale");
    return WN_OK;
}
