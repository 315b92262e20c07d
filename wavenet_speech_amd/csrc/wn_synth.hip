// Synthetic nanopore-like reads on the device: the reference's on-line generator
// utils/gaussian_kmer_model.py:53-104 (gaussian_model_fn -> quantize_fn -> one_hot_fn) as three launches, float64 like
// the reference's numpy arithmetic:
//
//   bases    nucleotides 1..4, counter-based Philox4x32-10 (the reference draws them with numpy's global RNG)
//   signal   5-mer index of the window bases[p+2 .. p+6] (scipy generic_filter's centred window after the [4:-4] trim),
//            held `upsampling` samples; picoamps = mean[kmer] + stdv[kmer] * z, z ~ N(0,1) by Box-Muller on Philox (or
//            taken from a caller-supplied array: the deterministic part is then checked against the reference's fixture);
//            per-workgroup (sum, min, max) partials for the per-read normalisation
//   quantize (x - mean) / (max - min), mu-law with mu = num_levels, np.digitize against the caller's edge array
//            (torch.linspace(-1, 1, num_levels), the reference's np.linspace), then levels [B][L] int64 and, if asked,
//            the dense one-hot [B][num_levels][L] fp32 written as row-contiguous 16-byte stores
//
// All of it is byte work bound by the one-hot store (262 MB at 16 x 256 x 16000); with the level-index entry conv
// (wn_embed.hip) the one-hot is not needed at all and the generator moves 8 B per sample.
// The per-read mean is a fixed-order sum (per-workgroup partials combined in index order): bit-reproducible run to run.
#include "../../include/wavenet_amd.h"
#include "wn_kernels.h"

namespace wn {

constexpr int kSynTile = 256;   // samples per workgroup

struct SynPart { double sum, mn, mx; };

__device__ __forceinline__ void philox4x32_10(unsigned (&c)[4], unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
        const unsigned n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
        c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

// stream 0: nucleotides, stream 1: Gaussian noise (the stream number sits in the counter's top word)
__device__ __forceinline__ void draw(unsigned long long seed, unsigned stream, unsigned long long index, unsigned (&c)[4]) {
    c[0] = (unsigned)index; c[1] = (unsigned)(index >> 32); c[2] = 0u; c[3] = stream;
    philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32));
}

__global__ __launch_bounds__(256) void synth_bases_kernel(unsigned long long seed, long long n, long long* bases) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    unsigned c[4];
    draw(seed, 0u, (unsigned long long)i, c);
    bases[i] = 1 + (long long)(c[0] & 3u);
}

__global__ __launch_bounds__(kSynTile) void synth_signal_kernel(const long long* __restrict__ bases, int nbases, int L, int ups,
                                                                 const double* __restrict__ means, const double* __restrict__ stdvs,
                                                                 unsigned long long seed, const double* __restrict__ noise,
                                                                 double* __restrict__ pico, SynPart* __restrict__ part, int nblk,
                                                                 int* __restrict__ bad) {
    const int b = blockIdx.y, t = blockIdx.x * kSynTile + threadIdx.x;
    double x = 0.0;
    const bool on = t < L;
    if (on) {
        const int p = t / ups;                                      // k-mer p = window bases[p+2 .. p+6]
        const long long* w = bases + (long long)b * nbases + p + 2;
        int kmer = 0;
        bool ok = true;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const long long nt = w[j];
            ok = ok && nt >= 1 && nt <= 4;
            kmer = kmer * 4 + (int)((nt - 1) & 3);
        }
        if (!ok && bad) atomicAdd(bad, 1);
        double z;
        const long long flat = (long long)b * L + t;
        if (noise) {
            z = noise[flat];
        } else {
            unsigned c[4];
            draw(seed, 1u, (unsigned long long)flat, c);
            const double u1 = ((double)(((unsigned long long)(c[0] >> 5) << 26) | (c[1] >> 6)) + 0.5) * (1.0 / 9007199254740992.0);
            const double u2 = ((double)(((unsigned long long)(c[2] >> 5) << 26) | (c[3] >> 6)) + 0.5) * (1.0 / 9007199254740992.0);
            z = sqrt(-2.0 * log(u1)) * cospi(2.0 * u2);
        }
        x = means[kmer] + stdvs[kmer] * z;
        pico[flat] = x;
    }
    // (sum, min, max) of the tile: wave shuffles, then the four wave results through LDS, combined in wave order
    double s = on ? x : 0.0, mn = on ? x : __builtin_huge_val(), mx = on ? x : -__builtin_huge_val();
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_down(s, o);
        mn = fmin(mn, __shfl_down(mn, o));
        mx = fmax(mx, __shfl_down(mx, o));
    }
    __shared__ double red[3][kSynTile / 64];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][wave] = s; red[1][wave] = mn; red[2][wave] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        SynPart r = {red[0][0], red[1][0], red[2][0]};
        for (int w = 1; w < kSynTile / 64; ++w) { r.sum += red[0][w]; r.mn = fmin(r.mn, red[1][w]); r.mx = fmax(r.mx, red[2][w]); }
        part[(long long)b * nblk + blockIdx.x] = r;
    }
}

template <bool VEC>
__global__ __launch_bounds__(kSynTile) void synth_quantize_kernel(const double* __restrict__ pico, const SynPart* __restrict__ part,
                                                                   int nblk, int L, int N, const double* __restrict__ edges,
                                                                   long long* __restrict__ levels, float* __restrict__ onehot) {
    const int b = blockIdx.y, t0 = blockIdx.x * kSynTile, t = t0 + threadIdx.x;
    // every thread folds the read's partials in the same (index) order: identical mean / range in every workgroup
    double sum = 0.0, mn = __builtin_huge_val(), mx = -__builtin_huge_val();
    for (int i = 0; i < nblk; ++i) {
        const SynPart p = part[(long long)b * nblk + i];
        sum += p.sum; mn = fmin(mn, p.mn); mx = fmax(mx, p.mx);
    }
    const double mean = sum / (double)L, span = mx - mn;
    __shared__ int lv[kSynTile];
    int level = -1;
    if (t < L) {
        const double x = pico[(long long)b * L + t];
        const double v = (x - mean) / span;                                 // gaussian_kmer_model.py:83
        const double mu = (double)N;
        const double m = copysign(log(1.0 + mu * fabs(v)) * (1.0 / log(1.0 + mu)), v);   // :38, the reference's own form (sign(0) * 0 = 0 either way)
        // np.digitize(m, edges): number of edges <= m (edges ascending) -- upper bound by bisection
        int lo = 0, hi = N;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (edges[mid] <= m) lo = mid + 1; else hi = mid;
        }
        level = lo < N ? lo : N - 1;                                        // the one-hot has N rows (m < 1 always: lo <= N-1)
        levels[(long long)b * L + t] = level;
    }
    if (!onehot) return;
    lv[threadIdx.x] = level;
    __syncthreads();
    // dense one-hot tile [N][256]: a wave writes one row segment of 256 samples as 64 x 16 B (4 rows per pass)
    float* ob = onehot + (long long)b * N * L + t0;
    if constexpr (VEC) {
        const int c4 = (threadIdx.x & 63) * 4, r0 = threadIdx.x >> 6;
        if (t0 + c4 < L) {                                                  // L % 4 == 0: a quad is wholly inside or outside
            const int l0 = lv[c4], l1 = lv[c4 + 1], l2 = lv[c4 + 2], l3 = lv[c4 + 3];
            for (int n = r0; n < N; n += kSynTile / 64) {
                const f32x4 v = {l0 == n ? 1.0f : 0.0f, l1 == n ? 1.0f : 0.0f, l2 == n ? 1.0f : 0.0f, l3 == n ? 1.0f : 0.0f};
                *reinterpret_cast<f32x4*>(ob + (long long)n * L + c4) = v;
            }
        }
    } else {
        if (t < L)
            for (int n = 0; n < N; ++n) ob[(long long)n * L + threadIdx.x] = level == n ? 1.0f : 0.0f;
    }
}

}  // namespace wn

namespace wn {
int hip_fail_shared(hipError_t e, const char* what);
struct ProfScopeShared { void* impl; ProfScopeShared(int kc, double flops, hipStream_t st); ~ProfScopeShared(); };
}
using namespace wn;
static const int KC_SYNTH = 18;   // index into wn_api.hip's kernel-class table

static int check_synth(int batch, int length) {
    if (batch <= 0 || length <= 0) return WN_ERR_BAD_SHAPE;
    if (batch > 65535 || (double)batch * (double)length >= 2147483648.0) return WN_ERR_UNSUPPORTED;
    return WN_OK;
}

size_t wn_synth_workspace_bytes(int batch, int length) {
    if (check_synth(batch, length) != WN_OK) return 0;
    return (size_t)batch * (size_t)((length + kSynTile - 1) / kSynTile) * sizeof(SynPart);
}

int wn_synth_bases(unsigned long long seed, int batch, int nbases, long long* bases, wn_stream_t stream) {
    if (batch <= 0 || nbases <= 0) return WN_ERR_BAD_SHAPE;
    if (!bases) return WN_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    const long long n = (long long)batch * nbases;
    ProfScopeShared prof(KC_SYNTH, 0.0, st);
    hipLaunchKernelGGL(synth_bases_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, seed, n, bases);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail_shared(e, "synth_bases");
    return WN_OK;
}

int wn_synth_signal(const long long* bases, int batch, int nbases, int length, int upsampling, const double* means,
                    const double* stdvs, unsigned long long seed, const double* noise, double* picoamps, void* workspace,
                    size_t workspace_bytes, int* bad_bases, wn_stream_t stream) {
    int rc = check_synth(batch, length);
    if (rc != WN_OK) return rc;
    if (upsampling < 1 || nbases < 9) return WN_ERR_BAD_SHAPE;
    if ((long long)(nbases - 8) * upsampling < length) return WN_ERR_BAD_SHAPE;   // n bases give (n - 8) * upsampling samples
    if (!bases || !means || !stdvs || !picoamps || !workspace) return WN_ERR_NULL;
    if (workspace_bytes < wn_synth_workspace_bytes(batch, length)) return WN_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int nblk = (length + kSynTile - 1) / kSynTile;
    ProfScopeShared prof(KC_SYNTH, 0.0, st);
    hipLaunchKernelGGL(synth_signal_kernel, dim3(nblk, batch), dim3(kSynTile), 0, st, bases, nbases, length, upsampling, means, stdvs,
                       seed, noise, picoamps, (SynPart*)workspace, nblk, bad_bases);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail_shared(e, "synth_signal");
    return WN_OK;
}

int wn_synth_quantize(const double* picoamps, const void* workspace, size_t workspace_bytes, int batch, int length, int num_levels,
                      const double* edges, long long* levels, float* one_hot, wn_stream_t stream) {
    int rc = check_synth(batch, length);
    if (rc != WN_OK) return rc;
    if (num_levels < 2 || num_levels > 65536) return WN_ERR_BAD_SHAPE;
    if (!picoamps || !workspace || !edges || !levels) return WN_ERR_NULL;
    if (workspace_bytes < wn_synth_workspace_bytes(batch, length)) return WN_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int nblk = (length + kSynTile - 1) / kSynTile;
    ProfScopeShared prof(KC_SYNTH, 0.0, st);
    const bool vec = one_hot && (length % 4 == 0) && ((reinterpret_cast<uintptr_t>(one_hot) & 15) == 0);
    if (vec)
        hipLaunchKernelGGL((synth_quantize_kernel<true>), dim3(nblk, batch), dim3(kSynTile), 0, st, picoamps, (const SynPart*)workspace,
                           nblk, length, num_levels, edges, levels, one_hot);
    else
        hipLaunchKernelGGL((synth_quantize_kernel<false>), dim3(nblk, batch), dim3(kSynTile), 0, st, picoamps, (const SynPart*)workspace,
                           nblk, length, num_levels, edges, levels, one_hot);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail_shared(e, "synth_quantize");
    return WN_OK;
}
