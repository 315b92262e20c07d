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

// ---- backward ---------------------------------------------------------------------------------------------------------
// There is no gather-form backward here.  dW[co][class][j] = sum over the positions whose level is `class` of dy[co][t] is a
// contraction over 256 k positions: the host (functional._EmbedConvFn.backward) forms it with the exact-fp32 weight-gradient
// GEMM (wgrad_kernel, fixed summation order) against a one-hot that exists only inside the backward call -- 0.55 ms at
// 16 x 256 x 16000.  A per-class segmented-sum kernel that never builds a one-hot was written and measured in round 2
// (one writer per table cell, deterministic): 1.95 ms, instruction-bound; it was removed in round 3 rather than kept as a
// slower alternative.  The forward pass and the state saved for backward never hold a one-hot.
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
