// Fused next-sample NLL head: sum over time of the batch-averaged cross entropy of logits [B][C][L] against integer
// targets [B][L] -- the reference computes it with an L-iteration Python loop of CrossEntropyLoss calls
// (Loss.py:38-43, legacy_code/train.py:37-39).
//
// Memory-bound elementwise/reduction work (HBM roofline): time is the contiguous axis, so each thread owns four
// consecutive time steps and walks the C channel rows with 16-byte loads (a wave reads 1 KB contiguous per row);
// an online max/sum gives log-sum-exp in ONE pass over the logits.  Forward writes lse[B][L] (kept for backward) and
// one partial loss per workgroup (summed in a fixed order by the caller: deterministic).  Backward is one more pass:
// dlogits = (softmax - onehot) * scale.
#include "wn_kernels.h"

namespace wn {

__global__ __launch_bounds__(256) void nll_forward_kernel(const float* __restrict__ logits, const long long* __restrict__ target,
                                                          float* __restrict__ lse, float* __restrict__ partial,
                                                          int* __restrict__ bad_targets, int B, int C, int L) {
    const int L4 = (L + 3) / 4;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    float loss = 0.0f;
    if (gid < (long long)B * L4) {
        const int b = (int)(gid / L4), t0 = (int)(gid - (long long)b * L4) * 4;
        const float* p = logits + (long long)b * C * L + t0;
        const bool vec = (L % 4 == 0);               // rows are 16-byte aligned only then
        float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, s[4] = {0.f, 0.f, 0.f, 0.f};
        // online log-sum-exp with ONE exponential per element: e = exp(-|v - m|) is either the new term (v <= m) or
        // the rescale of the running sum (v > m, new term = 1)
#pragma unroll 8
        for (int c = 0; c < C; ++c) {
            float v[4];
            if (vec) {
                const f32x4 q = *reinterpret_cast<const f32x4*>(p + (long long)c * L);
                v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = t0 + j < L ? p[(long long)c * L + j] : 0.0f;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = v[j] - m[j];
                const float e = __builtin_amdgcn_exp2f(-1.44269504088896341f * __builtin_fabsf(d));   // exp(-inf) = 0 on the first row
                s[j] = d > 0.0f ? s[j] * e + 1.0f : s[j] + e;
                m[j] = fmaxf(m[j], v[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (t0 + j < L) {
                const float l = m[j] + __logf(s[j]);
                lse[(long long)b * L + t0 + j] = l;
                long long tg = target[(long long)b * L + t0 + j];
                if (tg < 0 || tg >= C) {   // never index the logits with an unchecked label: count it, read class 0, poison the loss
                    if (bad_targets) atomicAdd(bad_targets, 1);
                    tg = 0;
                    loss = __builtin_nanf("");
                }
                loss += l - logits[((long long)b * C + tg) * L + t0 + j];
            }
        }
    }
    // workgroup sum in a fixed order: wave shuffles, then 4 partials through LDS
    __shared__ float red[4];
    for (int o = 32; o > 0; o >>= 1) loss += __shfl_down(loss, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = loss;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void nll_backward_kernel(const float* __restrict__ logits, const long long* __restrict__ target,
                                                           const float* __restrict__ lse, const float* __restrict__ gscale,
                                                           float* __restrict__ dlogits, int B, int C, int L) {
    const int L4 = (L + 3) / 4;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long long)B * L4) return;
    const int b = (int)(gid / L4), t0 = (int)(gid - (long long)b * L4) * 4;
    const float g = gscale[0];
    const bool vec = (L % 4 == 0);
    float l[4];
    long long tg[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bool ok = t0 + j < L;
        l[j] = ok ? lse[(long long)b * L + t0 + j] : 0.0f;
        tg[j] = ok ? target[(long long)b * L + t0 + j] : -1;
    }
    const float* p = logits + (long long)b * C * L + t0;
    float* d = dlogits + (long long)b * C * L + t0;
#pragma unroll 8
    for (int c = 0; c < C; ++c) {
        if (vec) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(p + (long long)c * L);
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (__expf(q[j] - l[j]) - (tg[j] == c ? 1.0f : 0.0f)) * g;
            *reinterpret_cast<f32x4*>(d + (long long)c * L) = o;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (t0 + j < L) d[(long long)c * L + j] = (__expf(p[(long long)c * L + j] - l[j]) - (tg[j] == c ? 1.0f : 0.0f)) * g;
        }
    }
}

hipError_t launch_nll_forward(const float* logits, const long long* target, float* lse, float* partial, int* bad_targets, int B, int C,
                              int L, hipStream_t st) {
    const long long n = (long long)B * ((L + 3) / 4);
    hipLaunchKernelGGL(nll_forward_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, logits, target, lse, partial,
                       bad_targets, B, C, L);
    return hipGetLastError();
}

hipError_t launch_nll_backward(const float* logits, const long long* target, const float* lse, const float* gscale, float* dlogits,
                               int B, int C, int L, hipStream_t st) {
    const long long n = (long long)B * ((L + 3) / 4);
    hipLaunchKernelGGL(nll_backward_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, logits, target, lse, gscale, dlogits, B, C, L);
    return hipGetLastError();
}

}  // namespace wn
