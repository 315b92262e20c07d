// EXPLORATION, not product code: can a 3-product split-fp16 MFMA GEMM beat the exact-fp32 MFMA path on MI355X, and is it
// accurate?   out[M x N] = W[M x K] * X[K x N],  W = Wh + Wl, X = Xh + Xl (fp16 halves, W pre-scaled by 2^8),
// out ~= Wh Xh + Wh Xl + Wl Xh with fp32 accumulation (v_mfma_f32_32x32x16_f16, 16x the fp32 MFMA rate).
// Same per-wave structure as series_gemm_kernel (128 rows x 128 columns, operands straight from L2, no LDS):
//   A (weights): packed [k16][row tile][plane][lane][8 halves]  -> one 16-byte load per lane per (tile, plane)
//   B (activations): channel-interleaved [K/8][ld][8 halves] per plane -> lane (n, h) loads 16 bytes = channels 8(2*k16+h)..+7 of column n
// hipcc --offload-arch=gfx950 -O3 -o f16x3_probe tools/f16x3_probe.hip && ./f16x3_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int NPROD>
__global__ __launch_bounds__(64) void probe(const h8* __restrict__ Wp, const h8* __restrict__ Xh, const h8* __restrict__ Xl,
                                            float* __restrict__ out, int K16, int ld, int N, int nslab) {
    const int lane = threadIdx.x, n = lane & 31, h = lane >> 5;
    const int slab = blockIdx.x % nslab, coltile = blockIdx.x / nslab;
    const int t0 = coltile * 128;
    f32x16 acc[4][4];
    for (int m = 0; m < 4; ++m) for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.f;
    const h8* wp = Wp + (size_t)slab * K16 * 4 * 2 * 64 + lane;            // [k16][m][plane][lane]
    const h8* xh = Xh + (size_t)h * ld + t0 + n;                            // [(2*k16+h)][column]
    const h8* xl = Xl + (size_t)h * ld + t0 + n;
    h8 Ah[2][4], Al[2][4], Bh[2][4], Bl[2][4];
    auto load = [&](int s, int k) {
        const int kc = k < K16 ? k : K16 - 1;
#pragma unroll
        for (int m = 0; m < 4; ++m) { Ah[s][m] = wp[((size_t)kc * 4 + m) * 2 * 64]; Al[s][m] = wp[((size_t)kc * 4 + m) * 2 * 64 + 64]; }
#pragma unroll
        for (int t = 0; t < 4; ++t) { Bh[s][t] = xh[(size_t)(2 * kc) * ld + 32 * t]; Bl[s][t] = xl[(size_t)(2 * kc) * ld + 32 * t]; }
    };
    auto compute = [&](int s) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[s][m], Bh[s][t], acc[m][t], 0, 0, 0);
                if (NPROD >= 3) {
                    acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[s][m], Bl[s][t], acc[m][t], 0, 0, 0);
                    acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al[s][m], Bh[s][t], acc[m][t], 0, 0, 0);
                }
            }
    };
    load(0, 0);
    for (int k = 0; k < K16; k += 2) {
        load(1, k + 1);
        __builtin_amdgcn_sched_barrier(0);
        compute(0);
        __builtin_amdgcn_sched_barrier(0);
        load(0, k + 2);
        __builtin_amdgcn_sched_barrier(0);
        compute(1);
        __builtin_amdgcn_sched_barrier(0);
    }
    // C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    for (int m = 0; m < 4; ++m) for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) {
        const int row = slab * 128 + 32 * m + (r & 3) + 8 * (r >> 2) + 4 * h;
        out[(size_t)row * N + t0 + 32 * t + n] = acc[m][t][r] * (1.0f / 256.0f);
    }
}

int main() {
    const int M = 512, K = 512, N = 128 * 2000, K16 = K / 16, nslab = M / 128, ld = N;
    std::vector<float> W((size_t)M * K), X((size_t)K * N);
    srand(1);
    for (auto& v : W) v = (rand() / (float)RAND_MAX - 0.5f) * 0.3f;
    for (auto& v : X) v = (rand() / (float)RAND_MAX - 0.5f) * 4.0f;
    // split + pack
    std::vector<_Float16> Wp((size_t)nslab * K16 * 4 * 2 * 64 * 8), Xh((size_t)(K / 8) * ld * 8), Xl((size_t)(K / 8) * ld * 8);
    for (int s = 0; s < nslab; ++s) for (int k = 0; k < K16; ++k) for (int m = 0; m < 4; ++m) for (int l = 0; l < 64; ++l) for (int j = 0; j < 8; ++j) {
        const float w = W[(size_t)(s * 128 + 32 * m + (l & 31)) * K + 16 * k + 8 * (l >> 5) + j] * 256.0f;
        const _Float16 hi = (_Float16)w, lo = (_Float16)(w - (float)hi);
        const size_t base = ((((size_t)s * K16 + k) * 4 + m) * 2) * 64 * 8;
        Wp[base + (size_t)l * 8 + j] = hi; Wp[base + 64 * 8 + (size_t)l * 8 + j] = lo;
    }
    for (int c = 0; c < K; ++c) for (int t = 0; t < N; ++t) {
        const float x = X[(size_t)c * N + t]; const _Float16 hi = (_Float16)x, lo = (_Float16)(x - (float)hi);
        Xh[((size_t)(c / 8) * ld + t) * 8 + c % 8] = hi; Xl[((size_t)(c / 8) * ld + t) * 8 + c % 8] = lo;
    }
    h8 *dW, *dXh, *dXl; float* dO;
    CK(hipMalloc(&dW, Wp.size() * 2)); CK(hipMalloc(&dXh, Xh.size() * 2)); CK(hipMalloc(&dXl, Xl.size() * 2)); CK(hipMalloc(&dO, (size_t)M * N * 4));
    CK(hipMemcpy(dW, Wp.data(), Wp.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dXh, Xh.data(), Xh.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dXl, Xl.data(), Xl.size() * 2, hipMemcpyHostToDevice));
    const int grid = nslab * (N / 128);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int nprod : {3, 1}) {
        for (int it = 0; it < 3; ++it) { if (nprod == 3) hipLaunchKernelGGL(probe<3>, dim3(grid), dim3(64), 0, 0, dW, dXh, dXl, dO, K16, ld, N, nslab); else hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(64), 0, 0, dW, dXh, dXl, dO, K16, ld, N, nslab); }
        CK(hipEventRecord(e0));
        const int reps = 10;
        for (int it = 0; it < reps; ++it) { if (nprod == 3) hipLaunchKernelGGL(probe<3>, dim3(grid), dim3(64), 0, 0, dW, dXh, dXl, dO, K16, ld, N, nslab); else hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(64), 0, 0, dW, dXh, dXl, dO, K16, ld, N, nslab); }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        std::vector<float> O((size_t)M * 4096);   // check the first 4096 columns of every row against fp64
        CK(hipMemcpy2D(O.data(), 4096 * 4, dO, (size_t)N * 4, 4096 * 4, M, hipMemcpyDeviceToHost));
        double maxerr = 0, maxref = 0;
        for (int r = 0; r < M; r += 7) for (int t = 0; t < 4096; t += 13) {
            double ref = 0; for (int c = 0; c < K; ++c) ref += (double)W[(size_t)r * K + c] * (double)X[(size_t)c * N + t];
            maxerr = fmax(maxerr, fabs(ref - O[(size_t)r * 4096 + t])); maxref = fmax(maxref, fabs(ref));
        }
        printf("%d-product split fp16: %.3f ms  = %.1f fp32-equivalent TFLOP/s (2MKN)   max rel err vs fp64 %.2e\n", nprod, ms,
               2.0 * M * K * (double)N / (ms * 1e-3) / 1e12, maxerr / maxref);
    }
    printf("(exact fp32 MFMA series GEMM at this size: 1.07 ms = 125 TFLOP/s, rel err ~1e-6)\n");
    return 0;
}
