// Does v_mfma_f32_32x32x16_f16 honour fp16 subnormal inputs (the lo plane of the f16x3 split is often subnormal)?
// hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_f16_denorm tools/probes/mfma_f16_denorm.hip && /tmp/mfma_f16_denorm
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float a_val, float b_val, float* out) {
    h8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)a_val; b[j] = (_Float16)b_val; }
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = acc[0]; out[1] = (float)a[0]; out[2] = (float)a[0] * (float)b[0] * 16.0f; }
}
int main() {
    float* d; hipMalloc(&d, 16);
    const float vals[] = {1.0f, 1e-3f, 6.2e-5f, 3.0e-5f, 1.0e-6f, 6.0e-8f};
    for (float v : vals) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, v, 2.0f, d);
        float h[3]; hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
        printf("a=%.3e (as fp16 %.6e) x b=2, K=16: mfma %.6e  expected %.6e  %s\n", v, h[1], h[0], h[2], h[0] == h[2] ? "ok" : "DIFFERENT");
    }
    return 0;
}
