// How fast does a CU deliver hwgrad_kernel's transposed fragment reads (ds_read_b64_tr_b16) compared with plain ds_read_b128 /
// ds_read_b64?  Four waves per workgroup (one per SIMD, as hwgrad_kernel), every CU busy, 4096 back-to-back reads per wave at the
// kernel's own address pattern; bytes per clock per CU from s_memtime.  Measurement only.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_tr_rate tools/probes/lds_tr_rate.hip && /tmp/lds_tr_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4v __attribute__((__vector_size__(4 * sizeof(short))));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int KIND>   // 0: ds_read_b64_tr_b16 (hwgrad's pattern), 1: ds_read_b64 (same addresses), 2: ds_read_b128 (lane * 16)
__global__ __launch_bounds__(256, 1) void k(unsigned long long* cyc, unsigned* sink, int iters) {
    __shared__ __attribute__((aligned(1024))) char lds[65536];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 65536 / 4; i += 256) reinterpret_cast<unsigned*>(lds)[i] = i;
    __syncthreads();
    const int kb = lane >> 4, q4 = (lane >> 2) & 3, pp = lane & 3;
    const unsigned rd = (unsigned)((32 * (kb >> 1) + 8 * (pp >> 1) + 4 * (kb & 1) + q4) * 16 + 8 * (pp & 1));   // wn_half_wgrad.hip
    const char* base = lds + wave * 16384 + (KIND == 2 ? lane * 16 : rd);
    unsigned acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const char* p = base + j * 1024 + (it & 1) * 256;
            if constexpr (KIND == 0) {
                const s4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(p));
                acc ^= __builtin_bit_cast(u32x2, v)[0] + __builtin_bit_cast(u32x2, v)[1];
            } else if constexpr (KIND == 1) {
                const u32x2 v = *reinterpret_cast<const u32x2*>(p);
                acc ^= v[0] + v[1];
            } else {
                const u32x4 v = *reinterpret_cast<const u32x4*>(p);
                acc ^= v[0] + v[1] + v[2] + v[3];
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
    sink[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
    unsigned long long* d; unsigned* s;
    hipMalloc(&d, 8 * 4 * 256); hipMalloc(&s, 4 * 256 * 256);
    const int iters = 256;
    const char* names[3] = {"ds_read_b64_tr_b16 (hwgrad pattern)", "ds_read_b64 (same addresses)", "ds_read_b128 (lane * 16)"};
    for (int kind = 0; kind < 3; ++kind) {
        for (int rep = 0; rep < 2; ++rep) {
            if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, d, s, iters);
            else if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, d, s, iters);
            else hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, d, s, iters);
        }
        unsigned long long h[1024]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        double mean = 0; for (int i = 0; i < 1024; ++i) mean += (double)h[i]; mean /= 1024;
        const double bytes_per_wave = (double)iters * 16 * 64 * (kind == 2 ? 16 : 8);
        // s_memtime counts a constant 100 MHz clock on this chip: convert with the kernel's wall time instead when in doubt
        printf("%-40s %.0f memtime ticks per wave for %.0f KB  -> %.2f bytes per tick per wave, %.2f per CU (4 waves)\n", names[kind], mean,
               bytes_per_wave / 1024, bytes_per_wave / mean, 4 * bytes_per_wave / mean);
    }
    // wall-clock version: total bytes / kernel time
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int kind = 0; kind < 3; ++kind) {
        hipEventRecord(e0);
        for (int rep = 0; rep < 20; ++rep) {
            if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, d, s, 4096);
            else if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, d, s, 4096);
            else hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, d, s, 4096);
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double bytes = 20.0 * 256 * 4 * 4096.0 * 16 * 64 * (kind == 2 ? 16 : 8);
        printf("%-40s %.1f TB/s chip-wide = %.1f bytes per ns per CU (at 2.1 GHz: %.1f per clock)\n", names[kind], bytes / (ms * 1e-3) / 1e12,
               bytes / (ms * 1e-3) / 1e9 / 256, bytes / (ms * 1e-3) / 1e9 / 256 / 2.1);
    }
    return 0;
}
