// Stand-alone consumer of the C ABI (include/wavenet_amd.h): no torch, no Python -- plain HIP host code.
// Reads a residual-block problem from a little binary file, runs pack / forward / backward-data / backward-weights
// through libwavenet_amd.so on hipMalloc'd buffers and writes every result back.  tests/test_gpu_cabi.py produces the
// input from seeded data and checks the output against the CPU oracle.
//
//   hipcc --offload-arch=gfx950 -O2 -I include tests/cabi/cabi_block.cpp -L wavenet_speech_amd -lwavenet_amd -o cabi_block
//   ./cabi_block problem.bin result.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "wavenet_amd.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)
#define WN(x) do { int s_ = (x); if (s_ != WN_OK) { fprintf(stderr, "wn error %d (%s) [%s] at %s:%d\n", s_, wn_strerror(s_), wn_last_hip_error(), __FILE__, __LINE__); return 3; } } while (0)

static std::vector<float> rd(FILE* f, size_t n) { std::vector<float> v(n); if (fread(v.data(), 4, n, f) != n) { fprintf(stderr, "short read\n"); exit(4); } return v; }

// dense [B][C][L] <-> padded series [B][Cp][ld] on the host
static std::vector<float> to_series(const std::vector<float>& d, int B, int C, int L, int ld, int halo) {
    const int Cp = wn_round_up(C, 8);
    std::vector<float> s((size_t)B * Cp * ld, 0.0f);
    for (int b = 0; b < B; ++b) for (int c = 0; c < C; ++c) for (int t = 0; t < L; ++t)
        s[((size_t)b * Cp + c) * ld + halo + t] = d[((size_t)b * C + c) * L + t];
    return s;
}
static std::vector<float> from_series(const std::vector<float>& s, int B, int C, int L, int ld, int halo) {
    const int Cp = wn_round_up(C, 8);
    std::vector<float> d((size_t)B * C * L);
    for (int b = 0; b < B; ++b) for (int c = 0; c < C; ++c) for (int t = 0; t < L; ++t)
        d[((size_t)b * C + c) * L + t] = s[((size_t)b * Cp + c) * ld + halo + t];
    return d;
}
static float* up(const std::vector<float>& h) { float* d = nullptr; if (hipMalloc(&d, h.size() * 4) != hipSuccess) exit(5); if (hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice) != hipSuccess) exit(5); return d; }
static float* dzero(size_t n) { float* d = nullptr; if (hipMalloc(&d, n * 4) != hipSuccess) exit(5); if (hipMemset(d, 0, n * 4) != hipSuccess) exit(5); return d; }
static std::vector<float> down(const float* d, size_t n) { std::vector<float> h(n); if (hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost) != hipSuccess) exit(5); return h; }

int main(int argc, char** argv) {
    if (argc != 3) { fprintf(stderr, "usage: %s problem.bin result.bin\n", argv[0]); return 1; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror("open"); return 1; }
    int hdr[8];   // B, L, Ci, Co, k, d, causal, 0
    if (fread(hdr, 4, 8, f) != 8) return 4;
    const int B = hdr[0], L = hdr[1], Ci = hdr[2], Co = hdr[3], k = hdr[4], d = hdr[5], causal = hdr[6];
    int off[WN_MAX_TAPS], reach = 0;
    WN(wn_tap_offsets(k, d, causal, off));
    for (int j = 0; j < k; ++j) reach = abs(off[j]) > reach ? abs(off[j]) : reach;
    int ld = 0, halo = 0;
    WN(wn_series_layout(L, reach, &ld, &halo));
    wn_block_shape s = {B, L, Ci, Co, Co, k, d, causal, ld, halo};

    // parameters in PyTorch layouts, then x, then the cotangents of (residual_out, skip_out)
    std::vector<float> wt = rd(f, (size_t)Co * Ci * k), bt = rd(f, Co), ws = rd(f, (size_t)Co * Ci * k), bs = rd(f, Co),
                       wr = rd(f, (size_t)Co * Co), br = rd(f, Co), wk = rd(f, (size_t)Co * Co), bk = rd(f, Co),
                       wp = rd(f, (size_t)Co * Ci), bp = rd(f, Co);
    std::vector<float> x = rd(f, (size_t)B * Ci * L), cr = rd(f, (size_t)B * Co * L), cs = rd(f, (size_t)B * Co * L);
    fclose(f);

    wn_block_params P = {up(wt), up(bt), up(ws), up(bs), up(wr), up(br), up(wk), up(bk), up(wp), up(bp)};
    const size_t nx = wn_series_floats(B, Ci, ld), ny = wn_series_floats(B, Co, ld);
    float* dx_in = up(to_series(x, B, Ci, L, ld, halo));
    float *r = dzero(ny), *sk = dzero(ny), *sg = dzero(ny), *z = dzero(ny);
    float *dr = up(to_series(cr, B, Co, L, ld, halo)), *ds = up(to_series(cs, B, Co, L, ld, halo));
    float *da = dzero(ny), *dg = dzero(ny), *dx = dzero(nx);
    void* packed = nullptr;
    CK(hipMalloc(&packed, wn_block_packed_bytes(&s)));
    const size_t wsb = wn_block_wgrad_workspace_bytes(&s);
    void* wsp = nullptr;
    CK(hipMalloc(&wsp, wsb));
    wn_block_params G = {dzero((size_t)Co * Ci * k), dzero(Co), dzero((size_t)Co * Ci * k), dzero(Co), dzero((size_t)Co * Co), dzero(Co),
                         dzero((size_t)Co * Co), dzero(Co), dzero((size_t)Co * Ci), dzero(Co)};
    hipStream_t st;
    CK(hipStreamCreate(&st));
    WN(wn_block_pack(&s, &P, packed, st));
    WN(wn_block_forward(&s, packed, dx_in, r, sk, 0, sg, z, st));
    WN(wn_block_backward_data(&s, packed, dr, ds, z, sg, da, dg, dx, st));
    WN(wn_block_backward_weights(&s, dx_in, z, da, dg, dr, ds, &G, wsp, wsb, st));
    CK(hipStreamSynchronize(st));

    FILE* o = fopen(argv[2], "wb");
    if (!o) { perror("open out"); return 1; }
    auto wr_ = [&](const std::vector<float>& v) { fwrite(v.data(), 4, v.size(), o); };
    wr_(from_series(down(r, ny), B, Co, L, ld, halo));
    wr_(from_series(down(sk, ny), B, Co, L, ld, halo));
    wr_(from_series(down(dx, nx), B, Ci, L, ld, halo));
    wr_(down(G.w_tanh, (size_t)Co * Ci * k)); wr_(down(G.b_tanh, Co)); wr_(down(G.w_sigmoid, (size_t)Co * Ci * k)); wr_(down(G.b_sigmoid, Co));
    wr_(down(G.w_res, (size_t)Co * Co)); wr_(down(G.b_res, Co)); wr_(down(G.w_skip, (size_t)Co * Co)); wr_(down(G.b_skip, Co));
    wr_(down(G.w_proj, (size_t)Co * Ci)); wr_(down(G.b_proj, Co));
    // the zero padding of an output series must still be zero (layout invariant)
    std::vector<float> rs = down(r, ny);
    double padsum = 0;
    const int Cp = wn_round_up(Co, 8);
    for (int b = 0; b < B; ++b) for (int c = 0; c < Cp; ++c) for (int t = 0; t < ld; ++t)
        if (c >= Co || t < halo || t >= halo + L) padsum += rs[((size_t)b * Cp + c) * ld + t] != 0.0f;
    float pad = (float)padsum;
    fwrite(&pad, 4, 1, o);
    fclose(o);
    printf("cabi_block ok: B=%d L=%d Ci=%d Co=%d k=%d d=%d causal=%d ld=%d halo=%d version=%d\n", B, L, Ci, Co, k, d, causal, ld, halo, wn_version());
    return 0;
}
