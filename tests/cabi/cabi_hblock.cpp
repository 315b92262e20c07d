// Stand-alone consumer of the half-precision-MFMA entry points of the C ABI (include/wavenet_amd.h, wn_h*): no torch, no
// Python -- plain HIP host code.  Reads a residual-block problem, runs load / pack / forward / backward-data /
// backward-weights in the requested wn_precision on hipMalloc'd buffers and writes every result back
// (tests/test_gpu_cabi.py prepares the input and checks the output against the CPU oracle).
//
//   hipcc --offload-arch=gfx950 -O2 -I include tests/cabi/cabi_hblock.cpp -L wavenet_speech_amd -lwavenet_amd -o cabi_hblock
//   ./cabi_hblock problem.bin result.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "wavenet_amd.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)
#define WN(x) do { int s_ = (x); if (s_ != WN_OK) { fprintf(stderr, "wn error %d (%s) [%s] at %s:%d\n", s_, wn_strerror(s_), wn_last_hip_error(), __FILE__, __LINE__); return 3; } } while (0)

static std::vector<float> rd(FILE* f, size_t n) { std::vector<float> v(n); if (fread(v.data(), 4, n, f) != n) { fprintf(stderr, "short read\n"); exit(4); } return v; }
static float* up(const std::vector<float>& h) { float* d = nullptr; if (hipMalloc(&d, h.size() * 4) != hipSuccess) exit(5); if (hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice) != hipSuccess) exit(5); return d; }
static void* dzero(size_t bytes) { void* d = nullptr; if (hipMalloc(&d, bytes) != hipSuccess) exit(5); if (hipMemset(d, 0, bytes) != hipSuccess) exit(5); return d; }
template <typename T> static std::vector<T> down(const void* d, size_t n) { std::vector<T> h(n); if (hipMemcpy(h.data(), d, n * sizeof(T), hipMemcpyDeviceToHost) != hipSuccess) exit(5); return h; }

int main(int argc, char** argv) {
    if (argc != 3) { fprintf(stderr, "usage: %s problem.bin result.bin\n", argv[0]); return 1; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror("open"); return 1; }
    int hdr[8];   // B, L, Ci, Co, k, d, causal, precision
    if (fread(hdr, 4, 8, f) != 8) return 4;
    const int B = hdr[0], L = hdr[1], Ci = hdr[2], Co = hdr[3], k = hdr[4], d = hdr[5], causal = hdr[6], prec = hdr[7];
    int off[WN_MAX_TAPS], reach = 0;
    WN(wn_tap_offsets(k, d, causal, off));
    for (int j = 0; j < k; ++j) reach = abs(off[j]) > reach ? abs(off[j]) : reach;
    int ld = 0, halo = 0;
    WN(wn_hseries_layout(L, reach, &ld, &halo));
    wn_block_shape s = {B, L, Ci, Co, Co, k, d, causal, ld, halo};

    std::vector<float> wt = rd(f, (size_t)Co * Ci * k), bt = rd(f, Co), ws = rd(f, (size_t)Co * Ci * k), bs = rd(f, Co),
                       wr = rd(f, (size_t)Co * Co), br = rd(f, Co), wk = rd(f, (size_t)Co * Co), bk = rd(f, Co),
                       wp = rd(f, (size_t)Co * Ci), bp = rd(f, Co);
    std::vector<float> x = rd(f, (size_t)B * Ci * L), cr = rd(f, (size_t)B * Co * L), cs = rd(f, (size_t)B * Co * L);
    fclose(f);

    hipStream_t st;
    CK(hipStreamCreate(&st));
    wn_block_params p = {up(wt), up(bt), up(ws), up(bs), up(wr), up(br), up(wk), up(bk), up(wp), up(bp)};
    const size_t bytes_i = wn_hseries_bytes(prec, B, Ci, ld), bytes_o = wn_hseries_bytes(prec, B, Co, ld);
    if (!bytes_i || !bytes_o) { fprintf(stderr, "unsupported precision %d\n", prec); return 3; }
    unsigned* flag = (unsigned*)dzero(4);
    void* hx = dzero(bytes_i);
    float* dx_in = up(x);
    WN(wn_hseries_load(prec, dx_in, hx, B, Ci, L, ld, halo, wn_hseries_residual_scale(), nullptr, flag, st));
    const size_t pk = wn_hblock_packed_bytes(&s, prec);
    if (!pk) return 3;
    void* packed = dzero(pk);
    WN(wn_hblock_pack(&s, prec, &p, packed, st));
    void *r = dzero(bytes_o), *sg = dzero(bytes_o), *z = dzero(bytes_o);
    float* skip = (float*)dzero((size_t)B * Co * L * 4);
    WN(wn_hblock_forward(&s, prec, packed, hx, r, skip, 0, sg, z, flag, st));

    // backward: the gradient domain carries one power-of-two scale chosen by the caller (here 8) as a DEVICE scalar
    const float scale_h = 8.0f, inv_h = 0.125f;
    float *scale = up(std::vector<float>(1, scale_h)), *inv = up(std::vector<float>(1, inv_h));
    void *hdr_ = dzero(bytes_o), *hds = dzero(bytes_o), *da = dzero(bytes_o), *dg = dzero(bytes_o);
    float *dcr = up(cr), *dcs = up(cs);
    WN(wn_hseries_load(prec, dcr, hdr_, B, Co, L, ld, halo, 1.0f, scale, flag, st));
    WN(wn_hseries_load(prec, dcs, hds, B, Co, L, ld, halo, 1.0f, scale, flag, st));
    float* dxd = (float*)dzero((size_t)B * Ci * L * 4);
    WN(wn_hblock_backward_data(&s, prec, packed, hdr_, hds, z, sg, da, dg, nullptr, dxd, inv, flag, st));
    wn_block_params g = {(float*)dzero(wt.size() * 4), (float*)dzero(Co * 4), (float*)dzero(ws.size() * 4), (float*)dzero(Co * 4),
                         (float*)dzero(wr.size() * 4), (float*)dzero(Co * 4), (float*)dzero(wk.size() * 4), (float*)dzero(Co * 4),
                         (float*)dzero(wp.size() * 4), (float*)dzero(Co * 4)};
    const size_t wsb = wn_hblock_wgrad_workspace_bytes(&s, prec);
    void* wspace = dzero(wsb ? wsb : 16);
    WN(wn_hblock_backward_weights(&s, prec, hx, z, da, dg, hdr_, hds, &g, inv, wspace, wsb, st));
    CK(hipStreamSynchronize(st));

    FILE* o = fopen(argv[2], "wb");
    if (!o) { perror("open out"); return 1; }
    const unsigned fl = down<unsigned>(flag, 1)[0];
    int ohdr[4] = {ld, halo, (int)bytes_o, (int)fl};
    fwrite(ohdr, 4, 4, o);
    std::vector<unsigned char> rraw = down<unsigned char>(r, bytes_o);          // r as the raw half series (decoded by the test)
    fwrite(rraw.data(), 1, rraw.size(), o);
    auto wr_f = [&](const float* dptr, size_t n) { std::vector<float> h = down<float>(dptr, n); fwrite(h.data(), 4, n, o); };
    wr_f(skip, (size_t)B * Co * L);
    wr_f(dxd, (size_t)B * Ci * L);
    wr_f(g.w_tanh, wt.size()); wr_f(g.b_tanh, Co); wr_f(g.w_sigmoid, ws.size()); wr_f(g.b_sigmoid, Co);
    wr_f(g.w_res, wr.size()); wr_f(g.b_res, Co); wr_f(g.w_skip, wk.size()); wr_f(g.b_skip, Co);
    wr_f(g.w_proj, wp.size()); wr_f(g.b_proj, Co);
    fclose(o);
    printf("cabi_hblock ok: precision=%d B=%d L=%d Ci=%d Co=%d k=%d d=%d causal=%d ld=%d halo=%d overflow=%u version=%d\n", prec, B, L, Ci, Co,
           k, d, causal, ld, halo, fl, wn_version());
    return 0;
}
