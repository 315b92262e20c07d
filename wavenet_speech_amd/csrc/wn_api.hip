// C ABI of libwavenet_amd.so (declared in include/wavenet_amd.h): shape checking, GEMM planning,
// weight-pack descriptors and kernel launch orchestration.  Host code only; the kernels live in
// wn_gemm.hip / wn_wgrad.hip / wn_pack.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/wavenet_amd.h"
#include "wn_kernels.h"

using namespace wn;

#ifdef WN_STAMPS
namespace wn { extern int g_debug_kc; }   // diagnostic build: which kernel class the next launch belongs to
#endif

namespace {

thread_local std::string g_hip_err;

int hip_fail(hipError_t e, const char* what) {
    g_hip_err = std::string(what) + ": " + hipGetErrorString(e);
    return WN_ERR_HIP;
}
#define WN_HIP(call, what)                                  \
    do {                                                    \
        hipError_t e__ = (call);                            \
        if (e__ != hipSuccess) return hip_fail(e__, what);  \
    } while (0)

}  // namespace
namespace wn {
int hip_fail_shared(hipError_t e, const char* what) { return hip_fail(e, what); }   // for the other host translation units
}
namespace {
inline int rup(int x, int m) { return (x + m - 1) / m * m; }
inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline int tiles32(int c) { return cdiv(c, 32); }
inline int cp8(int c) { return rup(c, 8); }
inline int pick_mt(int ntiles) { return ntiles > 2 ? 4 : (ntiles == 2 ? 2 : 1); }
inline size_t align256(size_t b) { return (b + 255) / 256 * 256; }

// ------------------------------------------------------------------------------------------
// profiling (HIP events on the launch stream)
// ------------------------------------------------------------------------------------------
enum KernelClass {   // the half-precision classes (KC_HLOAD...) are used by wn_half_api.hip, which repeats this order
    KC_PACK = 0, KC_GATE_GEMM, KC_OUT_GEMM, KC_DZ_GEMM, KC_DX_GEMM, KC_WGRAD, KC_WGRAD_REDUCE,
    KC_CONV_FWD, KC_CONV_BWD_DATA, KC_SKIP_GEMM, KC_HLOAD, KC_HGATE, KC_HRES, KC_HDZ, KC_HDX, KC_HSKIP, KC_HWGRAD, KC_EMBED,
    KC_SYNTH, KC_CTC, KC_HFUSED, KC_HCONV_FWD, KC_HCONV_BWD_DATA, KC_HCOL_DZ, KC_HCOL_DX, KC_HCOL_DXDZ, KC_HCOL_SKIP,
    KC_COUNT
};
const char* const kKernelNames[KC_COUNT] = {
    "pack_kernel", "series_gemm_kernel<gate>", "series_gemm_kernel<res>", "series_gemm_kernel<dz,dgate>",
    "series_gemm_kernel<dx>", "wgrad_kernel", "wgrad_reduce_kernel", "series_gemm_kernel<conv_fwd>",
    "series_gemm_kernel<conv_bwd_data>", "series_gemm_kernel<skips_sum>", "hload_kernel", "hgemm_kernel<gate>",
    "hgemm_kernel<res>", "hgemm_kernel<dz,dgate>", "hgemm_kernel<dx>", "hgemm_kernel<skips_sum>", "hwgrad_kernel",
    "embed_kernel", "synth_kernel", "ctc_kernel", "hfused_fwd_kernel",
    "hgemm_kernel<conv_fwd>", "hgemm_kernel<conv_bwd_data>", "hcol_kernel<dz,dgate>", "hcol_kernel<dx>", "hcol2_kernel<dx+dz>", "hcol_kernel<skips_sum>"};

struct ProfRec { int kc; hipEvent_t e0, e1; double flops; };
struct Prof {
    std::mutex mu;
    bool on = false;
    std::vector<ProfRec> pending;
    std::vector<hipEvent_t> pool;
    double ms[KC_COUNT] = {0};
    long long n[KC_COUNT] = {0};
    double flops[KC_COUNT] = {0};
} g_prof;

struct ProfScope {
    bool active = false;
    ProfRec rec{};
    hipStream_t st;
    ProfScope(int kc, double flops, hipStream_t s) : st(s) {
#ifdef WN_STAMPS
        wn::g_debug_kc = kc;
#endif
        std::lock_guard<std::mutex> lk(g_prof.mu);
        if (!g_prof.on) return;
        auto get = [&]() {
            hipEvent_t e;
            if (!g_prof.pool.empty()) { e = g_prof.pool.back(); g_prof.pool.pop_back(); return e; }
            if (hipEventCreate(&e) != hipSuccess) return (hipEvent_t) nullptr;
            return e;
        };
        rec.kc = kc; rec.flops = flops; rec.e0 = get(); rec.e1 = get();
        if (!rec.e0 || !rec.e1) return;
        active = hipEventRecord(rec.e0, st) == hipSuccess;
    }
    ~ProfScope() {
        if (!active) return;
        (void)hipEventRecord(rec.e1, st);
        std::lock_guard<std::mutex> lk(g_prof.mu);
        g_prof.pending.push_back(rec);
    }
};

}  // namespace
namespace wn {
struct ProfScopeShared {   // the same scope for wn_half_api.hip / wn_embed.hip
    void* impl;
    ProfScopeShared(int kc, double flops, hipStream_t st);
    ~ProfScopeShared();
};
ProfScopeShared::ProfScopeShared(int kc, double flops, hipStream_t st) : impl(new ProfScope(kc, flops, st)) {}
ProfScopeShared::~ProfScopeShared() { delete static_cast<ProfScope*>(impl); }
}  // namespace wn
namespace {
// ------------------------------------------------------------------------------------------
// geometry
// ------------------------------------------------------------------------------------------
int tap_offsets(int k, int d, int causal, int* off) {
    const int p = causal ? (k - 1) * d : wn_autopad(k, d);
    for (int j = 0; j < k; ++j) off[j] = j * d - p;
    return 0;
}

int check_layout(int B, int L, int ld, int halo, int max_abs_off) {
    if (B <= 0 || L <= 0) return WN_ERR_BAD_SHAPE;
    if (halo < 0 || (halo & 3) || (ld & 3)) return WN_ERR_BAD_SHAPE;
    if (halo < max_abs_off) return WN_ERR_BAD_SHAPE;
    if (ld < 2 * halo + rup(L, kColTile)) return WN_ERR_BAD_SHAPE;
    // the kernels address one utterance's [channels][ld] plane with 32-bit byte offsets (buffer loads)
    if ((double)WN_MAX_CHANNELS * (double)ld * 4.0 >= 4294967296.0) return WN_ERR_UNSUPPORTED;
    return WN_OK;
}

int check_block(const wn_block_shape* s, int* off) {
    if (!s) return WN_ERR_NULL;
    if (s->in_channels <= 0 || s->out_channels <= 0 || s->skip_rows <= 0 || s->dilation <= 0) return WN_ERR_BAD_SHAPE;
    if (s->kernel_width < 1) return WN_ERR_BAD_SHAPE;
    if (s->kernel_width > WN_MAX_TAPS) return WN_ERR_UNSUPPORTED;
    if (s->in_channels > WN_MAX_CHANNELS || s->out_channels > WN_MAX_CHANNELS || s->skip_rows > WN_MAX_CHANNELS)
        return WN_ERR_UNSUPPORTED;
    tap_offsets(s->kernel_width, s->dilation, s->causal, off);
    int mx = 0;
    for (int j = 0; j < s->kernel_width; ++j) mx = std::max(mx, std::abs(off[j]));
    return check_layout(s->batch, s->length, s->ld, s->halo, mx);
}

// ------------------------------------------------------------------------------------------
// GEMM plans: how M is cut into slabs, where each slab's packed weights / bias live
// ------------------------------------------------------------------------------------------
struct GemmPlan {
    int MT = 1, nslab = 0, nseg = 0;
    int seg_nkb[kMaxSeg] = {0};
    long long slab_woff[kMaxSlab] = {0};
    int slab_nseg[kMaxSlab] = {0}, slab_row0[kMaxSlab] = {0}, slab_dst[kMaxSlab] = {0}, slab_boff[kMaxSlab] = {0};
    long long wfloats = 0;
    int bfloats = 0;
    void add_slab(int nseg_used, int row0, int dst) {
        const int s = nslab++;
        long long kb = 0;
        for (int i = 0; i < nseg_used; ++i) kb += seg_nkb[i];
        slab_woff[s] = wfloats;
        slab_nseg[s] = nseg_used;
        slab_row0[s] = row0;
        slab_dst[s] = dst;
        slab_boff[s] = bfloats;
        wfloats += kb * MT * 256;
        bfloats += MT * 32;
    }
    size_t bytes() const { return align256((size_t)wfloats * 4) + align256((size_t)bfloats * 4); }
};

struct BlockPlan {
    GemmPlan fa, fb, ka, kb;
    int fb_first_skip_slab = 0;
    size_t off_fa = 0, off_fb = 0, off_ka = 0, off_kb = 0, total = 0;
};

BlockPlan plan_block(const wn_block_shape* s) {
    BlockPlan p;
    static const int mt2 = getenv("WN_MT2_MASK") ? atoi(getenv("WN_MT2_MASK")) : 0;   // measurement knob: 1 gate, 2 res/skip, 4 dx
    const int Ci = s->in_channels, Co = s->out_channels, Ms = s->skip_rows, k = s->kernel_width;
    // FA: [a;g] interleaved by 32-channel tile pairs;  K = k taps of x
    {
        GemmPlan& g = p.fa;
        const int pairs = tiles32(Co);
        g.MT = (pairs >= 2 && !(mt2 & 1)) ? 4 : 2;
        g.nseg = k;
        for (int j = 0; j < k; ++j) g.seg_nkb[j] = cp8(Ci) / 8;
        const int ns = cdiv(2 * pairs, g.MT);
        for (int i = 0; i < ns; ++i) g.add_slab(k, i * (g.MT / 2) * 32, 0);
    }
    // FB: r rows contract [z ; x], skip rows contract [z]
    {
        GemmPlan& g = p.fb;
        const int rt = tiles32(Co), st = tiles32(Ms);
        g.MT = pick_mt(std::max(rt, st));
        if ((mt2 & 2) && g.MT == 4) g.MT = 2;
        g.nseg = 2;
        g.seg_nkb[0] = cp8(Co) / 8;
        g.seg_nkb[1] = cp8(Ci) / 8;
        for (int i = 0; i < cdiv(rt, g.MT); ++i) g.add_slab(2, i * g.MT * 32, 0);
        p.fb_first_skip_slab = g.nslab;
        for (int i = 0; i < cdiv(st, g.MT); ++i) g.add_slab(1, i * g.MT * 32, 1);
    }
    // KA: dz rows (z channels) contract [dskip ; dr]
    {
        GemmPlan& g = p.ka;
        const int zt = tiles32(Co);
        g.MT = pick_mt(zt);
        // dz has the shortest K loop (2C) and the heaviest epilogue (reads z, sg, writes da, dg): 64-row slabs at two waves
        // per SIMD let one wave's epilogue overlap the other's MFMAs: 0.616 -> 0.589 ms per launch at 256 ch x 16 x 16000
        // (WN_DZ_MT=4 restores the 128-row slabs for A/B runs)
        static const int dz_mt = getenv("WN_DZ_MT") ? atoi(getenv("WN_DZ_MT")) : 2;
        if (dz_mt == 2 && zt >= 2) g.MT = 2;
        g.nseg = 2;
        g.seg_nkb[0] = cp8(Ms) / 8;
        g.seg_nkb[1] = cp8(Co) / 8;
        for (int i = 0; i < cdiv(zt, g.MT); ++i) g.add_slab(2, i * g.MT * 32, 0);
    }
    // KB: dx rows (input channels) contract [da_0; dg_0; ... da_{k-1}; dg_{k-1}; dr]
    {
        GemmPlan& g = p.kb;
        const int xt = tiles32(Ci);
        g.MT = pick_mt(xt);
        if ((mt2 & 4) && g.MT == 4) g.MT = 2;
        g.nseg = 2 * k + 1;
        for (int j = 0; j < 2 * k + 1; ++j) g.seg_nkb[j] = cp8(Co) / 8;
        for (int i = 0; i < cdiv(xt, g.MT); ++i) g.add_slab(2 * k + 1, i * g.MT * 32, 0);
    }
    p.off_fa = 0;
    p.off_fb = p.off_fa + p.fa.bytes();
    p.off_ka = p.off_fb + p.fb.bytes();
    p.off_kb = p.off_ka + p.ka.bytes();
    p.total = p.off_kb + p.kb.bytes();
    return p;
}

inline const float* plan_w(const void* packed, size_t off) { return reinterpret_cast<const float*>((const char*)packed + off); }
inline const float* plan_b(const void* packed, size_t off, const GemmPlan& g) {
    return reinterpret_cast<const float*>((const char*)packed + off + align256((size_t)g.wfloats * 4));
}

void fill_pack_common(PackArgs& a, const GemmPlan& g, void* packed, size_t off) {
    std::memset(&a, 0, sizeof(a));
    a.nslab = g.nslab;
    a.MT = g.MT;
    for (int i = 0; i < kMaxSeg; ++i) a.seg_nkb[i] = g.seg_nkb[i];
    for (int i = 0; i < g.nslab; ++i) {
        a.slab_woff[i] = g.slab_woff[i];
        a.slab_nseg[i] = g.slab_nseg[i];
        a.slab_boff[i] = g.slab_boff[i];
    }
    a.wpacked = reinterpret_cast<float*>((char*)packed + off);
    a.bias = reinterpret_cast<float*>((char*)packed + off + align256((size_t)g.wfloats * 4));
    a.total = g.wfloats;
}

void fill_gemm_common(GemmArgs& a, const GemmPlan& g, const void* packed, size_t off, int first_slab, int nslab,
                      int B, int L, int ld, int halo) {
    std::memset(&a, 0, sizeof(a));
    a.wpacked = plan_w(packed, off);
    a.bias = plan_b(packed, off, g);
    a.nslab = nslab;
    for (int i = 0; i < nslab; ++i) {
        GemmSlab& s = a.slab[i];
        s.woff = g.slab_woff[first_slab + i];
        s.nseg = g.slab_nseg[first_slab + i];
        s.dst = g.slab_dst[first_slab + i];
        s.row0 = g.slab_row0[first_slab + i];
        s.boff = g.slab_boff[first_slab + i];
    }
    a.B = B; a.L = L; a.ld = ld; a.halo = halo;
    a.tiles_per_row = cdiv(L, kColTile);
    a.ncol = B * a.tiles_per_row;
}

inline void set_seg(GemmArgs& a, int i, const float* base, int c, int off, int nkb) {
    a.seg[i].base = base; a.seg[i].cp = cp8(c); a.seg[i].off = off; a.seg[i].nkb = nkb;
}

inline PackSrc mk_src(const float* p, int rows, int cols, int sr, int sc) {
    PackSrc s; s.ptr = p; s.rows = rows; s.cols = cols; s.stride_r = sr; s.stride_c = sc; return s;
}

// ------------------------------------------------------------------------------------------
// weight-gradient planning
// ------------------------------------------------------------------------------------------
struct WgradPlan {
    int WT = 1, npair = 0, ntile_total = 0, nsplit = 1;
    int Mp[kMaxPair], Np[kMaxPair], mt[kMaxPair], nt[kMaxPair], tile0[kMaxPair], rs_off[kMaxPair];
    long long slab_off[kMaxPair];
    long long slab_floats = 0;
    int rs_floats = 0;
    void add(int M, int N) {
        const int T = 64 * WT, i = npair++;
        mt[i] = cdiv(M, T); nt[i] = cdiv(N, T);
        Mp[i] = mt[i] * T; Np[i] = nt[i] * T;
        tile0[i] = ntile_total; ntile_total += mt[i] * nt[i];
        slab_off[i] = slab_floats; slab_floats += (long long)Mp[i] * Np[i];
        rs_off[i] = rs_floats; rs_floats += Mp[i];
    }
    void finish(int nchunk) {
        nsplit = std::max(1, 512 / std::max(1, ntile_total));
        nsplit = std::min(nsplit, std::max(1, nchunk));
        if (nsplit >= 16) nsplit = nsplit / 8 * 8;   // multiple of 8: all tiles of a split go to one XCD (wgrad_kernel)
    }
    bool xcd_map() const { return nsplit % 8 == 0; }
    size_t bytes() const { return (size_t)nsplit * (size_t)(slab_floats + rs_floats) * 4; }
};
inline int pick_wt(int maxdim) { return maxdim > 128 ? 4 : (maxdim > 64 ? 2 : 1); }

}  // namespace

// ==========================================================================================
// C ABI
// ==========================================================================================
// (every function below was declared extern "C" by include/wavenet_amd.h and keeps that linkage)

int wn_version(void) { return WN_VERSION; }

const char* wn_strerror(int status) {
    switch (status) {
        case WN_OK: return "ok";
        case WN_ERR_BAD_SHAPE: return "bad shape or series layout (dimension <= 0, ld/halo not multiples of 4, halo smaller than the taps reach, or ld too small)";
        case WN_ERR_UNSUPPORTED: return "unsupported configuration (kernel_width > WN_MAX_TAPS or channels > WN_MAX_CHANNELS)";
        case WN_ERR_NULL: return "required pointer is NULL";
        case WN_ERR_HIP: return "HIP runtime / kernel launch error (see wn_last_hip_error)";
        case WN_ERR_WORKSPACE: return "workspace too small";
    }
    return "unknown wn_status";
}

const char* wn_last_hip_error(void) { return g_hip_err.c_str(); }

int wn_round_up(int x, int multiple) { return multiple > 0 ? rup(x, multiple) : x; }

int wn_autopad(int k, int d) {
    const int total = (k - 1) * d;
    return (total % 2 == 1) ? (total - 1) / 2 + 1 : total / 2;
}

int wn_tap_offsets(int k, int d, int causal, int* off) {
    if (!off) return WN_ERR_NULL;
    if (k < 1 || d < 1) return WN_ERR_BAD_SHAPE;
    return tap_offsets(k, d, causal, off);
}

int wn_series_layout(int length, int max_abs_offset, int* ld, int* halo) {
    if (!ld || !halo) return WN_ERR_NULL;
    if (length <= 0 || max_abs_offset < 0) return WN_ERR_BAD_SHAPE;
    *halo = rup(max_abs_offset, 4);
    *ld = 2 * (*halo) + rup(length, kColTile);
    return WN_OK;
}

size_t wn_series_floats(int batch, int channels, int ld) { return (size_t)batch * cp8(channels) * (size_t)ld; }

// ------------------------------------------------------------------------------------------
// residual block
// ------------------------------------------------------------------------------------------
size_t wn_block_packed_bytes(const wn_block_shape* s) {
    int off[WN_MAX_TAPS];
    if (check_block(s, off) != WN_OK) return 0;
    return plan_block(s).total;
}

int wn_block_pack(const wn_block_shape* s, const wn_block_params* p, void* packed, wn_stream_t stream) {
    int off[WN_MAX_TAPS];
    int rc = check_block(s, off);
    if (rc != WN_OK) return rc;
    if (!p || !packed || !p->w_tanh || !p->w_sigmoid || !p->w_res || !p->w_skip || !p->w_proj) return WN_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    const BlockPlan bp = plan_block(s);
    const int Ci = s->in_channels, Co = s->out_channels, Ms = s->skip_rows, k = s->kernel_width;
    ProfScope prof(KC_PACK, 0.0, st);
    PackArgs a;
    {   // FA
        const GemmPlan& g = bp.fa;
        fill_pack_common(a, g, packed, bp.off_fa);
        for (int j = 0; j < k; ++j) {
            a.set[0].seg[j] = mk_src(p->w_tanh + j, Co, Ci, Ci * k, k);
            a.set[1].seg[j] = mk_src(p->w_sigmoid + j, Co, Ci, Ci * k, k);
        }
        a.set[0].bias0 = p->b_tanh; a.set[0].bias_rows = Co;
        a.set[1].bias0 = p->b_sigmoid; a.set[1].bias_rows = Co;
        const int pairs = tiles32(Co);
        for (int sl = 0; sl < g.nslab; ++sl)
            for (int m = 0; m < g.MT; ++m) {
                const int pair = sl * (g.MT / 2) + m / 2;
                a.tile[sl * g.MT + m].set = m & 1;
                a.tile[sl * g.MT + m].row0 = pair < pairs ? 32 * pair : -1;
            }
        WN_HIP(launch_pack(a, st), "pack(gate)");
    }
    {   // FB
        const GemmPlan& g = bp.fb;
        fill_pack_common(a, g, packed, bp.off_fb);
        a.set[0].seg[0] = mk_src(p->w_res, Co, Co, Co, 1);
        a.set[0].seg[1] = mk_src(p->w_proj, Co, Ci, Ci, 1);
        a.set[0].bias0 = p->b_res; a.set[0].bias1 = p->b_proj; a.set[0].bias_rows = Co;
        a.set[1].seg[0] = mk_src(p->w_skip, Ms, Co, Co, 1);
        a.set[1].bias0 = p->b_skip; a.set[1].bias_rows = Ms;
        for (int sl = 0; sl < g.nslab; ++sl)
            for (int m = 0; m < g.MT; ++m) {
                const bool skip = sl >= bp.fb_first_skip_slab;
                const int row0 = g.slab_row0[sl] + 32 * m;
                a.tile[sl * g.MT + m].set = skip ? 1 : 0;
                a.tile[sl * g.MT + m].row0 = row0 < (skip ? Ms : Co) ? row0 : -1;
            }
        WN_HIP(launch_pack(a, st), "pack(res+skip)");
    }
    {   // KA: rows = z channel c ; seg0 cols = skip row m : w_skip[m][c] ; seg1 cols = r row m : w_res[m][c]
        const GemmPlan& g = bp.ka;
        fill_pack_common(a, g, packed, bp.off_ka);
        a.set[0].seg[0] = mk_src(p->w_skip, Co, Ms, 1, Co);
        a.set[0].seg[1] = mk_src(p->w_res, Co, Co, 1, Co);
        for (int sl = 0; sl < g.nslab; ++sl)
            for (int m = 0; m < g.MT; ++m) {
                const int row0 = g.slab_row0[sl] + 32 * m;
                a.tile[sl * g.MT + m].set = 0;
                a.tile[sl * g.MT + m].row0 = row0 < Co ? row0 : -1;
            }
        WN_HIP(launch_pack(a, st), "pack(dz)");
    }
    {   // KB: rows = input channel ci ; cols = output channel co
        const GemmPlan& g = bp.kb;
        fill_pack_common(a, g, packed, bp.off_kb);
        for (int j = 0; j < k; ++j) {
            a.set[0].seg[2 * j] = mk_src(p->w_tanh + j, Ci, Co, k, Ci * k);
            a.set[0].seg[2 * j + 1] = mk_src(p->w_sigmoid + j, Ci, Co, k, Ci * k);
        }
        a.set[0].seg[2 * k] = mk_src(p->w_proj, Ci, Co, 1, Ci);
        for (int sl = 0; sl < g.nslab; ++sl)
            for (int m = 0; m < g.MT; ++m) {
                const int row0 = g.slab_row0[sl] + 32 * m;
                a.tile[sl * g.MT + m].set = 0;
                a.tile[sl * g.MT + m].row0 = row0 < Ci ? row0 : -1;
            }
        WN_HIP(launch_pack(a, st), "pack(dx)");
    }
    return WN_OK;
}

int wn_block_forward(const wn_block_shape* s, const void* packed, const float* x, float* r_out, float* skip,
                     int skip_accumulate, float* sg, float* z, wn_stream_t stream) {
    int off[WN_MAX_TAPS];
    int rc = check_block(s, off);
    if (rc != WN_OK) return rc;
    if (!packed || !x || !z) return WN_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    const BlockPlan bp = plan_block(s);
    const int Ci = s->in_channels, Co = s->out_channels, Ms = s->skip_rows, k = s->kernel_width;
    const double BL = (double)s->batch * s->length;
    GemmArgs a;
    {   // a,g = dilated convs ; sg, z
        const GemmPlan& g = bp.fa;
        fill_gemm_common(a, g, packed, bp.off_fa, 0, g.nslab, s->batch, s->length, s->ld, s->halo);
        for (int j = 0; j < k; ++j) set_seg(a, j, x, Ci, off[j], g.seg_nkb[j]);
        a.sg = sg; a.z = z; a.gate_cp = cp8(Co); a.gate_rows = Co;
        ProfScope prof(KC_GATE_GEMM, 2.0 * (2.0 * Co) * (double)(k * Ci) * BL, st);
        WN_HIP(launch_gemm(g.MT, EPI_GATE, a, st), "series_gemm<gate>");
    }
    // r = W_res z + W_proj x + b   and   skip (+)= W_skip z + b : two launches of the same packed GEMM, so that the
    // accumulating skip slabs get their own lean instantiation (EPI_ACCUM) and waves of equal length run together
    {
        const GemmPlan& g = bp.fb;
        const double fl_r = 2.0 * Co * (double)(Co + Ci) * BL, fl_s = 2.0 * Ms * (double)Co * BL;
        if (r_out) {
            fill_gemm_common(a, g, packed, bp.off_fb, 0, bp.fb_first_skip_slab, s->batch, s->length, s->ld, s->halo);
            set_seg(a, 0, z, Co, 0, g.seg_nkb[0]);
            set_seg(a, 1, x, Ci, 0, g.seg_nkb[1]);
            a.dst[0].base = r_out; a.dst[0].cp = cp8(Co); a.dst[0].rows = Co; a.dst[0].accumulate = 0;
            ProfScope prof(KC_OUT_GEMM, fl_r, st);
            WN_HIP(launch_gemm(g.MT, EPI_LINEAR, a, st), "series_gemm<res>");
        }
        if (skip) {
            fill_gemm_common(a, g, packed, bp.off_fb, bp.fb_first_skip_slab, g.nslab - bp.fb_first_skip_slab, s->batch,
                             s->length, s->ld, s->halo);
            set_seg(a, 0, z, Co, 0, g.seg_nkb[0]);
            a.dst[1].base = skip; a.dst[1].cp = cp8(Ms); a.dst[1].rows = Ms; a.dst[1].accumulate = skip_accumulate ? 1 : 0;
            ProfScope prof(KC_SKIP_GEMM, fl_s, st);
            WN_HIP(launch_gemm(g.MT, skip_accumulate ? EPI_ACCUM : EPI_LINEAR, a, st), "series_gemm<skip>");
        }
    }
    return WN_OK;
}

int wn_block_backward_data(const wn_block_shape* s, const void* packed, const float* dr, const float* dskip,
                           const float* z, const float* sg, float* da, float* dg, float* dx, wn_stream_t stream) {
    int off[WN_MAX_TAPS];
    int rc = check_block(s, off);
    if (rc != WN_OK) return rc;
    if (!packed || !dskip || !z || !sg || !da || !dg) return WN_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    const BlockPlan bp = plan_block(s);
    const int Ci = s->in_channels, Co = s->out_channels, Ms = s->skip_rows, k = s->kernel_width;
    const double BL = (double)s->batch * s->length;
    GemmArgs a;
    {   // dz = W_skip^T dskip + W_res^T dr ; da, dg
        const GemmPlan& g = bp.ka;
        fill_gemm_common(a, g, packed, bp.off_ka, 0, g.nslab, s->batch, s->length, s->ld, s->halo);
        a.bias = nullptr;
        set_seg(a, 0, dskip, Ms, 0, g.seg_nkb[0]);
        set_seg(a, 1, dr, Co, 0, g.seg_nkb[1]);
        const int nseg = dr ? 2 : 1;
        for (int i = 0; i < a.nslab; ++i) a.slab[i].nseg = nseg;
        a.z = const_cast<float*>(z); a.sg = const_cast<float*>(sg); a.da = da; a.dg = dg;
        a.gate_cp = cp8(Co); a.gate_rows = Co;
        ProfScope prof(KC_DZ_GEMM, 2.0 * Co * (double)(Ms + (dr ? Co : 0)) * BL, st);
        WN_HIP(launch_gemm(g.MT, EPI_DGATE, a, st), "series_gemm<dz>");
    }
    if (dx) {  // dx[t] = sum_j (W_tanh_j^T da + W_sigmoid_j^T dg)[t - off_j] + W_proj^T dr[t]
        const GemmPlan& g = bp.kb;
        fill_gemm_common(a, g, packed, bp.off_kb, 0, g.nslab, s->batch, s->length, s->ld, s->halo);
        a.bias = nullptr;
        for (int j = 0; j < k; ++j) {
            set_seg(a, 2 * j, da, Co, -off[j], g.seg_nkb[2 * j]);
            set_seg(a, 2 * j + 1, dg, Co, -off[j], g.seg_nkb[2 * j + 1]);
        }
        set_seg(a, 2 * k, dr, Co, 0, g.seg_nkb[2 * k]);
        const int nseg = 2 * k + (dr ? 1 : 0);
        for (int i = 0; i < a.nslab; ++i) a.slab[i].nseg = nseg;
        a.dst[0].base = dx; a.dst[0].cp = cp8(Ci); a.dst[0].rows = Ci; a.dst[0].accumulate = 0;
        ProfScope prof(KC_DX_GEMM, 2.0 * Ci * (double)(2 * k * Co + (dr ? Co : 0)) * BL, st);
        WN_HIP(launch_gemm(g.MT, EPI_LINEAR, a, st), "series_gemm<dx>");
    }
    return WN_OK;
}

// ---- skips_sum of a whole stack ---------------------------------------------------------------------
namespace {
int check_skipsum(const wn_skipsum_shape* s) {
    if (!s) return WN_ERR_NULL;
    if (s->nblocks < 1 || s->skip_rows <= 0) return WN_ERR_BAD_SHAPE;
    if (s->nblocks > WN_MAX_STACK_GROUP || s->skip_rows > WN_MAX_CHANNELS) return WN_ERR_UNSUPPORTED;
    for (int l = 0; l < s->nblocks; ++l) {
        if (s->channels[l] <= 0) return WN_ERR_BAD_SHAPE;
        if (s->channels[l] > WN_MAX_CHANNELS) return WN_ERR_UNSUPPORTED;
    }
    return check_layout(s->batch, s->length, s->ld, s->halo, 0);
}
GemmPlan plan_skipsum(const wn_skipsum_shape* s) {
    GemmPlan g;
    const int t = tiles32(s->skip_rows);
    g.MT = pick_mt(t);
    g.nseg = s->nblocks;
    for (int l = 0; l < s->nblocks; ++l) g.seg_nkb[l] = cp8(s->channels[l]) / 8;
    for (int i = 0; i < cdiv(t, g.MT); ++i) g.add_slab(s->nblocks, i * g.MT * 32, 0);
    return g;
}
}  // namespace

size_t wn_skipsum_packed_bytes(const wn_skipsum_shape* s) {
    if (check_skipsum(s) != WN_OK) return 0;
    return plan_skipsum(s).bytes();
}

int wn_skipsum_pack(const wn_skipsum_shape* s, const float* const* w_skip, const float* bias_total, void* packed,
                    wn_stream_t stream) {
    int rc = check_skipsum(s);
    if (rc != WN_OK) return rc;
    if (!w_skip || !packed) return WN_ERR_NULL;
    for (int l = 0; l < s->nblocks; ++l) if (!w_skip[l]) return WN_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    const GemmPlan g = plan_skipsum(s);
    ProfScope prof(KC_PACK, 0.0, st);
    PackArgs a;
    fill_pack_common(a, g, packed, 0);
    for (int l = 0; l < s->nblocks; ++l) a.set[0].seg[l] = mk_src(w_skip[l], s->skip_rows, s->channels[l], s->channels[l], 1);
    a.set[0].bias0 = bias_total; a.set[0].bias_rows = s->skip_rows;
    for (int sl = 0; sl < g.nslab; ++sl)
        for (int m = 0; m < g.MT; ++m) {
            const int row0 = g.slab_row0[sl] + 32 * m;
            a.tile[sl * g.MT + m].set = 0;
            a.tile[sl * g.MT + m].row0 = row0 < s->skip_rows ? row0 : -1;
        }
    WN_HIP(launch_pack(a, st), "pack(skipsum)");
    return WN_OK;
}

int wn_skipsum_forward(const wn_skipsum_shape* s, const void* packed, const float* const* z, float* skip, int accumulate,
                       wn_stream_t stream) {
    int rc = check_skipsum(s);
    if (rc != WN_OK) return rc;
    if (!packed || !z || !skip) return WN_ERR_NULL;
    for (int l = 0; l < s->nblocks; ++l) if (!z[l]) return WN_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    const GemmPlan g = plan_skipsum(s);
    GemmArgs a;
    fill_gemm_common(a, g, packed, 0, 0, g.nslab, s->batch, s->length, s->ld, s->halo);
    double ksum = 0;
    for (int l = 0; l < s->nblocks; ++l) { set_seg(a, l, z[l], s->channels[l], 0, g.seg_nkb[l]); ksum += s->channels[l]; }
    a.dst[0].base = skip; a.dst[0].cp = cp8(s->skip_rows); a.dst[0].rows = s->skip_rows; a.dst[0].accumulate = accumulate ? 1 : 0;
    ProfScope prof(KC_SKIP_GEMM, 2.0 * s->skip_rows * ksum * (double)s->batch * s->length, st);
    WN_HIP(launch_gemm(g.MT, accumulate ? EPI_ACCUM : EPI_LINEAR, a, st), "series_gemm<skipsum>");
    return WN_OK;
}

// ---- weight gradients --------------------------------------------------------------------------
namespace {
struct PairSpec { const float* A; int a_rows; const float* Bm; int b_rows; int off; int rowsum;
                  float* w; int sm, sn; float* b0; float* b1; };

int run_wgrad(const std::vector<PairSpec>& ps, int maxdim, int B, int L, int ld, int halo, void* workspace,
              size_t workspace_bytes, bool dry, size_t* need, int kc, hipStream_t st) {
    WgradPlan wp;
    wp.WT = pick_wt(maxdim);
    for (const PairSpec& p : ps) wp.add(p.a_rows, p.b_rows);
    const int cpr = cdiv(L, 32);
    wp.finish(B * cpr);
    if (need) *need = wp.bytes();
    if (dry) return WN_OK;
    // wgrad_kernel addresses a chunk as a wave-uniform base + a 32-bit per-thread byte offset (row * ld + column) * 4
    for (const PairSpec& p : ps)
        if ((double)cp8(std::max(p.a_rows, p.b_rows)) * (double)ld * 4.0 >= 4294967296.0) return WN_ERR_UNSUPPORTED;
    if (ps.empty()) return WN_OK;
    if (!workspace) return WN_ERR_NULL;
    if (workspace_bytes < wp.bytes() || (reinterpret_cast<uintptr_t>(workspace) & 15)) return WN_ERR_WORKSPACE;   // 16-byte aligned
    WgradArgs a;
    std::memset(&a, 0, sizeof(a));
    ReduceArgs r;
    std::memset(&r, 0, sizeof(r));
    double flops = 0;
    for (int i = 0; i < wp.npair; ++i) {
        WgradPair& q = a.pair[i];
        q.A = ps[i].A; q.Bm = ps[i].Bm; q.a_cp = cp8(ps[i].a_rows); q.b_cp = cp8(ps[i].b_rows);
        q.off = ps[i].off; q.mt = wp.mt[i]; q.nt = wp.nt[i]; q.tile0 = wp.tile0[i];
        q.slab_off = wp.slab_off[i]; q.Mp = wp.Mp[i]; q.Np = wp.Np[i];
        q.rowsum = ps[i].rowsum; q.rs_off = wp.rs_off[i];
        ReduceDst& d = r.d[i];
        d.w = ps[i].w; d.M = ps[i].a_rows; d.N = ps[i].b_rows; d.sm = ps[i].sm; d.sn = ps[i].sn;
        d.slab_off = wp.slab_off[i]; d.Np = wp.Np[i];
        d.b0 = ps[i].rowsum ? ps[i].b0 : nullptr; d.b1 = ps[i].rowsum ? ps[i].b1 : nullptr; d.rs_off = wp.rs_off[i];
        d.post = 1.0f;
        flops += 2.0 * ps[i].a_rows * (double)ps[i].b_rows * (double)B * L;
    }
    a.npair = wp.npair; a.ntile_total = wp.ntile_total; a.nsplit = wp.nsplit; a.xcd_map = wp.xcd_map() ? 1 : 0;
    a.B = B; a.L = L; a.ld = ld; a.halo = halo; a.chunks_per_row = cpr; a.nchunk = B * cpr;
    a.slab = reinterpret_cast<float*>(workspace);
    a.rowsum = a.slab + (size_t)wp.nsplit * wp.slab_floats;
    a.slab_floats = wp.slab_floats; a.rs_floats = wp.rs_floats;
    r.npair = wp.npair; r.nsplit = wp.nsplit; r.slab = a.slab; r.rowsum = a.rowsum;
    r.slab_floats = wp.slab_floats; r.rs_floats = wp.rs_floats;
    {
        ProfScope prof(kc, flops, st);
        WN_HIP(launch_wgrad(wp.WT, a, st), "wgrad");
    }
    {
        ProfScope prof(KC_WGRAD_REDUCE, 0.0, st);
        WN_HIP(launch_wgrad_reduce(r, st), "wgrad_reduce");
    }
    return WN_OK;
}

std::vector<PairSpec> block_pairs(const wn_block_shape* s, const int* off, const float* x, const float* z,
                                  const float* da, const float* dg, const float* dr, const float* dskip,
                                  const wn_block_params* g) {
    const int Ci = s->in_channels, Co = s->out_channels, Ms = s->skip_rows, k = s->kernel_width;
    std::vector<PairSpec> ps;
    for (int j = 0; j < k; ++j) {
        ps.push_back({da, Co, x, Ci, off[j], j == 0, g ? g->w_tanh + j : nullptr, Ci * k, k, g ? g->b_tanh : nullptr, nullptr});
        ps.push_back({dg, Co, x, Ci, off[j], j == 0, g ? g->w_sigmoid + j : nullptr, Ci * k, k, g ? g->b_sigmoid : nullptr, nullptr});
    }
    ps.push_back({dskip, Ms, z, Co, 0, 1, g ? g->w_skip : nullptr, Co, 1, g ? g->b_skip : nullptr, nullptr});
    if (dr) {
        ps.push_back({dr, Co, z, Co, 0, 1, g ? g->w_res : nullptr, Co, 1, g ? g->b_res : nullptr, g ? g->b_proj : nullptr});
        ps.push_back({dr, Co, x, Ci, 0, 0, g ? g->w_proj : nullptr, Ci, 1, nullptr, nullptr});
    }
    return ps;
}
}  // namespace

size_t wn_block_wgrad_workspace_bytes(const wn_block_shape* s) {
    int off[WN_MAX_TAPS];
    if (check_block(s, off) != WN_OK) return 0;
    static const float dummy = 0;
    // the split-K plan depends on the number of pairs, which depends on whether dr is given: cover both
    size_t need = 0;
    for (int with_dr = 0; with_dr < 2; ++with_dr) {
        std::vector<PairSpec> ps = block_pairs(s, off, &dummy, &dummy, &dummy, &dummy, with_dr ? &dummy : nullptr, &dummy, nullptr);
        size_t n = 0;
        run_wgrad(ps, std::max(std::max(s->in_channels, s->out_channels), s->skip_rows), s->batch, s->length, s->ld,
                  s->halo, nullptr, 0, true, &n, KC_WGRAD, nullptr);
        need = std::max(need, n);
    }
    return need;
}

int wn_block_backward_weights(const wn_block_shape* s, const float* x, const float* z, const float* da, const float* dg,
                              const float* dr, const float* dskip, const wn_block_params* grads, void* workspace,
                              size_t workspace_bytes, wn_stream_t stream) {
    int off[WN_MAX_TAPS];
    int rc = check_block(s, off);
    if (rc != WN_OK) return rc;
    if (!x || !z || !da || !dg || !dskip || !grads) return WN_ERR_NULL;
    if (!grads->w_tanh || !grads->b_tanh || !grads->w_sigmoid || !grads->b_sigmoid || !grads->w_skip || !grads->b_skip)
        return WN_ERR_NULL;
    // the residual path's four gradients are required only when the block HAS a residual consumer (dr != NULL);
    // with dr == NULL they may be NULL ("no gradient", what autograd gives the reference's last block) or buffers to zero
    if (dr && (!grads->w_res || !grads->b_res || !grads->w_proj || !grads->b_proj)) return WN_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    const int Ci = s->in_channels, Co = s->out_channels;
    if (!dr) {
        if (grads->w_res) WN_HIP(hipMemsetAsync(grads->w_res, 0, (size_t)Co * Co * 4, st), "memset dW_res");
        if (grads->b_res) WN_HIP(hipMemsetAsync(grads->b_res, 0, (size_t)Co * 4, st), "memset db_res");
        if (grads->w_proj) WN_HIP(hipMemsetAsync(grads->w_proj, 0, (size_t)Co * Ci * 4, st), "memset dW_proj");
        if (grads->b_proj) WN_HIP(hipMemsetAsync(grads->b_proj, 0, (size_t)Co * 4, st), "memset db_proj");
    }
    std::vector<PairSpec> ps = block_pairs(s, off, x, z, da, dg, dr, dskip, grads);
    return run_wgrad(ps, std::max(std::max(Ci, Co), s->skip_rows), s->batch, s->length, s->ld, s->halo, workspace,
                     workspace_bytes, false, nullptr, KC_WGRAD, st);
}

// ------------------------------------------------------------------------------------------
// stand-alone dilated conv
// ------------------------------------------------------------------------------------------
namespace {
int check_conv(const wn_conv_shape* s, int* off) {
    if (!s) return WN_ERR_NULL;
    if (s->in_channels <= 0 || s->out_channels <= 0 || s->dilation <= 0 || s->kernel_width < 1) return WN_ERR_BAD_SHAPE;
    if (s->kernel_width > WN_MAX_TAPS) return WN_ERR_UNSUPPORTED;
    if (s->in_channels > WN_MAX_CHANNELS || s->out_channels > WN_MAX_CHANNELS) return WN_ERR_UNSUPPORTED;
    tap_offsets(s->kernel_width, s->dilation, s->causal, off);
    int mx = 0;
    for (int j = 0; j < s->kernel_width; ++j) mx = std::max(mx, std::abs(off[j]));
    return check_layout(s->batch, s->length, s->ld, s->halo, mx);
}
struct ConvPlan { GemmPlan cf, cb; size_t off_cf = 0, off_cb = 0, total = 0; };
ConvPlan plan_conv(const wn_conv_shape* s) {
    ConvPlan p;
    const int Ci = s->in_channels, Co = s->out_channels, k = s->kernel_width;
    {
        GemmPlan& g = p.cf;
        const int t = tiles32(Co);
        g.MT = pick_mt(t); g.nseg = k;
        for (int j = 0; j < k; ++j) g.seg_nkb[j] = cp8(Ci) / 8;
        for (int i = 0; i < cdiv(t, g.MT); ++i) g.add_slab(k, i * g.MT * 32, 0);
    }
    {
        GemmPlan& g = p.cb;
        const int t = tiles32(Ci);
        g.MT = pick_mt(t); g.nseg = k;
        for (int j = 0; j < k; ++j) g.seg_nkb[j] = cp8(Co) / 8;
        for (int i = 0; i < cdiv(t, g.MT); ++i) g.add_slab(k, i * g.MT * 32, 0);
    }
    p.off_cf = 0; p.off_cb = p.cf.bytes(); p.total = p.off_cb + p.cb.bytes();
    return p;
}
}  // namespace

size_t wn_conv_packed_bytes(const wn_conv_shape* s) {
    int off[WN_MAX_TAPS];
    if (check_conv(s, off) != WN_OK) return 0;
    return plan_conv(s).total;
}

int wn_conv_pack(const wn_conv_shape* s, const float* weight, const float* bias, void* packed, wn_stream_t stream) {
    int off[WN_MAX_TAPS];
    int rc = check_conv(s, off);
    if (rc != WN_OK) return rc;
    if (!weight || !packed) return WN_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    const ConvPlan cp = plan_conv(s);
    const int Ci = s->in_channels, Co = s->out_channels, k = s->kernel_width;
    ProfScope prof(KC_PACK, 0.0, st);
    PackArgs a;
    {
        const GemmPlan& g = cp.cf;
        fill_pack_common(a, g, packed, cp.off_cf);
        for (int j = 0; j < k; ++j) a.set[0].seg[j] = mk_src(weight + j, Co, Ci, Ci * k, k);
        a.set[0].bias0 = bias; a.set[0].bias_rows = Co;
        for (int sl = 0; sl < g.nslab; ++sl)
            for (int m = 0; m < g.MT; ++m) {
                const int row0 = g.slab_row0[sl] + 32 * m;
                a.tile[sl * g.MT + m].set = 0;
                a.tile[sl * g.MT + m].row0 = row0 < Co ? row0 : -1;
            }
        WN_HIP(launch_pack(a, st), "pack(conv fwd)");
    }
    {
        const GemmPlan& g = cp.cb;
        fill_pack_common(a, g, packed, cp.off_cb);
        for (int j = 0; j < k; ++j) a.set[0].seg[j] = mk_src(weight + j, Ci, Co, k, Ci * k);
        for (int sl = 0; sl < g.nslab; ++sl)
            for (int m = 0; m < g.MT; ++m) {
                const int row0 = g.slab_row0[sl] + 32 * m;
                a.tile[sl * g.MT + m].set = 0;
                a.tile[sl * g.MT + m].row0 = row0 < Ci ? row0 : -1;
            }
        WN_HIP(launch_pack(a, st), "pack(conv bwd)");
    }
    return WN_OK;
}

int wn_conv_forward(const wn_conv_shape* s, const void* packed, const float* x, float* y, wn_stream_t stream) {
    int off[WN_MAX_TAPS];
    int rc = check_conv(s, off);
    if (rc != WN_OK) return rc;
    if (!packed || !x || !y) return WN_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    const ConvPlan cp = plan_conv(s);
    const GemmPlan& g = cp.cf;
    GemmArgs a;
    fill_gemm_common(a, g, packed, cp.off_cf, 0, g.nslab, s->batch, s->length, s->ld, s->halo);
    for (int j = 0; j < s->kernel_width; ++j) set_seg(a, j, x, s->in_channels, off[j], g.seg_nkb[j]);
    a.dst[0].base = y; a.dst[0].cp = cp8(s->out_channels); a.dst[0].rows = s->out_channels; a.dst[0].accumulate = 0;
    ProfScope prof(KC_CONV_FWD, 2.0 * s->out_channels * (double)(s->kernel_width * s->in_channels) * s->batch * s->length, st);
    WN_HIP(launch_gemm(g.MT, EPI_LINEAR, a, st), "series_gemm<conv fwd>");
    return WN_OK;
}

int wn_conv_backward_data(const wn_conv_shape* s, const void* packed, const float* dy, float* dx, wn_stream_t stream) {
    int off[WN_MAX_TAPS];
    int rc = check_conv(s, off);
    if (rc != WN_OK) return rc;
    if (!packed || !dy || !dx) return WN_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    const ConvPlan cp = plan_conv(s);
    const GemmPlan& g = cp.cb;
    GemmArgs a;
    fill_gemm_common(a, g, packed, cp.off_cb, 0, g.nslab, s->batch, s->length, s->ld, s->halo);
    a.bias = nullptr;
    for (int j = 0; j < s->kernel_width; ++j) set_seg(a, j, dy, s->out_channels, -off[j], g.seg_nkb[j]);
    a.dst[0].base = dx; a.dst[0].cp = cp8(s->in_channels); a.dst[0].rows = s->in_channels; a.dst[0].accumulate = 0;
    ProfScope prof(KC_CONV_BWD_DATA, 2.0 * s->in_channels * (double)(s->kernel_width * s->out_channels) * s->batch * s->length, st);
    WN_HIP(launch_gemm(g.MT, EPI_LINEAR, a, st), "series_gemm<conv bwd>");
    return WN_OK;
}

namespace {
std::vector<PairSpec> conv_pairs(const wn_conv_shape* s, const int* off, const float* x, const float* dy, float* dw, float* db) {
    std::vector<PairSpec> ps;
    const int Ci = s->in_channels, Co = s->out_channels, k = s->kernel_width;
    for (int j = 0; j < k; ++j)
        ps.push_back({dy, Co, x, Ci, off[j], (j == 0 && db) ? 1 : 0, dw ? dw + j : nullptr, Ci * k, k, db, nullptr});
    return ps;
}
}  // namespace

size_t wn_conv_wgrad_workspace_bytes(const wn_conv_shape* s) {
    int off[WN_MAX_TAPS];
    if (check_conv(s, off) != WN_OK) return 0;
    static const float dummy = 0;
    std::vector<PairSpec> ps = conv_pairs(s, off, &dummy, &dummy, nullptr, nullptr);
    size_t need = 0;
    run_wgrad(ps, std::max(s->in_channels, s->out_channels), s->batch, s->length, s->ld, s->halo, nullptr, 0, true, &need,
              KC_WGRAD, nullptr);
    return need;
}

int wn_conv_backward_weights(const wn_conv_shape* s, const float* x, const float* dy, float* dweight, float* dbias,
                             void* workspace, size_t workspace_bytes, wn_stream_t stream) {
    int off[WN_MAX_TAPS];
    int rc = check_conv(s, off);
    if (rc != WN_OK) return rc;
    if (!x || !dy || !dweight) return WN_ERR_NULL;
    std::vector<PairSpec> ps = conv_pairs(s, off, x, dy, dweight, dbias);
    return run_wgrad(ps, std::max(s->in_channels, s->out_channels), s->batch, s->length, s->ld, s->halo, workspace,
                     workspace_bytes, false, nullptr, KC_WGRAD, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------
// next-sample NLL head
// ------------------------------------------------------------------------------------------
size_t wn_nll_partials(int batch, int length) {
    if (batch <= 0 || length <= 0) return 0;
    return (size_t)(((long long)batch * ((length + 3) / 4) + 255) / 256);
}

int wn_nll_forward(const float* logits, const long long* target, float* lse, float* partial, int* bad_targets, int batch,
                   int classes, int length, wn_stream_t stream) {
    if (batch <= 0 || classes <= 0 || length <= 0) return WN_ERR_BAD_SHAPE;
    if (!logits || !target || !lse || !partial) return WN_ERR_NULL;
    WN_HIP(launch_nll_forward(logits, target, lse, partial, bad_targets, batch, classes, length, (hipStream_t)stream), "nll_forward");
    return WN_OK;
}

int wn_nll_backward(const float* logits, const long long* target, const float* lse, const float* gscale, float* dlogits,
                    int batch, int classes, int length, wn_stream_t stream) {
    if (batch <= 0 || classes <= 0 || length <= 0) return WN_ERR_BAD_SHAPE;
    if (!logits || !target || !lse || !gscale || !dlogits) return WN_ERR_NULL;
    WN_HIP(launch_nll_backward(logits, target, lse, gscale, dlogits, batch, classes, length, (hipStream_t)stream), "nll_backward");
    return WN_OK;
}

// ------------------------------------------------------------------------------------------
// profiling hooks
// ------------------------------------------------------------------------------------------
int wn_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    g_prof.on = on != 0;
    return WN_OK;
}

int wn_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    for (ProfRec& r : g_prof.pending) { g_prof.pool.push_back(r.e0); g_prof.pool.push_back(r.e1); }
    g_prof.pending.clear();
    for (int i = 0; i < KC_COUNT; ++i) { g_prof.ms[i] = 0; g_prof.n[i] = 0; g_prof.flops[i] = 0; }
    return WN_OK;
}

int wn_prof_collect(void) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    for (ProfRec& r : g_prof.pending) {
        WN_HIP(hipEventSynchronize(r.e1), "hipEventSynchronize");
        float ms = 0;
        WN_HIP(hipEventElapsedTime(&ms, r.e0, r.e1), "hipEventElapsedTime");
        g_prof.ms[r.kc] += ms; g_prof.n[r.kc] += 1; g_prof.flops[r.kc] += r.flops;
        g_prof.pool.push_back(r.e0); g_prof.pool.push_back(r.e1);
    }
    g_prof.pending.clear();
    return WN_OK;
}

int wn_prof_num_kernels(void) { return KC_COUNT; }
const char* wn_prof_kernel_name(int kc) { return (kc >= 0 && kc < KC_COUNT) ? kKernelNames[kc] : ""; }

int wn_prof_get(int kc, double* total_ms, long long* launches, double* flops) {
    if (kc < 0 || kc >= KC_COUNT) return WN_ERR_BAD_SHAPE;
    std::lock_guard<std::mutex> lk(g_prof.mu);
    if (total_ms) *total_ms = g_prof.ms[kc];
    if (launches) *launches = g_prof.n[kc];
    if (flops) *flops = g_prof.flops[kc];
    return WN_OK;
}

