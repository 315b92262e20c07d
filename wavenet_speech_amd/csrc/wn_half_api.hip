// C ABI of the half-precision-MFMA modes (f16x3 / f16 / bf16) of the residual-block path: shape checks, GEMM plans,
// power-of-two scale bookkeeping and launches.  Host code only; kernels: wn_half.hip, wn_half_wgrad.hip, and the
// split-K reduction of wn_wgrad.hip.
//
// Scales (all exact powers of two, so they never cost a rounding):
//   residual stream x / r      stored as  value * kResidualScale (1/16): fp16 then holds |r| up to 1.0e6
//   sg, z                      stored as is (|.| <= 1); tanh is not stored: z / sg
//   gradient series            stored as  value * s, s = a per-backward-call device scalar chosen by the host from max|d skips_sum|
//   packed weights             256 * w / (scale of the segment's input), so an accumulator is 256 * true result
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/wavenet_amd.h"
#include "wn_half.h"

using namespace wn;

namespace wn { int hip_fail_shared(hipError_t e, const char* what); }

namespace {

#define WN_HIP(call, what)                                           \
    do {                                                             \
        hipError_t e__ = (call);                                     \
        if (e__ != hipSuccess) return wn::hip_fail_shared(e__, what); \
    } while (0)

inline int rup(int x, int m) { return (x + m - 1) / m * m; }
inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline int cp32(int c) { return rup(c, 32); }
inline size_t align256(size_t b) { return (b + 255) / 256 * 256; }

bool half_prec(int p) { return p == WN_F16X3 || p == WN_F16 || p == WN_BF16; }

int check_hlayout(int B, int L, int ld, int halo, int max_abs_off) {
    if (B <= 0 || L <= 0) return WN_ERR_BAD_SHAPE;
    if (halo < 0 || halo < max_abs_off) return WN_ERR_BAD_SHAPE;
    if (ld < 2 * halo + rup(L, kHCol)) return WN_ERR_BAD_SHAPE;
    return WN_OK;
}

int check_hblock(const wn_block_shape* s, int prec, int* off) {
    if (!s) return WN_ERR_NULL;
    if (!half_prec(prec)) return WN_ERR_UNSUPPORTED;
    if (s->in_channels <= 0 || s->out_channels <= 0 || s->skip_rows <= 0 || s->dilation <= 0) return WN_ERR_BAD_SHAPE;
    if (s->kernel_width < 1) return WN_ERR_BAD_SHAPE;
    if (s->kernel_width > WN_MAX_TAPS) return WN_ERR_UNSUPPORTED;
    if (s->in_channels > WN_MAX_CHANNELS || s->out_channels > WN_MAX_CHANNELS || s->skip_rows > WN_MAX_CHANNELS)
        return WN_ERR_UNSUPPORTED;
    wn_tap_offsets(s->kernel_width, s->dilation, s->causal, off);
    int mx = 0;
    for (int j = 0; j < s->kernel_width; ++j) mx = std::max(mx, std::abs(off[j]));
    return check_hlayout(s->batch, s->length, s->ld, s->halo, mx);
}

// a half series as the kernels address it
struct HView { char* base; long long ustride, pstride; int cp; };
HView view(const void* p, int channels, int ld, int planes) {
    HView v;
    v.base = (char*)p;
    v.cp = cp32(channels);
    v.pstride = (long long)(v.cp / 8) * ld * 16;
    v.ustride = v.pstride * planes;
    return v;
}
HDst dst_of(const HView& v) { HDst d; d.base = v.base; d.ustride = v.ustride; d.pstride = v.pstride; d.cp = v.cp; d._pad = 0; return d; }

// ---- GEMM plans -------------------------------------------------------------------------------------------------
struct HPlan {
    int MT = 4, rows = 256, nslab = 0, nseg = 0, planes = 1;
    int k32 = 0;        // 1: hgemm8_kernel (v_mfma 16x16x32, 256 x 128 tiles, 32-channel k-steps); weights packed for it
    int seg_nks[kMaxSeg] = {0};
    long long slab_woff[kHMaxSlab] = {0};
    int slab_nseg[kHMaxSlab] = {0}, slab_row0[kHMaxSlab] = {0}, slab_boff[kHMaxSlab] = {0};
    long long wbytes = 0;
    int bfloats = 0;
    // Tile choice (measured at 256 ch x 16 x 16000, f16x3, same run; profiles/r02/half_tuning.txt):
    //   long_k = false (the block GEMMs, K <= 5 C): 128-row tiles, two workgroups per CU -- their epilogues are HBM-bound
    //            store bursts (gate writes three tensors) that overlap the other workgroup's K loop: gate 0.510 -> 0.467 ms,
    //            dz 0.370 -> 0.302, res 0.242 -> 0.229, dx 0.413 -> 0.403
    //   long_k = true  (skips_sum, K = 30 C): 256-row tiles, one workgroup per CU -- half the weight re-staging: 2.20 vs 2.53 ms
    //   hgemm8_kernel (k32): the 16x16x32 MFMA shape sustains a higher clock at the power limit.  One eight-wave workgroup
    //            per CU on a 256-row tile, weights packed in 32-channel k-steps.  f16x3 runs it on 256 x 256 tiles (`wide`,
    //            two stages; L a multiple of 16); the 256 x 128 / three-stage form needs L a multiple of 128.  Measured
    //            against hgemm_kernel at cfg3 (alternating runs on one device, ms per launch): gate 0.363 vs 0.386, skips_sum
    //            2.03 vs 2.24, dx 0.381 vs 0.380, res 0.201 vs 0.199, dz 0.317 vs 0.300; step 65.5 vs 66.7 ms -> `prefer` is set
    //            for the gate and skips_sum GEMMs.  The one-plane modes lose with it (cfg5 f16 89.1 vs 84.7 ms/step).
    //            WN_HGEMM16=0: never; =2: wherever the shape allows, every mode (tests); WN_HGEMM16_COLS=128: narrow form.
    //            One-plane modes: the same code as FOUR waves on 128 x 256 tiles, two workgroups per CU (`quad`; a 32-channel
    //            stage is only 24 KiB there): every block GEMM whose row count is a multiple of 128 (`allow_quad`).
    int wide = 0, quad = 0;
    void init(int out_rows, int planes_, bool long_k = false, int length = 0, bool prefer_k32 = false, bool allow_quad = false) {
        static const int knob = getenv("WN_HGEMM16") ? atoi(getenv("WN_HGEMM16")) : 1;
        static const bool narrow_only = getenv("WN_HGEMM16_COLS") && atoi(getenv("WN_HGEMM16_COLS")) == 128;
        const bool rows_ok = out_rows % 256 == 0 && length > 0;
        const bool fits_wide = rows_ok && planes_ == 2 && length % 16 == 0 && !narrow_only;
        const bool fits_narrow = rows_ok && length % 128 == 0;
        quad = (knob != 0 && knob != 2 && allow_quad && planes_ == 1 && out_rows % 128 == 0 && length > 0 && length % 16 == 0) ? 1 : 0;
        k32 = (quad || (knob == 2 && (fits_wide || fits_narrow)) ||
               (knob == 1 && prefer_k32 && planes_ == 2 && (fits_wide || fits_narrow))) ? 1 : 0;
        wide = (k32 && !quad && fits_wide) ? 1 : 0;
        MT = quad ? 2 : ((k32 || (long_k && out_rows > 128)) ? 4 : 2);
        rows = 64 * MT;
        planes = planes_;
    }
    int kernel() const { return quad ? 10 : (k32 ? (wide ? 9 : 8) : MT); } // what launch_hgemm is told
    int wave_rows() const { return (k32 && !quad) ? 4 : 2; } // wave rows of the workgroup (each wave owns rows / wave_rows rows)
    bool add_slab(int nseg_used, int row0) {
        if (nslab >= kHMaxSlab) return false;
        const int s = nslab++;
        long long ks = 0;
        for (int i = 0; i < nseg_used; ++i) ks += seg_nks[i];
        slab_woff[s] = wbytes;
        slab_nseg[s] = nseg_used;
        slab_row0[s] = row0;
        slab_boff[s] = bfloats;
        wbytes += ks * planes * 2LL * rows * 16;
        bfloats += rows;
        return true;
    }
    size_t bytes() const { return align256((size_t)wbytes) + align256((size_t)bfloats * 4); }
};

// fused forward (wn_fused.hip): one launch for gate -> z -> res (+ skip), z stays in registers.  One-plane modes, all channel
// counts <= 128 (one wave owns every output channel of its 32 time columns), the gate's K within the x-fragment register budget.
struct HFusedPlan {
    bool on = false;
    int nzt = 0, nci16 = 0, nkg = 0, ngh = 0;     // z tiles of 32 channels; cp32(Ci)/16; gate k-steps; gate halves of 64 channels
    int st_gate = 0, st_res = 0, st_skip = 0;     // ring stages (32 channels of K x 128 rows) per gate half / res / skip phase
    size_t off_w = 0, off_bias = 0;
    int stages() const { return ngh * st_gate + st_res + st_skip; }
    size_t bytes() const { return on ? align256((size_t)stages() * kFStageBytes) + align256(4 * kFRows * sizeof(float)) + 1024 : 0; }   // + dump line
};

bool fused_forward_enabled() {
    const char* e = getenv("WN_FUSED_FWD");      // read per call: the parity tests compare both forms in one process
    return !(e && atoi(e) == 0);
}

// column-owner dz / dx (wn_col.hip): the blocks the fused forward takes, when the skip path is as wide as the block
bool col_backward_enabled() {
    const char* e = getenv("WN_COL_BWD");        // read per call: the parity tests compare both forms in one process
    return !(e && atoi(e) == 0);
}

struct HBlockPlan {
    HPlan fa, fr, fs, ka, kb;
    HFusedPlan fu;
    bool col = false;               // dz and dx run as hcol_kernel (KA / KB keep hgemm_kernel's 128-row packing for it)
    size_t off_fa = 0, off_fr = 0, off_fs = 0, off_ka = 0, off_kb = 0, off_kap = 0, total = 0;
};

HBlockPlan plan_hblock(const wn_block_shape* s, int prec) {
    HBlockPlan p;
    const int P = hp_planes(prec);
    const int Ci = s->in_channels, Co = s->out_channels, Ms = s->skip_rows, k = s->kernel_width;
    const bool fused_ok = fused_forward_enabled() && P == 1 && k == 2 && cp32(Ci) == cp32(Co) && cp32(Co) <= 128 && Ms <= 128 &&
                          k * cp32(Ci) / 16 <= kFMaxGateK;
    p.col = fused_ok && col_backward_enabled() && cp32(Ms) == cp32(Co);
    {   // FA: [a ; g] interleaved in 32-channel tile pairs; K = k taps of x
        HPlan& g = p.fa;
        g.init(2 * Co, P, false, s->length, true, true);
        g.nseg = k;
        for (int j = 0; j < k; ++j) g.seg_nks[j] = cp32(Ci) / 16;
        const int ch_per_slab = g.rows / 2;
        for (int c0 = 0; c0 < Co; c0 += ch_per_slab) g.add_slab(k, c0);
    }
    {   // FR: r rows contract [z ; x]
        HPlan& g = p.fr;
        g.init(Co, P, false, s->length, false, true);
        g.nseg = 2;
        g.seg_nks[0] = cp32(Co) / 16;
        g.seg_nks[1] = cp32(Ci) / 16;
        for (int r0 = 0; r0 < Co; r0 += g.rows) g.add_slab(2, r0);
    }
    {   // FS: skip rows contract [z]
        HPlan& g = p.fs;
        g.init(Ms, P);                                   // inference: accumulates per block -- always hgemm_kernel
        g.nseg = 1;
        g.seg_nks[0] = cp32(Co) / 16;
        for (int r0 = 0; r0 < Ms; r0 += g.rows) g.add_slab(1, r0);
    }
    {   // KA: dz rows (z channels) contract [dskip ; dr]
        HPlan& g = p.ka;
        g.init(Co, P, false, s->length, false, !p.col);
        g.nseg = 2;
        g.seg_nks[0] = cp32(Ms) / 16;
        g.seg_nks[1] = cp32(Co) / 16;
        for (int r0 = 0; r0 < Co; r0 += g.rows) g.add_slab(2, r0);
    }
    {   // KB: dx rows (input channels) contract [da_0; dg_0; ...; dr]
        HPlan& g = p.kb;
        g.init(Ci, P, false, s->length, false, !p.col);
        g.nseg = 2 * k + 1;
        for (int j = 0; j < 2 * k + 1; ++j) g.seg_nks[j] = cp32(Co) / 16;
        for (int r0 = 0; r0 < Ci; r0 += g.rows) g.add_slab(2 * k + 1, r0);
    }
    {
        HFusedPlan& f = p.fu;
        f.nci16 = cp32(Ci) / 16; f.nzt = cp32(Co) / 32; f.nkg = k * f.nci16; f.ngh = (f.nzt + 1) / 2;
        f.st_gate = f.nkg / 2; f.st_res = f.nzt + f.nci16 / 2; f.st_skip = f.nzt;
        // (two taps and equally padded channel counts: the kernel is instantiated per z-tile count with every loop bound a constant)
        f.on = fused_ok;
    }
    // a block that runs the fused forward packs no separate gate / res / skip weights
    p.off_fa = 0;
    p.off_fr = p.off_fa + (p.fu.on ? 0 : p.fa.bytes());
    p.off_fs = p.off_fr + (p.fu.on ? 0 : p.fr.bytes());
    p.off_ka = p.off_fs + (p.fu.on ? 0 : p.fs.bytes());
    p.off_kb = p.off_ka + p.ka.bytes();
    p.off_kap = p.off_kb + p.kb.bytes();             // KA again with the dr segment in accumulator order (wn_col2.hip), col blocks only
    p.fu.off_w = p.off_kap + (p.col ? p.ka.bytes() : 0);
    p.fu.off_bias = p.fu.off_w + align256((size_t)p.fu.stages() * kFStageBytes);
    p.total = p.fu.off_w + p.fu.bytes();
    return p;
}

void fill_hpack(HPackArgs& a, const HPlan& g, void* packed, size_t off, int prec) {
    std::memset(&a, 0, sizeof(a));
    a.nslab = g.nslab;
    a.rows = g.rows;
    a.planes = g.planes;
    a.bf16 = prec == WN_BF16;
    a.kgroups = g.k32 ? 4 : 2;                                   // k-groups of 8 channels per k-step
    for (int i = 0; i < kMaxSeg; ++i) a.seg_nks[i] = g.seg_nks[i] / (g.k32 ? 2 : 1);
    for (int i = 0; i < g.nslab; ++i) {
        a.slab_woff[i] = g.slab_woff[i];
        a.slab_nseg[i] = g.slab_nseg[i];
        a.slab_boff[i] = g.slab_boff[i];
    }
    a.wpacked = (char*)packed + off;
    a.bias = reinterpret_cast<float*>((char*)packed + off + align256((size_t)g.wbytes));
    a.total_units = g.wbytes / (16 * g.planes);
    a.set[0].bias_scale = a.set[1].bias_scale = 1.0f;
}

inline HPackSrc hsrc(const float* p, int rows, int cols, int sr, int sc, float scale) {
    HPackSrc s; s.ptr = p; s.rows = rows; s.cols = cols; s.stride_r = sr; s.stride_c = sc; s.scale = scale; s.flags = 0; return s;
}

// tiles of a plain plan: row tile i of slab sl covers rows slab_row0 + 32 i
void plain_tiles(HPackArgs& a, const HPlan& g, int valid_rows) {
    for (int sl = 0; sl < g.nslab; ++sl)
        for (int i = 0; i < g.rows / 32; ++i) {
            const int row0 = g.slab_row0[sl] + 32 * i;
            a.tile[sl * (g.rows / 32) + i].set = 0;
            a.tile[sl * (g.rows / 32) + i].row0 = row0 < valid_rows ? row0 : -1;
        }
}

void fill_hgemm(HGemmArgs& a, const HPlan& g, const void* packed, size_t off, int B, int L, int ld, int halo) {
    std::memset(&a, 0, sizeof(a));
    a.wpacked = (const char*)packed + off;
    a.bias = reinterpret_cast<const float*>((const char*)packed + off + align256((size_t)g.wbytes));
    a.nslab = g.nslab;
    for (int i = 0; i < g.nslab; ++i) {
        a.slab[i].woff = g.slab_woff[i];
        a.slab[i].nseg = g.slab_nseg[i];
        a.slab[i].row0 = g.slab_row0[i];
        a.slab[i].boff = g.slab_boff[i];
        a.slab[i].dst = 0;
    }
    a.B = B; a.L = L; a.ld = ld; a.halo = halo;
    a.oscale = 1.0f / kWeightScale;
}

inline void set_hseg(HGemmArgs& a, int i, const HView& v, int off, int nks) {
    a.seg[i].base = v.base; a.seg[i].ustride = v.ustride; a.seg[i].pstride = v.pstride; a.seg[i].off = off; a.seg[i].nks = nks;
}

// profiling classes shared with wn_api.hip (same table, same order)
enum { KC_PACK = 0, KC_GATE_GEMM, KC_OUT_GEMM, KC_DZ_GEMM, KC_DX_GEMM, KC_WGRAD, KC_WGRAD_REDUCE, KC_CONV_FWD, KC_CONV_BWD_DATA,
       KC_SKIP_GEMM, KC_HLOAD, KC_HGATE, KC_HRES, KC_HDZ, KC_HDX, KC_HSKIP, KC_HWGRAD, KC_EMBED, KC_SYNTH, KC_CTC, KC_HFUSED, KC_HCONV_FWD, KC_HCONV_BWD_DATA, KC_HCOL_DZ, KC_HCOL_DX, KC_HCOL_DXDZ, KC_HCOL_SKIP };

}  // namespace

namespace wn {
struct ProfScopeShared {   // implemented in wn_api.hip (HIP events on the launch stream when profiling is on)
    void* impl;
    ProfScopeShared(int kc, double flops, hipStream_t st);
    ~ProfScopeShared();
};
}  // namespace wn

// ==========================================================================================================================
// C ABI
// ==========================================================================================================================
int wn_hseries_layout(int length, int max_abs_offset, int* ld, int* halo) {
    if (!ld || !halo) return WN_ERR_NULL;
    if (length <= 0 || max_abs_offset < 0) return WN_ERR_BAD_SHAPE;
    *halo = rup(max_abs_offset, 8);   // t = 0 then starts a 128-byte line of every channel group
    *ld = 2 * (*halo) + rup(length, kHCol);
    return WN_OK;
}

size_t wn_hseries_bytes(int precision, int batch, int channels, int ld) {
    if (!half_prec(precision) || batch <= 0 || channels <= 0 || ld <= 0) return 0;
    return (size_t)batch * hp_planes(precision) * (cp32(channels) / 8) * (size_t)ld * 16;
}

float wn_hseries_residual_scale(void) { return kResidualScale; }

int wn_hseries_load(int precision, const float* dense, void* series, int batch, int channels, int length, int ld, int halo,
                    float scale, const float* dyn_scale, unsigned* overflow_flag, wn_stream_t stream) {
    if (!half_prec(precision)) return WN_ERR_UNSUPPORTED;
    if (!dense || !series) return WN_ERR_NULL;
    int rc = check_hlayout(batch, length, ld, halo, 0);
    if (rc != WN_OK) return rc;
    if (channels <= 0 || channels > WN_MAX_CHANNELS) return WN_ERR_BAD_SHAPE;
    HLoadArgs a;
    std::memset(&a, 0, sizeof(a));
    a.src = dense; a.dst = (char*)series; a.dyn_scale = dyn_scale; a.flag = overflow_flag; a.scale = scale;
    a.B = batch; a.C = channels; a.L = length; a.G = cp32(channels) / 8; a.ld = ld; a.halo = halo;
    a.planes = hp_planes(precision); a.bf16 = precision == WN_BF16;
    wn::ProfScopeShared prof(KC_HLOAD, 0.0, (hipStream_t)stream);
    WN_HIP(launch_hload(a, (hipStream_t)stream), "hload");
    return WN_OK;
}

size_t wn_hblock_packed_bytes(const wn_block_shape* s, int precision) {
    int off[WN_MAX_TAPS];
    if (check_hblock(s, precision, off) != WN_OK) return 0;
    return plan_hblock(s, precision).total;
}

namespace {
// the pack jobs of one block (gate, res, skip, dz, dx), `packed` = where the block's packed weights will live
int fill_hblock_jobs(const wn_block_shape* s, int precision, const wn_block_params* p, void* packed, std::vector<HPackArgs>& jobs) {
    int off[WN_MAX_TAPS];
    int rc = check_hblock(s, precision, off);
    if (rc != WN_OK) return rc;
    if (!p || !packed || !p->w_tanh || !p->w_sigmoid || !p->w_res || !p->w_skip || !p->w_proj) return WN_ERR_NULL;
    const HBlockPlan bp = plan_hblock(s, precision);
    const int Ci = s->in_channels, Co = s->out_channels, Ms = s->skip_rows, k = s->kernel_width;
    const float WS = kWeightScale, RS = kResidualScale;
    HPackArgs a;
    if (bp.fu.on) {
        // fused forward: every phase is 128 rows; the stream is gate half 0, [gate half 1], res, skip (wn_fused.hip).  The biases
        // become accumulator start values: bias / (output scale of the phase) = bias * WS in all three cases.
        const HFusedPlan& f = bp.fu;
        auto blank = [&](int nslab, size_t woff, int boff_floats) {
            std::memset(&a, 0, sizeof(a));
            a.nslab = nslab; a.rows = kFRows; a.planes = 1; a.bf16 = precision == WN_BF16; a.kgroups = 2;
            a.wpacked = (char*)packed + f.off_w + woff;
            a.bias = reinterpret_cast<float*>((char*)packed + f.off_bias) + boff_floats;
            a.set[0].bias_scale = a.set[1].bias_scale = WS;
        };
        {   // gate: slab = half of 64 channels, row tiles [a(c), g(c), a(c + 32), g(c + 32)]
            blank(f.ngh, 0, 0);
            for (int j = 0; j < k; ++j) {
                a.seg_nks[j] = f.nci16;
                a.set[0].seg[j] = hsrc(p->w_tanh + j, Co, Ci, Ci * k, k, WS / RS);
                a.set[1].seg[j] = hsrc(p->w_sigmoid + j, Co, Ci, Ci * k, k, WS / RS);
            }
            a.set[0].bias0 = p->b_tanh; a.set[0].bias_rows = Co;
            a.set[1].bias0 = p->b_sigmoid; a.set[1].bias_rows = Co;
            for (int hf = 0; hf < f.ngh; ++hf) {
                a.slab_woff[hf] = (long long)hf * f.st_gate * kFStageBytes; a.slab_nseg[hf] = k; a.slab_boff[hf] = kFRows * hf;
                for (int i = 0; i < 4; ++i) {
                    const int ch0 = 64 * hf + 32 * (i >> 1);
                    a.tile[hf * 4 + i].set = i & 1;
                    a.tile[hf * 4 + i].row0 = ch0 < Co ? ch0 : -1;
                }
            }
            a.total_units = (long long)f.ngh * f.st_gate * kFStageBytes / 16;
            jobs.push_back(a);
        }
        {   // res: K = [z in accumulator order ; x], stored as r * RS
            blank(1, (size_t)f.ngh * f.st_gate * kFStageBytes, 2 * kFRows);
            a.seg_nks[0] = 2 * f.nzt; a.seg_nks[1] = f.nci16;
            a.set[0].seg[0] = hsrc(p->w_res, Co, Co, Co, 1, WS); a.set[0].seg[0].flags = HPACK_PERM;
            a.set[0].seg[1] = hsrc(p->w_proj, Co, Ci, Ci, 1, WS / RS);
            a.set[0].bias0 = p->b_res; a.set[0].bias1 = p->b_proj; a.set[0].bias_rows = Co;
            a.slab_nseg[0] = 2;
            for (int i = 0; i < 4; ++i) { a.tile[i].set = 0; a.tile[i].row0 = 32 * i < Co ? 32 * i : -1; }
            a.total_units = (long long)f.st_res * kFStageBytes / 16;
            jobs.push_back(a);
        }
        {   // skip (the per-block form of inference; training forms skips_sum afterwards from every block's z)
            blank(1, (size_t)(f.ngh * f.st_gate + f.st_res) * kFStageBytes, 3 * kFRows);
            a.seg_nks[0] = 2 * f.nzt;
            a.set[0].seg[0] = hsrc(p->w_skip, Ms, Co, Co, 1, WS); a.set[0].seg[0].flags = HPACK_PERM;
            a.set[0].bias0 = p->b_skip; a.set[0].bias_rows = Ms;
            a.slab_nseg[0] = 1;
            for (int i = 0; i < 4; ++i) { a.tile[i].set = 0; a.tile[i].row0 = 32 * i < Ms ? 32 * i : -1; }
            a.total_units = (long long)f.st_skip * kFStageBytes / 16;
            jobs.push_back(a);
        }
    } else {
        {   // FA: gate rows; the input x is stored as x * RS
            const HPlan& g = bp.fa;
            fill_hpack(a, g, packed, bp.off_fa, precision);
            for (int j = 0; j < k; ++j) {
                a.set[0].seg[j] = hsrc(p->w_tanh + j, Co, Ci, Ci * k, k, WS / RS);
                a.set[1].seg[j] = hsrc(p->w_sigmoid + j, Co, Ci, Ci * k, k, WS / RS);
            }
            a.set[0].bias0 = p->b_tanh; a.set[0].bias_rows = Co;
            a.set[1].bias0 = p->b_sigmoid; a.set[1].bias_rows = Co;
            const int tiles = g.rows / 32, per_wave = tiles / g.wave_rows();   // a wave owns per_wave consecutive tiles = per_wave/2 (a, g) pairs
            for (int sl = 0; sl < g.nslab; ++sl)
                for (int i = 0; i < tiles; ++i) {
                    const int wm = i / per_wave, tw = i % per_wave;            // wave row, tile inside the wave
                    const int ch0 = g.slab_row0[sl] + wm * 16 * per_wave + 32 * (tw / 2);
                    a.tile[sl * tiles + i].set = tw & 1;
                    a.tile[sl * tiles + i].row0 = ch0 < Co ? ch0 : -1;
                }
            jobs.push_back(a);
        }
        {   // FR: r = W_res z + W_proj x + b, stored as r * RS
            const HPlan& g = bp.fr;
            fill_hpack(a, g, packed, bp.off_fr, precision);
            a.set[0].seg[0] = hsrc(p->w_res, Co, Co, Co, 1, WS);
            a.set[0].seg[1] = hsrc(p->w_proj, Co, Ci, Ci, 1, WS / RS);
            a.set[0].bias0 = p->b_res; a.set[0].bias1 = p->b_proj; a.set[0].bias_rows = Co; a.set[0].bias_scale = RS;
            plain_tiles(a, g, Co);
            jobs.push_back(a);
        }
        {   // FS: skip = W_skip z + b (fp32 out)
            const HPlan& g = bp.fs;
            fill_hpack(a, g, packed, bp.off_fs, precision);
            a.set[0].seg[0] = hsrc(p->w_skip, Ms, Co, Co, 1, WS);
            a.set[0].bias0 = p->b_skip; a.set[0].bias_rows = Ms;
            plain_tiles(a, g, Ms);
            jobs.push_back(a);
        }
    }
    {   // KA: rows = z channel c; seg0 cols = skip row m: w_skip[m][c]; seg1 cols = r row m: w_res[m][c]
        const HPlan& g = bp.ka;
        fill_hpack(a, g, packed, bp.off_ka, precision);
        a.set[0].seg[0] = hsrc(p->w_skip, Co, Ms, 1, Co, WS);
        a.set[0].seg[1] = hsrc(p->w_res, Co, Co, 1, Co, WS);
        plain_tiles(a, g, Co);
        a.bias = nullptr;
        jobs.push_back(a);
    }
    if (bp.col) {   // KAP: KA with the dr segment's k in accumulator order: the dx tile of the block above is fed back as it stands
        const HPlan& g = bp.ka;
        fill_hpack(a, g, packed, bp.off_kap, precision);
        a.set[0].seg[0] = hsrc(p->w_skip, Co, Ms, 1, Co, WS);
        a.set[0].seg[1] = hsrc(p->w_res, Co, Co, 1, Co, WS); a.set[0].seg[1].flags = HPACK_PERM;
        plain_tiles(a, g, Co);
        a.bias = nullptr;
        jobs.push_back(a);
    }
    {   // KB: rows = input channel; cols = output channel
        const HPlan& g = bp.kb;
        fill_hpack(a, g, packed, bp.off_kb, precision);
        for (int j = 0; j < k; ++j) {
            a.set[0].seg[2 * j] = hsrc(p->w_tanh + j, Ci, Co, k, Ci * k, WS);
            a.set[0].seg[2 * j + 1] = hsrc(p->w_sigmoid + j, Ci, Co, k, Ci * k, WS);
        }
        a.set[0].seg[2 * k] = hsrc(p->w_proj, Ci, Co, 1, Ci, WS);
        plain_tiles(a, g, Ci);
        a.bias = nullptr;
        jobs.push_back(a);
    }
    return WN_OK;
}
}  // namespace

int wn_hblock_pack_checked(const wn_block_shape* s, int precision, const wn_block_params* p, void* packed, unsigned* overflow_flag,
                           wn_stream_t stream) {
    std::vector<HPackArgs> jobs;
    int rc = fill_hblock_jobs(s, precision, p, packed, jobs);
    if (rc != WN_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    wn::ProfScopeShared prof(KC_PACK, 0.0, st);
    for (HPackArgs& a : jobs) {
        a.flag = overflow_flag;
        WN_HIP(launch_hpack(a, st), "hpack(block)");
    }
    return WN_OK;
}

int wn_hblock_pack(const wn_block_shape* s, int precision, const wn_block_params* p, void* packed, wn_stream_t stream) {
    return wn_hblock_pack_checked(s, precision, p, packed, nullptr, stream);
}

int wn_hblock_forward_is_fused(const wn_block_shape* s, int precision) {
    int off[WN_MAX_TAPS];
    if (check_hblock(s, precision, off) != WN_OK) return 0;
    return plan_hblock(s, precision).fu.on ? 1 : 0;
}

int wn_hblock_forward(const wn_block_shape* s, int precision, const void* packed, const void* x, void* r_out,
                      float* skip_dense, int skip_accumulate, void* sg, void* z, unsigned* overflow_flag,
                      wn_stream_t stream) {
    int off[WN_MAX_TAPS];
    int rc = check_hblock(s, precision, off);
    if (rc != WN_OK) return rc;
    if (!packed || !x) return WN_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    const HBlockPlan bp = plan_hblock(s, precision);
    const int P = hp_planes(precision);
    const int Ci = s->in_channels, Co = s->out_channels, Ms = s->skip_rows, k = s->kernel_width;
    const double BL = (double)s->batch * s->length;
    const bool fused = bp.fu.on;
    if (!z && !(fused && !sg)) return WN_ERR_NULL;            // z may be omitted only where the fused inference form runs
    const HView vx = view(x, Ci, s->ld, P), vz = view(z, Co, s->ld, P);
    if (fused) {
        const HFusedPlan& f = bp.fu;
        HFusedArgs fa;
        std::memset(&fa, 0, sizeof(fa));
        fa.wstream = (const char*)packed + f.off_w;
        fa.bias = reinterpret_cast<const float*>((const char*)packed + f.off_bias);
        fa.x = vx.base; fa.x_ustride = vx.ustride;
        if (z) fa.z = dst_of(vz);
        if (sg) fa.sg = dst_of(view(sg, Co, s->ld, P));
        if (r_out) fa.r = dst_of(view(r_out, Co, s->ld, P));
        fa.skip = skip_dense; fa.skip_rows = Ms; fa.skip_accum = skip_accumulate ? 1 : 0;
        fa.flag = overflow_flag;
        fa.dump = (char*)packed + f.off_bias + align256(4 * kFRows * sizeof(float));
        for (int j = 0; j < k; ++j)
            for (int ks = 0; ks < f.nci16; ++ks) fa.xunit[j * f.nci16 + ks] = 2 * ks * s->ld + off[j];
        fa.nkg = f.nkg; fa.nci16 = f.nci16; fa.nzt = f.nzt; fa.co = Co;
        fa.do_res = r_out ? 1 : 0; fa.do_skip = skip_dense ? 1 : 0;
        fa.jump_at = r_out ? 0x7fffffff : f.ngh * f.st_gate; fa.jump = r_out ? 0 : f.st_res;
        fa.osc_gate = 1.0f / kWeightScale; fa.osc_res = kResidualScale / kWeightScale; fa.osc_skip = 1.0f / kWeightScale;
        fa.B = s->batch; fa.L = s->length; fa.ld = s->ld; fa.halo = s->halo;
        fa.units_per_row = cdiv(s->length, 32); fa.nunit = s->batch * fa.units_per_row;
        fa.nstage = f.ngh * f.st_gate + (r_out ? f.st_res : 0) + (skip_dense ? f.st_skip : 0);
        wn::ProfScopeShared prof(KC_HFUSED, 2.0 * ((2.0 * Co) * (double)(k * Ci) + (r_out ? Co * (double)(Co + Ci) : 0.0) +
                                                 (skip_dense ? Ms * (double)Co : 0.0)) * BL, st);
        WN_HIP(launch_hfused_fwd(precision, fa, st), "hfused_fwd");
        return WN_OK;
    }
    HGemmArgs a;
    {
        const HPlan& g = bp.fa;
        fill_hgemm(a, g, packed, bp.off_fa, s->batch, s->length, s->ld, s->halo);
        for (int j = 0; j < k; ++j) set_hseg(a, j, vx, off[j], g.seg_nks[j]);
        a.z = dst_of(vz);
        if (sg) a.sg = dst_of(view(sg, Co, s->ld, P));
        a.gate_rows = Co;
        a.flag = overflow_flag;
        wn::ProfScopeShared prof(KC_HGATE, 2.0 * (2.0 * Co) * (double)(k * Ci) * BL, st);
        WN_HIP(launch_hgemm(precision, g.kernel(), HEPI_GATE, a, st), "hgemm<gate>");
    }
    if (r_out) {
        const HPlan& g = bp.fr;
        fill_hgemm(a, g, packed, bp.off_fr, s->batch, s->length, s->ld, s->halo);
        set_hseg(a, 0, vz, 0, g.seg_nks[0]);
        set_hseg(a, 1, vx, 0, g.seg_nks[1]);
        a.dst[0] = dst_of(view(r_out, Co, s->ld, P));
        a.oscale = kResidualScale / kWeightScale;
        a.flag = overflow_flag;
        wn::ProfScopeShared prof(KC_HRES, 2.0 * Co * (double)(Co + Ci) * BL, st);
        WN_HIP(launch_hgemm(precision, g.kernel(), HEPI_STORE, a, st), "hgemm<res>");
    }
    if (skip_dense) {
        const HPlan& g = bp.fs;
        fill_hgemm(a, g, packed, bp.off_fs, s->batch, s->length, s->ld, s->halo);
        set_hseg(a, 0, vz, 0, g.seg_nks[0]);
        a.out32 = skip_dense; a.out32_rows = Ms; a.out32_accum = skip_accumulate ? 1 : 0;
        wn::ProfScopeShared prof(KC_HSKIP, 2.0 * Ms * (double)Co * BL, st);
        WN_HIP(launch_hgemm(precision, g.kernel(), HEPI_F32, a, st), "hgemm<skip>");
    }
    return WN_OK;
}

namespace {
// an hgemm argument block (one slab of <= 128 rows, segments set) as the column-owner kernel's
void col_args_common(HColArgs& c, const HGemmArgs& a, int out_rows, char* dump, int B, int L, int ld, int halo) {
    std::memset(&c, 0, sizeof(c));
    c.wstream = a.wpacked + a.slab[0].woff;
    int kk = 0;
    for (int i = 0; i < a.slab[0].nseg; ++i)
        for (int j = 0; j < a.seg[i].nks && kk < kColMaxK; ++j, ++kk) {
            c.kbase[kk] = a.seg[i].base + ((long long)a.seg[i].off + 2LL * j * ld) * 16;
            c.kustride[kk] = a.seg[i].ustride;
        }
    c.nks = kk;
    c.nt = cp32(out_rows) / 32;
    c.flag = a.flag;
    c.dump = dump;
    c.oscale = a.oscale;
    c.B = B; c.L = L; c.ld = ld; c.halo = halo;
    c.nunit = (int)(((long long)B * L + 31) / 32);
}
void col_args(HColArgs& c, const HGemmArgs& a, const HBlockPlan& bp, const void* packed, const wn_block_shape* s) {
    col_args_common(c, a, s->out_channels, (char*)packed + bp.fu.off_bias + align256(4 * kFRows * sizeof(float)), s->batch, s->length,
                    s->ld, s->halo);
}
int hblock_backward_data_impl(const wn_block_shape* s, int precision, const void* packed, const void* dr, const void* dskip,
                              const void* z, const void* sg, void* da, void* dg, void* dx, float* dx_dense,
                              const float* dyn_inv_scale, const void* dx_mask, float dx_slope, unsigned* overflow_flag,
                              wn_stream_t stream, bool do_dz = true) {
    int off[WN_MAX_TAPS];
    int rc = check_hblock(s, precision, off);
    if (rc != WN_OK) return rc;
    if (!packed || !da || !dg || (do_dz && (!dskip || !z || !sg))) return WN_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    const HBlockPlan bp = plan_hblock(s, precision);
    const int P = hp_planes(precision);
    const int Ci = s->in_channels, Co = s->out_channels, Ms = s->skip_rows, k = s->kernel_width;
    const double BL = (double)s->batch * s->length;
    const HView vda = view(da, Co, s->ld, P), vdg = view(dg, Co, s->ld, P);
    HGemmArgs a;
    if (do_dz) {   // dz = W_skip^T dskip + W_res^T dr ; da, dg
        const HPlan& g = bp.ka;
        fill_hgemm(a, g, packed, bp.off_ka, s->batch, s->length, s->ld, s->halo);
        a.bias = nullptr;
        set_hseg(a, 0, view(dskip, Ms, s->ld, P), 0, g.seg_nks[0]);
        if (dr) set_hseg(a, 1, view(dr, Co, s->ld, P), 0, g.seg_nks[1]);
        for (int i = 0; i < a.nslab; ++i) a.slab[i].nseg = dr ? 2 : 1;
        a.z = dst_of(view(z, Co, s->ld, P)); a.sg = dst_of(view(sg, Co, s->ld, P));
        a.da = dst_of(vda); a.dg = dst_of(vdg);
        a.gate_rows = Co;
        a.flag = overflow_flag;
        if (bp.col) {
            HColArgs c;
            col_args(c, a, bp, packed, s);
            c.z = a.z; c.sg = a.sg; c.da = a.da; c.dg = a.dg;
            wn::ProfScopeShared prof(KC_HCOL_DZ, 2.0 * Co * (double)(Ms + (dr ? Co : 0)) * BL, st);
            WN_HIP(launch_hcol(precision, HEPI_DGATE, c, st), "hcol<dz>");
        } else {
            wn::ProfScopeShared prof(KC_HDZ, 2.0 * Co * (double)(Ms + (dr ? Co : 0)) * BL, st);
            WN_HIP(launch_hgemm(precision, g.kernel(), HEPI_DGATE, a, st), "hgemm<dz>");
        }
    }
    if (dx || dx_dense) {
        const HPlan& g = bp.kb;
        fill_hgemm(a, g, packed, bp.off_kb, s->batch, s->length, s->ld, s->halo);
        a.bias = nullptr;
        for (int j = 0; j < k; ++j) {
            set_hseg(a, 2 * j, vda, -off[j], g.seg_nks[2 * j]);
            set_hseg(a, 2 * j + 1, vdg, -off[j], g.seg_nks[2 * j + 1]);
        }
        if (dr) set_hseg(a, 2 * k, view(dr, Co, s->ld, P), 0, g.seg_nks[2 * k]);
        for (int i = 0; i < a.nslab; ++i) a.slab[i].nseg = 2 * k + (dr ? 1 : 0);
        a.flag = overflow_flag;
        const bool col_dx = dx && bp.col;
        wn::ProfScopeShared prof(col_dx ? KC_HCOL_DX : KC_HDX, 2.0 * Ci * (double)(2 * k * Co + (dr ? Co : 0)) * BL, st);
        if (dx && dx_mask && bp.col) {
            HColArgs c;
            col_args(c, a, bp, packed, s);
            c.dst = dst_of(view(dx, Ci, s->ld, P));
            c.mask = dst_of(view(dx_mask, Ci, s->ld, P));
            c.oscale2 = 1.0f; c.leaky = dx_slope;
            WN_HIP(launch_hcol(precision, HEPI_LEAKY, c, st), "hcol<dx masked>");
        } else if (dx && dx_mask) {
            // the block's input was leaky(.) of a front-end conv kept in the series: dx is masked by its stored activation
            a.dst[0] = dst_of(view(dx, Ci, s->ld, P));
            a.z = dst_of(view(dx_mask, Ci, s->ld, P));
            a.oscale2 = 1.0f; a.leaky = dx_slope;
            WN_HIP(launch_hgemm(precision, g.kernel(), HEPI_LEAKY, a, st), "hgemm<dx masked>");
        } else if (col_dx) {
            HColArgs c;
            col_args(c, a, bp, packed, s);
            c.dst = dst_of(view(dx, Ci, s->ld, P));
            WN_HIP(launch_hcol(precision, HEPI_STORE, c, st), "hcol<dx>");
        } else if (dx) {
            a.dst[0] = dst_of(view(dx, Ci, s->ld, P));
            WN_HIP(launch_hgemm(precision, g.kernel(), HEPI_STORE, a, st), "hgemm<dx>");
        } else {
            a.out32 = dx_dense; a.out32_rows = Ci; a.out32_accum = 0; a.dyn_inv = dyn_inv_scale;
            WN_HIP(launch_hgemm(precision, g.kernel(), HEPI_F32, a, st), "hgemm<dx dense>");
        }
    }
    return WN_OK;
}
}  // namespace

int wn_hblock_backward_data(const wn_block_shape* s, int precision, const void* packed, const void* dr, const void* dskip,
                            const void* z, const void* sg, void* da, void* dg, void* dx, float* dx_dense,
                            const float* dyn_inv_scale, unsigned* overflow_flag, wn_stream_t stream) {
    return hblock_backward_data_impl(s, precision, packed, dr, dskip, z, sg, da, dg, dx, dx_dense, dyn_inv_scale, nullptr, 1.0f,
                                     overflow_flag, stream);
}

// the input gradient alone, from gate gradients that exist already (computed by wn_hblock_backward_pair): dx in the series,
// masked by the activation in front of the stack (x_act), or dense
int wn_hblock_backward_input(const wn_block_shape* s, int precision, const void* packed, const void* dr, const void* da, const void* dg,
                             void* dx, float* dx_dense, const float* dyn_inv_scale, const void* x_act, float leaky_slope,
                             unsigned* overflow_flag, wn_stream_t stream) {
    if (!dx && !dx_dense) return WN_ERR_NULL;
    if (x_act && !dx) return WN_ERR_NULL;
    return hblock_backward_data_impl(s, precision, packed, dr, nullptr, nullptr, nullptr, const_cast<void*>(da), const_cast<void*>(dg), dx,
                                     dx_dense, dyn_inv_scale, x_act, leaky_slope, overflow_flag, stream, false);
}

// ---- dx of `upper` and dz of `lower` (the block below it in the stack) in one launch ---------------------------------------------
namespace {
bool pair_fusable(const wn_block_shape* u, const wn_block_shape* l, int precision, const HBlockPlan& pu, const HBlockPlan& pl) {
    const char* e = getenv("WN_COL_PAIR");       // read per call (tests compare both forms)
    return !(e && atoi(e) == 0) && pu.col && pl.col && cp32(u->in_channels) == cp32(l->out_channels) && u->in_channels == l->out_channels &&
           u->batch == l->batch && u->length == l->length && u->ld == l->ld && u->halo == l->halo && u->skip_rows == l->skip_rows &&
           pu.kb.nslab == 1 && pl.ka.nslab == 1 && !pu.kb.k32 && !pl.ka.k32 && hp_planes(precision) == 1;
}
}  // namespace

int wn_hblock_backward_pair_is_fused(const wn_block_shape* upper, const wn_block_shape* lower, int precision) {
    int off[WN_MAX_TAPS];
    if (check_hblock(upper, precision, off) != WN_OK || check_hblock(lower, precision, off) != WN_OK) return 0;
    return pair_fusable(upper, lower, precision, plan_hblock(upper, precision), plan_hblock(lower, precision)) ? 1 : 0;
}

int wn_hblock_backward_pair(const wn_block_shape* upper, const void* packed_upper, const wn_block_shape* lower, const void* packed_lower,
                            int precision, const void* dr_upper, const void* da_upper, const void* dg_upper, const void* dskip,
                            const void* z_lower, const void* sg_lower, void* dx_upper, void* da_lower, void* dg_lower,
                            unsigned* overflow_flag, wn_stream_t stream) {
    int offu[WN_MAX_TAPS], offl[WN_MAX_TAPS];
    int rc = check_hblock(upper, precision, offu);
    if (rc != WN_OK) return rc;
    rc = check_hblock(lower, precision, offl);
    if (rc != WN_OK) return rc;
    if (!packed_upper || !packed_lower || !da_upper || !dg_upper || !dskip || !z_lower || !sg_lower || !dx_upper || !da_lower || !dg_lower)
        return WN_ERR_NULL;
    const HBlockPlan pu = plan_hblock(upper, precision), pl = plan_hblock(lower, precision);
    if (!pair_fusable(upper, lower, precision, pu, pl)) return WN_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int P = 1, k = upper->kernel_width;
    const int Ciu = upper->in_channels, Cou = upper->out_channels, Col = lower->out_channels, Ms = lower->skip_rows;
    const int ld = upper->ld;
    HCol2Args c;
    std::memset(&c, 0, sizeof(c));
    c.wstream1 = (const char*)packed_upper + pu.off_kb + pu.kb.slab_woff[0];
    c.wstream2 = (const char*)packed_lower + pl.off_kap + pl.ka.slab_woff[0];
    int kk = 0;
    auto add_seg = [&](const HView& v, int off, int nks) {
        for (int j = 0; j < nks && kk < kCol2MaxK; ++j, ++kk) {
            c.kbase[kk] = v.base + ((long long)off + 2LL * j * ld) * 16;
            c.kustride[kk] = v.ustride;
        }
    };
    const HView vda = view(da_upper, Cou, ld, P), vdg = view(dg_upper, Cou, ld, P);
    for (int j = 0; j < k; ++j) {
        add_seg(vda, -offu[j], pu.kb.seg_nks[2 * j]);
        add_seg(vdg, -offu[j], pu.kb.seg_nks[2 * j + 1]);
    }
    if (dr_upper) add_seg(view(dr_upper, Cou, ld, P), 0, pu.kb.seg_nks[2 * k]);
    add_seg(view(dskip, Ms, ld, P), 0, pl.ka.seg_nks[0]);
    c.nt = cp32(Ciu) / 32;
    c.hasdr = dr_upper ? 1 : 0;
    if (kk != (c.hasdr ? 10 : 8) * c.nt + 2 * c.nt) return WN_ERR_UNSUPPORTED;
    c.dx = dst_of(view(dx_upper, Ciu, ld, P));
    c.z = dst_of(view(z_lower, Col, ld, P)); c.sg = dst_of(view(sg_lower, Col, ld, P));
    c.da = dst_of(view(da_lower, Col, ld, P)); c.dg = dst_of(view(dg_lower, Col, ld, P));
    c.flag = overflow_flag;
    c.dump = (char*)packed_upper + pu.fu.off_bias + align256(4 * kFRows * sizeof(float));
    c.oscale1 = c.oscale2 = 1.0f / kWeightScale;
    c.B = upper->batch; c.L = upper->length; c.ld = ld; c.halo = upper->halo;
    c.nunit = (int)(((long long)upper->batch * upper->length + 31) / 32);
    const double BL = (double)upper->batch * upper->length;
    wn::ProfScopeShared prof(KC_HCOL_DXDZ, (2.0 * Ciu * (double)(2 * k * Cou + (dr_upper ? Cou : 0)) + 2.0 * Col * (double)(Ms + Col)) * BL, st);
    WN_HIP(launch_hcol2(precision, c, st), "hcol2<dx+dz>");
    return WN_OK;
}

int wn_hblock_backward_data_masked(const wn_block_shape* s, int precision, const void* packed, const void* dr, const void* dskip,
                                   const void* z, const void* sg, void* da, void* dg, void* dx, const void* x_act, float leaky_slope,
                                   unsigned* overflow_flag, wn_stream_t stream) {
    if (!dx || !x_act) return WN_ERR_NULL;
    return hblock_backward_data_impl(s, precision, packed, dr, dskip, z, sg, da, dg, dx, nullptr, nullptr, x_act, leaky_slope,
                                     overflow_flag, stream);
}

// ---- skips_sum of a whole stack ----------------------------------------------------------------------------------------
namespace {
int check_hskipsum(const wn_skipsum_shape* s, int prec) {
    if (!s) return WN_ERR_NULL;
    if (!half_prec(prec)) return WN_ERR_UNSUPPORTED;
    if (s->nblocks < 1 || s->skip_rows <= 0) return WN_ERR_BAD_SHAPE;
    if (s->nblocks > WN_MAX_STACK_GROUP || s->skip_rows > WN_MAX_CHANNELS) return WN_ERR_UNSUPPORTED;
    for (int l = 0; l < s->nblocks; ++l) {
        if (s->channels[l] <= 0) return WN_ERR_BAD_SHAPE;
        if (s->channels[l] > WN_MAX_CHANNELS) return WN_ERR_UNSUPPORTED;
    }
    return check_hlayout(s->batch, s->length, s->ld, s->halo, 0);
}
HPlan plan_hskipsum(const wn_skipsum_shape* s, int prec) {
    HPlan g;
    g.init(s->skip_rows, hp_planes(prec), true, s->length, true);
    g.nseg = s->nblocks;
    for (int l = 0; l < s->nblocks; ++l) g.seg_nks[l] = cp32(s->channels[l]) / 16;
    for (int r0 = 0; r0 < s->skip_rows; r0 += g.rows) g.add_slab(s->nblocks, r0);
    return g;
}
}  // namespace

// (+ the 1 KiB line that masked store lanes of the column-owner form write to)
static size_t hskipsum_bytes(const HPlan& g) { return g.bytes() + 1024; }
// skips_sum -> leaky series as hcol_kernel (wn_col_skip.hip): one-plane modes, <= 16 blocks of one padded width (64 or 128
// channels), skip rows of the same padded width
static bool skipsum_col(const wn_skipsum_shape* s, int precision, const HPlan& g) {
    if (!col_backward_enabled() || hp_planes(precision) != 1 || g.nslab != 1 || g.MT != 2 || g.k32) return false;
    const int w = cp32(s->skip_rows);
    if ((w != 64 && w != 128) || s->nblocks > 16) return false;
    for (int l = 0; l < s->nblocks; ++l) if (cp32(s->channels[l]) != w) return false;
    return true;
}

size_t wn_hskipsum_packed_bytes(const wn_skipsum_shape* s, int precision) {
    if (check_hskipsum(s, precision) != WN_OK) return 0;
    return hskipsum_bytes(plan_hskipsum(s, precision));
}

int wn_hskipsum_pack(const wn_skipsum_shape* s, int precision, const float* const* w_skip, const float* bias_total, void* packed,
                     wn_stream_t stream) {
    int rc = check_hskipsum(s, precision);
    if (rc != WN_OK) return rc;
    if (!w_skip || !packed) return WN_ERR_NULL;
    for (int l = 0; l < s->nblocks; ++l) if (!w_skip[l]) return WN_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    const HPlan g = plan_hskipsum(s, precision);
    wn::ProfScopeShared prof(KC_PACK, 0.0, st);
    HPackArgs a;
    fill_hpack(a, g, packed, 0, precision);
    for (int l = 0; l < s->nblocks; ++l) a.set[0].seg[l] = hsrc(w_skip[l], s->skip_rows, s->channels[l], s->channels[l], 1, kWeightScale);
    a.set[0].bias0 = bias_total; a.set[0].bias_rows = s->skip_rows;
    plain_tiles(a, g, s->skip_rows);
    WN_HIP(launch_hpack(a, st), "hpack(skipsum)");
    return WN_OK;
}

int wn_hskipsum_forward(const wn_skipsum_shape* s, int precision, const void* packed, const void* const* z, float* skip_dense,
                        int accumulate, wn_stream_t stream) {
    int rc = check_hskipsum(s, precision);
    if (rc != WN_OK) return rc;
    if (!packed || !z || !skip_dense) return WN_ERR_NULL;
    for (int l = 0; l < s->nblocks; ++l) if (!z[l]) return WN_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    const HPlan g = plan_hskipsum(s, precision);
    const int P = hp_planes(precision);
    HGemmArgs a;
    fill_hgemm(a, g, packed, 0, s->batch, s->length, s->ld, s->halo);
    double ksum = 0;
    for (int l = 0; l < s->nblocks; ++l) { set_hseg(a, l, view(z[l], s->channels[l], s->ld, P), 0, g.seg_nks[l]); ksum += s->channels[l]; }
    a.out32 = skip_dense; a.out32_rows = s->skip_rows; a.out32_accum = accumulate ? 1 : 0;
    wn::ProfScopeShared prof(KC_HSKIP, 2.0 * s->skip_rows * ksum * (double)s->batch * s->length, st);
    WN_HIP(launch_hgemm(precision, g.kernel(), HEPI_F32, a, st), "hgemm<skipsum>");
    return WN_OK;
}

// skips_sum straight into an activated half series (no dense fp32 S): out = leaky(S) * out_scale, the input of the output block's
// first 1x1 conv (modules/wavenet.py:67-71,103: LeakyReLU, Conv1d, LeakyReLU, Conv1d) when that block stays in the series layout
int wn_hskipsum_forward_series(const wn_skipsum_shape* s, int precision, const void* packed, const void* const* z, void* out_series,
                               float out_scale, float leaky_slope, unsigned* overflow_flag, wn_stream_t stream) {
    int rc = check_hskipsum(s, precision);
    if (rc != WN_OK) return rc;
    if (!packed || !z || !out_series) return WN_ERR_NULL;
    if (!(out_scale > 0.0f)) return WN_ERR_BAD_SHAPE;
    for (int l = 0; l < s->nblocks; ++l) if (!z[l]) return WN_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    const HPlan g = plan_hskipsum(s, precision);
    const int P = hp_planes(precision);
    HGemmArgs a;
    fill_hgemm(a, g, packed, 0, s->batch, s->length, s->ld, s->halo);
    double ksum = 0;
    for (int l = 0; l < s->nblocks; ++l) { set_hseg(a, l, view(z[l], s->channels[l], s->ld, P), 0, g.seg_nks[l]); ksum += s->channels[l]; }
    a.dst[0] = dst_of(view(out_series, s->skip_rows, s->ld, P));
    a.oscale2 = out_scale; a.leaky = leaky_slope; a.flag = overflow_flag;
    if (skipsum_col(s, precision, g)) {
        HColArgs c;
        col_args_common(c, a, s->skip_rows, (char*)packed + g.bytes(), s->batch, s->length, s->ld, s->halo);
        c.dst = a.dst[0]; c.bias = a.bias; c.oscale2 = out_scale; c.leaky = leaky_slope;
        wn::ProfScopeShared prof(KC_HCOL_SKIP, 2.0 * s->skip_rows * ksum * (double)s->batch * s->length, st);
        WN_HIP(launch_hcol_skipsum(precision, c, st), "hcol<skipsum series>");
        return WN_OK;
    }
    wn::ProfScopeShared prof(KC_HSKIP, 2.0 * s->skip_rows * ksum * (double)s->batch * s->length, st);
    WN_HIP(launch_hgemm(precision, g.kernel(), HEPI_LEAKY, a, st), "hgemm<skipsum series>");
    return WN_OK;
}

// ---- every pack job of a stack in one launch -------------------------------------------------------------------------------
// Training repacks every weight after each optimizer step: five small launches per block plus the long-K skips_sum pack.
// Their arguments depend only on shapes, precision and pointers, so they are built once into a TABLE that lives on the device
// and one kernel runs all of them (hpack_table_kernel).  Destinations are offsets into ONE packed buffer of the whole stack
// (a fresh allocation per forward costs no rebuild); sources inside a caller-declared "dynamic" range (tensors re-allocated
// every step: the folded skip weights) are stored as offsets from that range's base, which the launch supplies.
namespace {
constexpr int kMaxDynamic = 3;
int stack_jobs_upper_bound(int nblocks) { return 6 * nblocks + cdiv(nblocks, WN_MAX_STACK_GROUP); }   // (a fused-forward block has up to 6: gate, res, skip, dz, dz for the paired launch, dx)

void encode_dynamic(const float*& ptr, int& flags, const wn_mem_range* dyn, int ndyn) {
    if (!ptr) return;
    for (int i = 0; i < ndyn; ++i) {
        const char* b = (const char*)dyn[i].base;
        const char* q = (const char*)ptr;
        if (b && q >= b && q < b + dyn[i].bytes) {
            ptr = reinterpret_cast<const float*>((uintptr_t)(q - b));
            flags |= (i + 1);
            return;
        }
    }
}
}  // namespace

size_t wn_hstack_pack_table_bytes(int nblocks) {
    if (nblocks <= 0) return 0;
    const size_t nj = (size_t)stack_jobs_upper_bound(nblocks);
    return align256(nj * sizeof(HPackArgs)) + align256((nj + 1) * sizeof(int));
}

int wn_hstack_pack_table_build(const wn_block_shape* shapes, const wn_block_params* params, int nblocks, int precision,
                               int with_skipsum, const wn_mem_range* dynamic, int ndynamic, void* table_host, size_t table_bytes,
                               size_t* block_offsets, size_t* skipsum_offsets, size_t* packed_total, int* njobs_out,
                               int* launch_blocks_out) {
    if (!shapes || !params || !table_host || !block_offsets || !packed_total || !njobs_out || !launch_blocks_out) return WN_ERR_NULL;
    if (nblocks <= 0 || ndynamic < 0 || ndynamic > kMaxDynamic || (ndynamic && !dynamic)) return WN_ERR_BAD_SHAPE;
    if (!half_prec(precision)) return WN_ERR_UNSUPPORTED;
    if (table_bytes < wn_hstack_pack_table_bytes(nblocks)) return WN_ERR_WORKSPACE;
    std::vector<HPackArgs> jobs;
    size_t total = 256;                 // (offsets travel as pointers through the job builders: 0 would read as NULL)
    for (int l = 0; l < nblocks; ++l) {
        int off[WN_MAX_TAPS];
        int rc = check_hblock(&shapes[l], precision, off);
        if (rc != WN_OK) return rc;
        block_offsets[l] = total;
        rc = fill_hblock_jobs(&shapes[l], precision, &params[l], reinterpret_cast<void*>(total), jobs);
        if (rc != WN_OK) return rc;
        total += align256(plan_hblock(&shapes[l], precision).total);
    }
    if (with_skipsum) {
        if (!skipsum_offsets) return WN_ERR_NULL;
        // the biases of ALL blocks are summed into the first group's bias: they must be equally spaced (one stacked tensor)
        long long bstride = 0;
        for (int l = 1; l < nblocks; ++l) {
            const long long d = params[l].b_skip - params[l - 1].b_skip;
            if (l == 1) bstride = d;
            else if (d != bstride) return WN_ERR_UNSUPPORTED;
        }
        if (bstride < 0 || bstride > 0x7fffffffLL) return WN_ERR_UNSUPPORTED;
        for (int g0 = 0, gi = 0; g0 < nblocks; g0 += WN_MAX_STACK_GROUP, ++gi) {
            wn_skipsum_shape ss;
            std::memset(&ss, 0, sizeof(ss));
            const int m = std::min(WN_MAX_STACK_GROUP, nblocks - g0);
            ss.batch = shapes[0].batch; ss.length = shapes[0].length; ss.skip_rows = shapes[0].skip_rows; ss.nblocks = m;
            ss.ld = shapes[0].ld; ss.halo = shapes[0].halo;
            for (int i = 0; i < m; ++i) {
                if (shapes[g0 + i].skip_rows != ss.skip_rows) return WN_ERR_BAD_SHAPE;
                ss.channels[i] = shapes[g0 + i].out_channels;
            }
            int rc = check_hskipsum(&ss, precision);
            if (rc != WN_OK) return rc;
            const HPlan g = plan_hskipsum(&ss, precision);
            HPackArgs a;
            fill_hpack(a, g, reinterpret_cast<void*>(total), 0, precision);
            for (int i = 0; i < m; ++i)
                a.set[0].seg[i] = hsrc(params[g0 + i].w_skip, ss.skip_rows, ss.channels[i], ss.channels[i], 1, kWeightScale);
            if (g0 == 0) {
                a.set[0].bias0 = params[0].b_skip; a.set[0].bias0_rep = nblocks; a.set[0].bias0_stride = (int)bstride;
            }
            a.set[0].bias_rows = ss.skip_rows;
            plain_tiles(a, g, ss.skip_rows);
            jobs.push_back(a);
            skipsum_offsets[gi] = total;
            total += align256(hskipsum_bytes(g));
        }
    }
    const int nj = (int)jobs.size();
    if (nj > stack_jobs_upper_bound(nblocks)) return WN_ERR_WORKSPACE;
    int* block0 = reinterpret_cast<int*>((char*)table_host + align256((size_t)stack_jobs_upper_bound(nblocks) * sizeof(HPackArgs)));
    long long nb = 0;
    for (int j = 0; j < nj; ++j) {
        HPackArgs& a = jobs[j];
        for (int st = 0; st < 2; ++st) {
            for (int sg = 0; sg < kMaxSeg; ++sg) encode_dynamic(a.set[st].seg[sg].ptr, a.set[st].seg[sg].flags, dynamic, ndynamic);
            encode_dynamic(a.set[st].bias0, a.set[st].bias0_dyn, dynamic, ndynamic);
            encode_dynamic(a.set[st].bias1, a.set[st].bias1_dyn, dynamic, ndynamic);
        }
        if (a.bias) a.bias = reinterpret_cast<float*>(reinterpret_cast<uintptr_t>(a.bias) + 1);   // offset + 1: 0 stays "no bias"
        block0[j] = (int)nb;
        nb += (hpack_threads(a) + 255) / 256;
        if (nb > 0x7fffffffLL) return WN_ERR_UNSUPPORTED;
    }
    block0[nj] = (int)nb;
    std::memcpy(table_host, jobs.data(), (size_t)nj * sizeof(HPackArgs));
    *packed_total = total;
    *njobs_out = nj;
    *launch_blocks_out = (int)nb;
    return WN_OK;
}

int wn_hstack_pack_run(const void* table_dev, int nblocks, int njobs, int launch_blocks, const void* const* dynamic_bases, int ndynamic,
                       void* packed, unsigned* overflow_flag, wn_stream_t stream) {
    if (!table_dev || !packed) return WN_ERR_NULL;
    if (nblocks <= 0 || njobs <= 0 || njobs > stack_jobs_upper_bound(nblocks) || launch_blocks <= 0 || ndynamic < 0 ||
        ndynamic > kMaxDynamic || (ndynamic && !dynamic_bases))
        return WN_ERR_BAD_SHAPE;
    HPackDyn d;
    std::memset(&d, 0, sizeof(d));
    for (int i = 0; i < ndynamic; ++i) d.base[i] = (const char*)dynamic_bases[i];
    d.out = (char*)packed;
    d.flag = overflow_flag;
    const HPackArgs* jobs = reinterpret_cast<const HPackArgs*>(table_dev);
    const int* block0 = reinterpret_cast<const int*>((const char*)table_dev + align256((size_t)stack_jobs_upper_bound(nblocks) * sizeof(HPackArgs)));
    hipStream_t st = (hipStream_t)stream;
    wn::ProfScopeShared prof(KC_PACK, 0.0, st);
    WN_HIP(launch_hpack_table(jobs, block0, njobs, launch_blocks, d, st), "hpack(table)");
    return WN_OK;
}

// ---- weight gradients ------------------------------------------------------------------------------------------------------
namespace {
// One workgroup tile of hwgrad_kernel is 256 x 256 (A rows x B rows) and is staged as two 128-channel halves per operand.
// A pair spec names the tensors behind the halves and the destinations of the 128 x 128 quadrants (composite pairs), or one
// tensor per operand and one destination covering the whole matrix (ordinary pairs).
struct HHalf { const void* p; int rows; int off; };
struct HQuad { int ah, bh; float post; float* w; int sm, sn; float* b0; float* b1; };   // b0 / b1: destinations of the row sums of A half ah
struct HPairSpec {
    HHalf a[2], b[2];
    bool composite;
    int rowsum;
    std::vector<HQuad> q;
};

HPairSpec plain_pair(const void* A, int a_rows, const void* Bm, int b_rows, int off, int rowsum, float post, float* w, int sm, int sn,
                     float* b0, float* b1) {
    HPairSpec s;
    s.a[0] = s.a[1] = HHalf{A, a_rows, 0};
    s.b[0] = s.b[1] = HHalf{Bm, b_rows, off};
    s.composite = false;
    s.rowsum = rowsum;
    s.q.push_back(HQuad{0, 0, post, w, sm, sn, rowsum ? b0 : nullptr, rowsum ? b1 : nullptr});
    return s;
}

struct HWPlan {
    int npair = 0, ntile_total = 0, nsplit = 1;
    int Mp[kMaxPair], Np[kMaxPair], mt[kMaxPair], nt[kMaxPair], tile0[kMaxPair], rs_off[kMaxPair];
    long long slab_off[kMaxPair];
    long long slab_floats = 0;
    int rs_floats = 0;
    bool composite = false;
    void add(int M, int N) {
        const int T = 256, i = npair++;
        mt[i] = cdiv(M, T); nt[i] = cdiv(N, T);
        Mp[i] = mt[i] * T; Np[i] = nt[i] * T;
        tile0[i] = ntile_total; ntile_total += mt[i] * nt[i];
        slab_off[i] = slab_floats; slab_floats += (long long)Mp[i] * Np[i];
        rs_off[i] = rs_floats; rs_floats += Mp[i];
    }
    void finish(int nstep) {
        // ordinary pairs: two rounds of workgroups on the 256 CUs.  Composite pairs (2-3 tiles per block) are short: every
        // workgroup ends with a 256 KiB partial tile that the reduction re-reads, so ONE round and half the partial slabs
        const int target = composite ? 256 : 512;
        nsplit = std::max(1, target / std::max(1, ntile_total));
        nsplit = std::min(nsplit, std::max(1, nstep / 8));      // at least a few k-steps per split
        if (nsplit >= 16 && !(composite && ntile_total > 3)) nsplit = nsplit / 8 * 8;   // (XCD-aware placement needs a multiple of 8)
    }
    bool xcd_map() const { return nsplit % 8 == 0; }
    size_t bytes() const { return (size_t)nsplit * (size_t)(slab_floats + rs_floats) * 4; }
};

bool composite_wgrad(int ci, int co, int ms) {
    const char* e = getenv("WN_HWGRAD_COMPOSITE");   // read per call (tests compare both forms)
    return !(e && atoi(e) == 0) && cp32(ci) <= 128 && cp32(co) <= 128 && cp32(ms) <= 128;
}

std::vector<HPairSpec> hblock_pairs(const wn_block_shape* s, const int* off, const void* x, const void* z, const void* da,
                                    const void* dg, const void* dr, const void* dskip, const wn_block_params* g) {
    const int Ci = s->in_channels, Co = s->out_channels, Ms = s->skip_rows, k = s->kernel_width;
    const float inv_rs = 1.0f / kResidualScale;
    std::vector<HPairSpec> ps;
    if (composite_wgrad(Ci, Co, Ms)) {
        // [da ; dg] x [x(t + off_j) ; x(t + off_j+1)]: the four tap gradients of two taps in one tile
        for (int j = 0; j < k; j += 2) {
            const bool two = j + 1 < k;
            HPairSpec p;
            p.composite = true;
            p.rowsum = j == 0;
            p.a[0] = HHalf{da, Co, 0}; p.a[1] = HHalf{dg, Co, 0};
            p.b[0] = HHalf{x, Ci, off[j]}; p.b[1] = HHalf{x, Ci, two ? off[j + 1] : off[j]};
            p.q.push_back(HQuad{0, 0, inv_rs, g ? g->w_tanh + j : nullptr, Ci * k, k, (g && j == 0) ? g->b_tanh : nullptr, nullptr});
            p.q.push_back(HQuad{1, 0, inv_rs, g ? g->w_sigmoid + j : nullptr, Ci * k, k, (g && j == 0) ? g->b_sigmoid : nullptr, nullptr});
            if (two) {
                p.q.push_back(HQuad{0, 1, inv_rs, g ? g->w_tanh + j + 1 : nullptr, Ci * k, k, nullptr, nullptr});
                p.q.push_back(HQuad{1, 1, inv_rs, g ? g->w_sigmoid + j + 1 : nullptr, Ci * k, k, nullptr, nullptr});
            }
            ps.push_back(p);
        }
        // [dskip ; dr] x [z ; x]: skip, residual and projection gradients (the dskip x quadrant is not computed)
        HPairSpec p;
        p.composite = true;
        p.rowsum = 1;
        p.a[0] = HHalf{dskip, Ms, 0}; p.a[1] = dr ? HHalf{dr, Co, 0} : p.a[0];
        p.b[0] = HHalf{z, Co, 0}; p.b[1] = dr ? HHalf{x, Ci, 0} : p.b[0];
        p.q.push_back(HQuad{0, 0, 1.0f, g ? g->w_skip : nullptr, Co, 1, g ? g->b_skip : nullptr, nullptr});
        if (dr) {
            p.q.push_back(HQuad{1, 0, 1.0f, g ? g->w_res : nullptr, Co, 1, g ? g->b_res : nullptr, g ? g->b_proj : nullptr});
            p.q.push_back(HQuad{1, 1, inv_rs, g ? g->w_proj : nullptr, Ci, 1, nullptr, nullptr});
        }
        ps.push_back(p);
        return ps;
    }
    for (int j = 0; j < k; ++j) {
        ps.push_back(plain_pair(da, Co, x, Ci, off[j], j == 0, inv_rs, g ? g->w_tanh + j : nullptr, Ci * k, k, g ? g->b_tanh : nullptr, nullptr));
        ps.push_back(plain_pair(dg, Co, x, Ci, off[j], j == 0, inv_rs, g ? g->w_sigmoid + j : nullptr, Ci * k, k, g ? g->b_sigmoid : nullptr, nullptr));
    }
    ps.push_back(plain_pair(dskip, Ms, z, Co, 0, 1, 1.0f, g ? g->w_skip : nullptr, Co, 1, g ? g->b_skip : nullptr, nullptr));
    if (dr) {
        ps.push_back(plain_pair(dr, Co, z, Co, 0, 1, 1.0f, g ? g->w_res : nullptr, Co, 1, g ? g->b_res : nullptr, g ? g->b_proj : nullptr));
        ps.push_back(plain_pair(dr, Co, x, Ci, 0, 0, inv_rs, g ? g->w_proj : nullptr, Ci, 1, nullptr, nullptr));
    }
    return ps;
}

int run_hwgrad(const std::vector<HPairSpec>& ps, int prec, int B, int L, int ld, int halo, const float* dyn_inv, void* workspace,
               size_t workspace_bytes, bool dry, size_t* need, hipStream_t st) {
    HWPlan wp;
    int ndst = 0;
    for (const HPairSpec& p : ps) {
        if (p.composite) { wp.add(256, 256); wp.composite = true; }
        else wp.add(p.a[0].rows, p.b[0].rows);
        ndst += (int)p.q.size();
    }
    if (wp.npair > kMaxPair || ndst > kMaxReduceDst) return WN_ERR_UNSUPPORTED;
    const int spr = cdiv(L, 32);                      // stages of 32 time steps per utterance (hwgrad_kernel)
    wp.finish(B * spr);
    if (need) *need = wp.bytes();
    if (dry || ps.empty()) return WN_OK;
    if (!workspace) return WN_ERR_NULL;
    if (workspace_bytes < wp.bytes() || (reinterpret_cast<uintptr_t>(workspace) & 15)) return WN_ERR_WORKSPACE;
    const int P = hp_planes(prec);
    HWgradArgs a;
    std::memset(&a, 0, sizeof(a));
    ReduceArgs r;
    std::memset(&r, 0, sizeof(r));
    double flops = 0;
    int nd = 0;
    for (int i = 0; i < wp.npair; ++i) {
        HWgradPair& q = a.pair[i];
        const HPairSpec& p = ps[i];
        for (int hf = 0; hf < 2; ++hf) {
            // the kernel stages whole 128-channel halves: the operands' padded channel counts (multiples of 32) bound the groups read
            const HView va = view(p.a[hf].p, p.a[hf].rows, ld, P), vb = view(p.b[hf].p, p.b[hf].rows, ld, P);
            q.A[hf] = va.base; q.a_ustride[hf] = va.ustride; q.a_pstride[hf] = va.pstride; q.a_groups[hf] = va.cp / 8;
            q.Bm[hf] = vb.base; q.b_ustride[hf] = vb.ustride; q.b_pstride[hf] = vb.pstride; q.b_groups[hf] = vb.cp / 8;
            q.a_gb[hf] = q.b_gb[hf] = p.composite ? 0 : 16 * hf;
            q.off[hf] = p.b[hf].off;
        }
        q.mt = wp.mt[i]; q.nt = wp.nt[i]; q.tile0 = wp.tile0[i];
        q.slab_off = wp.slab_off[i]; q.Mp = wp.Mp[i]; q.Np = wp.Np[i];
        q.rowsum = p.rowsum; q.rs_off = wp.rs_off[i];
        q.quad_mask = p.composite ? 0 : 15;
        for (const HQuad& qd : p.q) {
            if (p.composite) q.quad_mask |= 1 << (2 * qd.ah + qd.bh);
            ReduceDst& d = r.d[nd++];
            d.w = qd.w; d.M = p.a[qd.ah].rows; d.N = p.b[qd.bh].rows; d.sm = qd.sm; d.sn = qd.sn;
            d.slab_off = wp.slab_off[i] + (p.composite ? (long long)(128 * qd.ah) * wp.Np[i] + 128 * qd.bh : 0);
            d.Np = wp.Np[i];
            d.b0 = qd.b0; d.b1 = qd.b1; d.rs_off = wp.rs_off[i] + (p.composite ? 128 * qd.ah : 0);
            d.post = qd.post;
            flops += 2.0 * d.M * (double)d.N * (double)B * L;
        }
    }
    a.npair = wp.npair; a.ntile_total = wp.ntile_total; a.nsplit = wp.nsplit; a.xcd_map = wp.xcd_map() ? 1 : 0;
    a.B = B; a.L = L; a.ld = ld; a.halo = halo; a.steps_per_row = spr; a.nstep = B * spr;
    a.slab = reinterpret_cast<float*>(workspace);
    a.rowsum = a.slab + (size_t)wp.nsplit * wp.slab_floats;
    a.slab_floats = wp.slab_floats; a.rs_floats = wp.rs_floats;
    r.npair = nd; r.nsplit = wp.nsplit; r.slab = a.slab; r.rowsum = a.rowsum;
    r.slab_floats = wp.slab_floats; r.rs_floats = wp.rs_floats; r.dyn_inv = dyn_inv;
    {
        wn::ProfScopeShared prof(KC_HWGRAD, flops, st);
        WN_HIP(launch_hwgrad(prec, a, st), "hwgrad");
    }
    {
        wn::ProfScopeShared prof(KC_WGRAD_REDUCE, 0.0, st);
        WN_HIP(launch_wgrad_reduce(r, st), "wgrad_reduce");
    }
    return WN_OK;
}
}  // namespace

size_t wn_hblock_wgrad_workspace_bytes(const wn_block_shape* s, int precision) {
    int off[WN_MAX_TAPS];
    if (check_hblock(s, precision, off) != WN_OK) return 0;
    static const float dummy = 0;
    size_t need = 0;
    for (int with_dr = 0; with_dr < 2; ++with_dr) {
        std::vector<HPairSpec> ps = hblock_pairs(s, off, &dummy, &dummy, &dummy, &dummy, with_dr ? &dummy : nullptr, &dummy, nullptr);
        size_t n = 0;
        run_hwgrad(ps, precision, s->batch, s->length, s->ld, s->halo, nullptr, nullptr, 0, true, &n, nullptr);
        need = std::max(need, n);
    }
    return need;
}

int wn_hblock_backward_weights(const wn_block_shape* s, int precision, const void* x, const void* z, const void* da,
                               const void* dg, const void* dr, const void* dskip, const wn_block_params* grads,
                               const float* dyn_inv_scale, void* workspace, size_t workspace_bytes, wn_stream_t stream) {
    int off[WN_MAX_TAPS];
    int rc = check_hblock(s, precision, off);
    if (rc != WN_OK) return rc;
    if (!x || !z || !da || !dg || !dskip || !grads) return WN_ERR_NULL;
    if (!grads->w_tanh || !grads->b_tanh || !grads->w_sigmoid || !grads->b_sigmoid || !grads->w_skip || !grads->b_skip)
        return WN_ERR_NULL;
    if (dr && (!grads->w_res || !grads->b_res || !grads->w_proj || !grads->b_proj)) return WN_ERR_NULL;
    // the 256-channel staging tiles must lie inside the operands: channel counts are padded to 32, tiles to 256
    std::vector<HPairSpec> ps = hblock_pairs(s, off, x, z, da, dg, dr, dskip, grads);
    return run_hwgrad(ps, precision, s->batch, s->length, s->ld, s->halo, dyn_inv_scale, workspace, workspace_bytes, false,
                      nullptr, (hipStream_t)stream);
}

// ---- the weight gradients of SEVERAL blocks in one launch -----------------------------------------------------------------------
// Small blocks (<= 128 channels: composite pairs, two or three 256 x 256 tiles per block) leave a split-K launch per block with
// 128 splits of 32 stages each and 67 MB of partial slabs that the reduction re-reads -- per block.  Their operands (x, z, da,
// dg, dr, dskip) can simply be kept until several blocks have run backward_data: one launch then holds all their tiles, the
// split count falls to ~256 / tiles (long K loops), and the partial slabs of the whole group are what ONE block wrote before.
namespace {
constexpr int kMaxGroupBlocks = 8;     // 2 pairs and <= 7 destinations per block: kMaxPair / kMaxReduceDst
}

int wn_hblocks_wgrad_group_max(const wn_block_shape* s, int precision) {
    int off[WN_MAX_TAPS];
    if (check_hblock(s, precision, off) != WN_OK) return 0;
    if (!composite_wgrad(s->in_channels, s->out_channels, s->skip_rows)) return 1;
    const int pairs = (s->kernel_width + 1) / 2 + 1, dsts = 2 * s->kernel_width + 3;
    return std::max(1, std::min(kMaxGroupBlocks, std::min(kMaxPair / pairs, kMaxReduceDst / dsts)));
}

size_t wn_hblocks_wgrad_workspace_bytes(const wn_block_shape* shapes, int nblocks, int precision) {
    if (!shapes || nblocks <= 0) return 0;
    static const float dummy = 0;
    std::vector<HPairSpec> all;
    for (int l = 0; l < nblocks; ++l) {
        int off[WN_MAX_TAPS];
        if (check_hblock(&shapes[l], precision, off) != WN_OK) return 0;
        if (shapes[l].batch != shapes[0].batch || shapes[l].length != shapes[0].length || shapes[l].ld != shapes[0].ld ||
            shapes[l].halo != shapes[0].halo)
            return 0;
        std::vector<HPairSpec> ps = hblock_pairs(&shapes[l], off, &dummy, &dummy, &dummy, &dummy, &dummy, &dummy, nullptr);
        all.insert(all.end(), ps.begin(), ps.end());
    }
    size_t need = 0;
    if (run_hwgrad(all, precision, shapes[0].batch, shapes[0].length, shapes[0].ld, shapes[0].halo, nullptr, nullptr, 0, true, &need,
                   nullptr) != WN_OK)
        return 0;
    return need;
}

int wn_hblocks_backward_weights(const wn_block_shape* shapes, int nblocks, int precision, const void* const* x, const void* const* z,
                                const void* const* da, const void* const* dg, const void* const* dr, const void* const* dskip,
                                const wn_block_params* grads, const float* dyn_inv_scale, void* workspace, size_t workspace_bytes,
                                wn_stream_t stream) {
    if (!shapes || !x || !z || !da || !dg || !dr || !dskip || !grads) return WN_ERR_NULL;
    if (nblocks <= 0) return WN_ERR_BAD_SHAPE;
    std::vector<HPairSpec> all;
    for (int l = 0; l < nblocks; ++l) {
        int off[WN_MAX_TAPS];
        int rc = check_hblock(&shapes[l], precision, off);
        if (rc != WN_OK) return rc;
        if (shapes[l].batch != shapes[0].batch || shapes[l].length != shapes[0].length || shapes[l].ld != shapes[0].ld ||
            shapes[l].halo != shapes[0].halo)
            return WN_ERR_BAD_SHAPE;                            // one series geometry per launch
        const wn_block_params* g = &grads[l];
        if (!x[l] || !z[l] || !da[l] || !dg[l] || !dskip[l]) return WN_ERR_NULL;
        if (!g->w_tanh || !g->b_tanh || !g->w_sigmoid || !g->b_sigmoid || !g->w_skip || !g->b_skip) return WN_ERR_NULL;
        if (dr[l] && (!g->w_res || !g->b_res || !g->w_proj || !g->b_proj)) return WN_ERR_NULL;
        std::vector<HPairSpec> ps = hblock_pairs(&shapes[l], off, x[l], z[l], da[l], dg[l], dr[l], dskip[l], g);
        all.insert(all.end(), ps.begin(), ps.end());
    }
    return run_hwgrad(all, precision, shapes[0].batch, shapes[0].length, shapes[0].ld, shapes[0].halo, dyn_inv_scale, workspace,
                      workspace_bytes, false, nullptr, (hipStream_t)stream);
}

// ==========================================================================================================================
// stand-alone dilated conv in the half-precision modes (the entry conv and the 1x1 convs of the output stacks, so that a
// model in a half mode runs all of its convolutions on the half kernels): wn_hconv_*, the half-series counterpart of wn_conv_*
// ==========================================================================================================================
namespace {
int check_hconv(const wn_conv_shape* s, int prec, int* off) {
    if (!s) return WN_ERR_NULL;
    if (!half_prec(prec)) return WN_ERR_UNSUPPORTED;
    if (s->in_channels <= 0 || s->out_channels <= 0 || s->dilation <= 0 || s->kernel_width < 1) return WN_ERR_BAD_SHAPE;
    if (s->kernel_width > WN_MAX_TAPS || s->in_channels > WN_MAX_CHANNELS || s->out_channels > WN_MAX_CHANNELS) return WN_ERR_UNSUPPORTED;
    wn_tap_offsets(s->kernel_width, s->dilation, s->causal, off);
    int mx = 0;
    for (int j = 0; j < s->kernel_width; ++j) mx = std::max(mx, std::abs(off[j]));
    return check_hlayout(s->batch, s->length, s->ld, s->halo, mx);
}
struct HConvPlan {
    HPlan f, kb; size_t off_f = 0, off_kb = 0, off_dump = 0, total = 0;
    bool col = false;    // the series-to-series forms of a 1x1 conv of <= 128 channels run as hcol_kernel (wn_col_conv.hip)
};
HConvPlan plan_hconv(const wn_conv_shape* s, int prec) {
    HConvPlan p;
    const int P = hp_planes(prec), Ci = s->in_channels, Co = s->out_channels, k = s->kernel_width;
    p.col = col_backward_enabled() && P == 1 && k == 1 && cp32(Ci) <= 128 && cp32(Co) <= 128;
    p.f.init(Co, P);
    p.f.nseg = k;
    for (int j = 0; j < k; ++j) p.f.seg_nks[j] = cp32(Ci) / 16;
    for (int r0 = 0; r0 < Co; r0 += p.f.rows) p.f.add_slab(k, r0);
    p.kb.init(Ci, P);
    p.kb.nseg = k;
    for (int j = 0; j < k; ++j) p.kb.seg_nks[j] = cp32(Co) / 16;
    for (int r0 = 0; r0 < Ci; r0 += p.kb.rows) p.kb.add_slab(k, r0);
    p.off_f = 0;
    p.off_kb = p.f.bytes();
    p.off_dump = p.off_kb + p.kb.bytes();
    p.total = p.off_dump + 1024;                     // the line that masked store lanes of hcol_kernel write to
    return p;
}
}  // namespace

size_t wn_hconv_packed_bytes(const wn_conv_shape* s, int precision) {
    int off[WN_MAX_TAPS];
    if (check_hconv(s, precision, off) != WN_OK) return 0;
    return plan_hconv(s, precision).total;
}

int wn_hconv_pack(const wn_conv_shape* s, int precision, const float* weight, const float* bias, float input_scale, void* packed,
                  wn_stream_t stream) {
    int off[WN_MAX_TAPS];
    int rc = check_hconv(s, precision, off);
    if (rc != WN_OK) return rc;
    if (!weight || !packed) return WN_ERR_NULL;
    if (!(input_scale > 0.0f)) return WN_ERR_BAD_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const HConvPlan cp = plan_hconv(s, precision);
    const int Ci = s->in_channels, Co = s->out_channels, k = s->kernel_width;
    wn::ProfScopeShared prof(KC_PACK, 0.0, st);
    HPackArgs a;
    {   // forward: rows = output channel, taps of x (stored as x * input_scale)
        fill_hpack(a, cp.f, packed, cp.off_f, precision);
        for (int j = 0; j < k; ++j) a.set[0].seg[j] = hsrc(weight + j, Co, Ci, Ci * k, k, kWeightScale / input_scale);
        a.set[0].bias0 = bias; a.set[0].bias_rows = bias ? Co : 0;
        plain_tiles(a, cp.f, Co);
        WN_HIP(launch_hpack(a, st), "hpack(conv)");
    }
    {   // backward data: rows = input channel, taps of dy
        fill_hpack(a, cp.kb, packed, cp.off_kb, precision);
        for (int j = 0; j < k; ++j) a.set[0].seg[j] = hsrc(weight + j, Ci, Co, k, Ci * k, kWeightScale);
        plain_tiles(a, cp.kb, Ci);
        a.bias = nullptr;
        WN_HIP(launch_hpack(a, st), "hpack(conv dx)");
    }
    return WN_OK;
}

int wn_hconv_forward(const wn_conv_shape* s, int precision, const void* packed, const void* x, float* y_dense,
                     wn_stream_t stream) {
    int off[WN_MAX_TAPS];
    int rc = check_hconv(s, precision, off);
    if (rc != WN_OK) return rc;
    if (!packed || !x || !y_dense) return WN_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    const HConvPlan cp = plan_hconv(s, precision);
    const int P = hp_planes(precision), Ci = s->in_channels, Co = s->out_channels, k = s->kernel_width;
    HGemmArgs a;
    fill_hgemm(a, cp.f, packed, cp.off_f, s->batch, s->length, s->ld, s->halo);
    const HView vx = view(x, Ci, s->ld, P);
    for (int j = 0; j < k; ++j) set_hseg(a, j, vx, off[j], cp.f.seg_nks[j]);
    a.out32 = y_dense; a.out32_rows = Co; a.out32_accum = 0;
    wn::ProfScopeShared prof(KC_HCONV_FWD, 2.0 * Co * (double)(k * Ci) * (double)s->batch * s->length, st);
    WN_HIP(launch_hgemm(precision, cp.f.kernel(), HEPI_F32, a, st), "hgemm<conv>");
    return WN_OK;
}

int wn_hconv_backward_data(const wn_conv_shape* s, int precision, const void* packed, const void* dy, float* dx_dense,
                           const float* dyn_inv_scale, wn_stream_t stream) {
    int off[WN_MAX_TAPS];
    int rc = check_hconv(s, precision, off);
    if (rc != WN_OK) return rc;
    if (!packed || !dy || !dx_dense) return WN_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    const HConvPlan cp = plan_hconv(s, precision);
    const int P = hp_planes(precision), Ci = s->in_channels, Co = s->out_channels, k = s->kernel_width;
    HGemmArgs a;
    fill_hgemm(a, cp.kb, packed, cp.off_kb, s->batch, s->length, s->ld, s->halo);
    a.bias = nullptr;
    const HView vdy = view(dy, Co, s->ld, P);
    for (int j = 0; j < k; ++j) set_hseg(a, j, vdy, -off[j], cp.kb.seg_nks[j]);
    a.out32 = dx_dense; a.out32_rows = Ci; a.out32_accum = 0; a.dyn_inv = dyn_inv_scale;
    wn::ProfScopeShared prof(KC_HCONV_BWD_DATA, 2.0 * Ci * (double)(k * Co) * (double)s->batch * s->length, st);
    WN_HIP(launch_hgemm(precision, cp.kb.kernel(), HEPI_F32, a, st), "hgemm<conv dx>");
    return WN_OK;
}

// The same convolution with the half series on BOTH sides (the feature layer / output block of a model in a half mode stay in the
// layout between their convs): y_series = leaky(conv(x) + b) * out_scale  (leaky_slope = 1: no activation).  `packed` from
// wn_hconv_pack with the input's scale.
int wn_hconv_forward_series(const wn_conv_shape* s, int precision, const void* packed, const void* x, void* y_series, float out_scale,
                            float leaky_slope, unsigned* overflow_flag, wn_stream_t stream) {
    int off[WN_MAX_TAPS];
    int rc = check_hconv(s, precision, off);
    if (rc != WN_OK) return rc;
    if (!packed || !x || !y_series) return WN_ERR_NULL;
    if (!(out_scale > 0.0f)) return WN_ERR_BAD_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const HConvPlan cp = plan_hconv(s, precision);
    const int P = hp_planes(precision), Ci = s->in_channels, Co = s->out_channels, k = s->kernel_width;
    HGemmArgs a;
    fill_hgemm(a, cp.f, packed, cp.off_f, s->batch, s->length, s->ld, s->halo);
    const HView vx = view(x, Ci, s->ld, P);
    for (int j = 0; j < k; ++j) set_hseg(a, j, vx, off[j], cp.f.seg_nks[j]);
    a.dst[0] = dst_of(view(y_series, Co, s->ld, P));
    a.oscale2 = out_scale; a.leaky = leaky_slope; a.flag = overflow_flag;
    wn::ProfScopeShared prof(KC_HCONV_FWD, 2.0 * Co * (double)(k * Ci) * (double)s->batch * s->length, st);
    if (cp.col && cp.f.nslab == 1 && cp.f.MT == 2 && !cp.f.k32) {
        HColArgs c;
        col_args_common(c, a, Co, (char*)packed + cp.off_dump, s->batch, s->length, s->ld, s->halo);
        c.dst = a.dst[0]; c.bias = a.bias; c.oscale2 = out_scale; c.leaky = leaky_slope;
        WN_HIP(launch_hcol_conv(precision, false, c, st), "hcol<conv series>");
        return WN_OK;
    }
    WN_HIP(launch_hgemm(precision, cp.f.kernel(), HEPI_LEAKY, a, st), "hgemm<conv series>");
    return WN_OK;
}

// dx_series = (W^T dy) * leaky'(act): `act` (nullable: no activation in front of this conv) is the conv's INPUT as it was stored,
// i.e. the activated value, whose sign is the sign of the pre-activation; gradients keep the scale dy carries
int wn_hconv_backward_data_series(const wn_conv_shape* s, int precision, const void* packed, const void* dy, const void* act,
                                  float leaky_slope, void* dx_series, unsigned* overflow_flag, wn_stream_t stream) {
    int off[WN_MAX_TAPS];
    int rc = check_hconv(s, precision, off);
    if (rc != WN_OK) return rc;
    if (!packed || !dy || !dx_series) return WN_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    const HConvPlan cp = plan_hconv(s, precision);
    const int P = hp_planes(precision), Ci = s->in_channels, Co = s->out_channels, k = s->kernel_width;
    HGemmArgs a;
    fill_hgemm(a, cp.kb, packed, cp.off_kb, s->batch, s->length, s->ld, s->halo);
    a.bias = nullptr;
    const HView vdy = view(dy, Co, s->ld, P);
    for (int j = 0; j < k; ++j) set_hseg(a, j, vdy, -off[j], cp.kb.seg_nks[j]);
    a.dst[0] = dst_of(view(dx_series, Ci, s->ld, P));
    if (act) a.z = dst_of(view(act, Ci, s->ld, P));
    a.oscale2 = 1.0f; a.leaky = act ? leaky_slope : 1.0f; a.flag = overflow_flag;
    wn::ProfScopeShared prof(KC_HCONV_BWD_DATA, 2.0 * Ci * (double)(k * Co) * (double)s->batch * s->length, st);
    if (cp.col && act && cp.kb.nslab == 1 && cp.kb.MT == 2 && !cp.kb.k32) {
        HColArgs c;
        col_args_common(c, a, Ci, (char*)packed + cp.off_dump, s->batch, s->length, s->ld, s->halo);
        c.dst = a.dst[0]; c.mask = a.z; c.oscale2 = 1.0f; c.leaky = leaky_slope;
        WN_HIP(launch_hcol_conv(precision, true, c, st), "hcol<conv dx series>");
        return WN_OK;
    }
    WN_HIP(launch_hgemm(precision, cp.kb.kernel(), HEPI_LEAKY, a, st), "hgemm<conv dx series>");
    return WN_OK;
}

namespace {
std::vector<HPairSpec> hconv_pairs(const wn_conv_shape* s, const int* off, const void* x, const void* dy, float* dw, float* db,
                                   float input_scale) {
    std::vector<HPairSpec> ps;
    const int Ci = s->in_channels, Co = s->out_channels, k = s->kernel_width;
    for (int j = 0; j < k; ++j)
        ps.push_back(plain_pair(dy, Co, x, Ci, off[j], (j == 0 && db) ? 1 : 0, 1.0f / input_scale, dw ? dw + j : nullptr, Ci * k, k,
                                (j == 0) ? db : nullptr, nullptr));
    return ps;
}
}  // namespace

size_t wn_hconv_wgrad_workspace_bytes(const wn_conv_shape* s, int precision) {
    int off[WN_MAX_TAPS];
    if (check_hconv(s, precision, off) != WN_OK) return 0;
    static float dummy = 0;
    size_t need = 0;
    std::vector<HPairSpec> ps = hconv_pairs(s, off, &dummy, &dummy, &dummy, &dummy, 1.0f);
    run_hwgrad(ps, precision, s->batch, s->length, s->ld, s->halo, nullptr, nullptr, 0, true, &need, nullptr);
    return need;
}

int wn_hconv_backward_weights(const wn_conv_shape* s, int precision, const void* x, const void* dy, float input_scale,
                              float* dweight, float* dbias, const float* dyn_inv_scale, void* workspace, size_t workspace_bytes,
                              wn_stream_t stream) {
    int off[WN_MAX_TAPS];
    int rc = check_hconv(s, precision, off);
    if (rc != WN_OK) return rc;
    if (!x || !dy || !dweight) return WN_ERR_NULL;
    if (!(input_scale > 0.0f)) return WN_ERR_BAD_SHAPE;
    std::vector<HPairSpec> ps = hconv_pairs(s, off, x, dy, dweight, dbias, input_scale);
    return run_hwgrad(ps, precision, s->batch, s->length, s->ld, s->halo, dyn_inv_scale, workspace, workspace_bytes, false, nullptr,
                      (hipStream_t)stream);
}
