// Internal declarations of the half-precision-MFMA path (wn_half*.hip).  Not part of the C ABI.
//
// "Half series" layout (every activation of the half path): planes x channel-groups x time x 8 channels
//     T buf[B][P][G][ld][8],   T = _Float16 or __bf16,  G = round_up(C, 32) / 8,  P = planes
//     sample (b, c, t) -> plane p value at ((((b*P + p)*G + c/8)*ld + halo + t)*8 + c%8
//   P = 2 for "f16x3": plane 0 = hi = fp16(x*s), plane 1 = lo = fp16(x*s - hi)  (x*s ~ hi + lo to 22 bits);
//   P = 1 for plain f16 / bf16.  `s` is a power-of-two scale that belongs to the tensor (1 for ta/sg/z, kResidualScale
//   for the residual stream, a per-call dynamic scale for every gradient tensor).
//   Eight consecutive channels of one time step are one 16-byte unit: exactly the B-operand fragment of
//   v_mfma_f32_32x32x16_{f16,bf16} (lane (n, h) holds k = 8h..8h+7 of column n), so a tile of the series is staged into
//   LDS by global_load_lds_dwordx4 with no conversion, and a dilated tap x[t+off] is the same load at another start
//   unit -- always 16-byte aligned.  Halos, the tail up to ld and the pad channels are zero and stay zero.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "wn_kernels.h"

namespace wn {

enum { HP_F16X3 = 1, HP_F16 = 2, HP_BF16 = 3 };   // = wn_precision values of include/wavenet_amd.h
constexpr int kHCol = 256;          // time steps per workgroup tile of hgemm_kernel
constexpr int kHMaxSlab = 16;       // WN_MAX_CHANNELS * 2 / 128
constexpr float kResidualScale = 0.0625f;   // the residual stream is stored as r/16: fp16 then holds |r| up to 1.0e6
constexpr float kWeightScale = 256.0f;      // weights are packed as 256*w: kaiming-sized weights (~0.05) sit in fp16's normal range

inline __host__ __device__ int hp_planes(int prec) { return prec == HP_F16X3 ? 2 : 1; }

struct HSeg {               // one K-segment: `nks` k-steps (16 channels each) of one half series read at column offset `off`
    const char* base;
    long long ustride;      // bytes per utterance  (P * G * ld * 16)
    long long pstride;      // bytes per plane      (G * ld * 16)
    int off;
    int nks;
};

struct HSlab {
    long long woff;         // byte offset of this slab's packed weights
    int nseg;               // leading segments this slab contracts over
    int row0;               // first destination row / channel
    int boff;               // float offset of this slab's bias (ROWS floats)
    int dst;
};

struct HDst {               // a half-series destination (or source, for the dgate epilogue)
    char* base;
    long long ustride, pstride;
    int cp;                 // padded channels (8 * G); rows >= cp are not stored
    int _pad;
};

enum { HEPI_STORE = 0, HEPI_GATE = 1, HEPI_DGATE = 2, HEPI_F32 = 3, HEPI_LEAKY = 4 };
// HEPI_LEAKY: the convolutions around the block stack (feature layer, output block) kept in the half series:
//   forward  (z.base == nullptr)  y = leaky((acc * oscale + bias) * oscale2)            leaky(v) = v > 0 ? v : leaky * v
//   backward (z.base = the conv's input activation h = leaky(s), same shape as the output)
//                                  dx = acc * oscale * (signbit(h) ? leaky : 1)
//   The stored activation is the mask, read by its SIGN BIT: the forward stores every s <= 0 with the bit set (-0 for s = 0,
//   LeakyReLU's `x > 0` rule) and a positive s that underflows the format still stores +0 -- a `h > 0` test flipped one
//   element of 131072 in plain f16 (1/16-scaled activations), which moved a weight gradient by 3e-2 (kink sensitivity).

struct HGemmArgs {
    const char* wpacked;
    const float* bias;      // packed fp32, already multiplied by the output scale; may be nullptr
    HSeg seg[kMaxSeg];
    HSlab slab[kHMaxSlab];
    HDst dst[2];            // HEPI_STORE: dst[slab.dst]
    HDst sg, z;             // HEPI_GATE out (sg.base may be nullptr: inference) / HEPI_DGATE in (tanh = z / sigmoid)
    HDst da, dg;            // HEPI_DGATE out
    float* out32;           // HEPI_F32: dense [B][out32_rows][L] fp32
    const float* dyn_inv;   // HEPI_F32: optional device scalar multiplied into the result (1 / dynamic gradient scale)
    unsigned* flag;         // set to 1 when an fp16 store overflowed (|v| > 65504)
    float oscale;           // accumulators are multiplied by this exact power of two
    float oscale2, leaky;   // HEPI_LEAKY: scale of the stored result, negative-side slope
    int out32_rows, out32_accum;
    int gate_rows;          // valid channels of the gate epilogues
    int nslab, B, L, ld, halo;
    int tiles_per_row, ncol;
    int dbg;                // measurement only (WN_HGEMM_DBG): 1 = skip the K loop, 2 = skip the epilogue
};

// ---- weight packing ------------------------------------------------------------------------------------------
// Source pointers of a job are absolute, or -- in a job TABLE kept on the device and reused step after step
// (wn_hstack_pack_*) -- byte offsets from one of the run-time bases the launch supplies (HPackDyn): tensors that are
// allocated anew every step (the folded skip weights) then cost no table rebuild.
enum { HPACK_DYN_MASK = 3, HPACK_PERM = 4 };
struct HPackSrc {
    const float* ptr;       // nullptr => zeros
    int rows, cols;
    int stride_r, stride_c;
    float scale;            // multiplied into the weight (weight scale / input-tensor scale), exact power of two
    int flags;              // bits 0-1: dynamic base id (0: ptr is absolute; i: ptr is a byte offset from HPackDyn::base[i - 1]);
                            // HPACK_PERM: the 16 channels of a k-step are packed in accumulator order (wn_fused.hip: the gated z
                            // tile stays in the MFMA result registers and is fed back as the B operand)
};
struct HPackSet {
    HPackSrc seg[kMaxSeg];
    const float* bias0;     // summed: bias0[r + i * bias0_stride] for i < bias0_rep, + bias1[r]
    const float* bias1;
    int bias_rows;
    float bias_scale;
    int bias0_dyn, bias1_dyn;      // dynamic base ids as in HPackSrc::flags
    int bias0_rep, bias0_stride;   // rep <= 1: a single vector
};
struct HPackDyn {
    const char* base[3];
    char* out;              // table jobs: HPackArgs::wpacked / ::bias are byte offsets from here
    unsigned* flag;         // set to 1 when a scaled weight leaves fp16's range (fp16 modes; may be nullptr)
};
struct HPackArgs {
    HPackSet set[2];
    PackTile tile[kHMaxSlab * 8];   // [slab * (ROWS/32) + row tile] (16 slabs x 4 tiles or 8 x 8): source set and its first row (row0 < 0: zero tile)
    int seg_nks[kMaxSeg];
    long long slab_woff[kHMaxSlab]; // bytes
    int slab_nseg[kHMaxSlab];
    int slab_boff[kHMaxSlab];
    int nslab, rows;                // rows per slab (64 * MT)
    int planes, bf16;
    int kgroups, _pad2;             // k-groups of 8 channels per k-step: 2 (hgemm_kernel) or 4 (hgemm8_kernel); seg_nks in those k-steps
    char* wpacked;
    float* bias;
    long long total_units;          // 16-byte units per plane over all slabs
    unsigned* flag;                 // direct launches: as HPackDyn::flag
};
static_assert(sizeof(HPackArgs) <= 4096, "HPackArgs travels as a kernel argument");

// ---- dense fp32 <-> half series --------------------------------------------------------------------------------
struct HLoadArgs {
    const float* src;       // dense [B][C][L]
    char* dst;              // half series
    const float* dyn_scale; // optional device scalar multiplied into the values
    unsigned* flag;
    float scale;
    int B, C, L, G, ld, halo, planes, bf16;
};

// ---- weight gradients ------------------------------------------------------------------------------------------
struct HWgradPair {
    // The 256-channel A tile and B tile of a workgroup are each staged as two HALVES of 128 channels.  An ordinary pair reads
    // both halves from one tensor (half 1 = the next 16 channel groups: a_gb = {0, 16}); a COMPOSITE pair (all channel counts
    // <= 128: wn_half_api.hip) takes them from two tensors -- A = [da ; dg], B = [x(t + off_0) ; x(t + off_1)] -- so that one
    // 256 x 256 accumulator tile holds up to four 128 x 128 gradient matrices instead of one padded to 256 x 256.
    const char* A[2]; const char* Bm[2];    // half series, per half
    long long a_ustride[2], a_pstride[2], b_ustride[2], b_pstride[2];
    int a_groups[2], b_groups[2];           // channel groups (padded channels / 8) the operands really have
    int a_gb[2], b_gb[2];                   // first channel group of the half inside its tensor (before the tile's 32 tm / 32 tn)
    int off[2];                             // column offset on the B side, per half
    int quad_mask;                          // bit (2 * A half + B half): that 128 x 128 quadrant is wanted (its wave computes and stores it)
    int mt, nt;                             // 256-row / 256-column workgroup tiles
    int tile0;
    long long slab_off;                     // float offset of this pair's [Mp x Np] block inside a split's slab
    int Mp, Np;
    int rowsum, rs_off;
};
struct HWgradArgs {
    HWgradPair pair[kMaxPair];
    int npair, ntile_total, nsplit, xcd_map;
    int B, L, ld, halo;
    int steps_per_row;                  // ceil(L / 32) stages per utterance
    int nstep;                          // B * steps_per_row
    float* slab; float* rowsum;
    long long slab_floats; int rs_floats;
};

// ---- fused block forward (wn_fused.hip): <= 128 channels, one-plane modes ----------------------------------------------------
constexpr int kFRows = 128;                         // rows of every phase (gate half: [a ; g] of 64 channels; res; skip)
constexpr int kFStageBytes = kFRows * 32 * 2;       // one ring stage: 32 channels of K for the 128 rows (8 KiB)
constexpr int kFMaxGateK = 16;                      // 16-channel k-steps of the gate product: taps * round_up(Ci, 32) / 16
struct HFusedArgs {
    const char* wstream;        // the weights of all phases as ONE sequence of stages in consumption order: gate half 0, [gate half 1], res, skip
    const float* bias;          // [4][128] fp32: accumulator start values (bias / output scale) of gate half 0 / 1, res, skip
    const char* x;              // input half series
    long long x_ustride;
    HDst z, sg, r;              // base == nullptr: not stored
    float* skip;                // dense fp32 [B][skip_rows][L] (inference) or nullptr
    unsigned* flag;
    char* dump;                 // 1 KiB that masked store lanes write to (so that every store instruction is issued by every wave)
    int xunit[kFMaxGateK];      // 16-byte unit offset of gate k-step kk = (tap j, channels 16 ks ..): (2 ks) * ld + tap offset
    int nkg, nci16, nzt, co;    // gate k-steps; round_up(Ci, 32) / 16; z tiles of 32 channels; valid z channels
    int do_res, do_skip, skip_rows, skip_accum;
    int jump_at, jump;          // stream stages >= jump_at are `jump` stages further on (res phase not run)
    float osc_gate, osc_res, osc_skip;
    int B, L, ld, halo, units_per_row, nunit, nstage;
    int dbg;                    // measurement only (WN_FUSED_DBG)
    unsigned long long* stamps; // measurement only (WN_FUSED_STAMPS): 8 s_memtime stamps per workgroup
};
hipError_t launch_hfused_fwd(int prec, const HFusedArgs& a, hipStream_t st);

// ---- column-owner backward-data GEMMs (wn_col.hip): dz and dx of a block of <= 128 channels, one-plane modes --------------------
constexpr int kColMaxK = 128;                       // 16-channel k-steps: dx at 128 channels = (2 taps x 2 + 1) x 8 = 40; skips_sum of 16 blocks = 128
struct HColArgs {
    const char* wstream;            // the slab's packed weights: `nks` k-step images of 4 KiB ([k-half][128 rows][16 B])
    const char* kbase[kColMaxK];    // k-step kk: lane (utterance b, time t, k-half h) reads its B fragment at
    long long kustride[kColMaxK];   //     kbase[kk] + b * kustride[kk] + ((long long)h * ld + halo + t) * 16   (tap offset and channel group inside kbase)
    HDst z, sg, da, dg;             // HEPI_DGATE: in (tanh = z / sigmoid), out
    HDst dst;                       // HEPI_STORE / leaky epilogues
    HDst mask;                      // leaky backward: the activation whose sign bit is the mask
    const float* bias;              // leaky forward: fp32 [128], already multiplied by the output scale (zeros past the valid rows)
    unsigned* flag;
    char* dump;                     // 1 KiB that masked store lanes write to
    float oscale, oscale2, leaky;
    int nks, nt;                    // k-steps; row tiles of 32 output channels
    int B, L, ld, halo, nunit;      // nunit = ceil(B * L / 32) units of 32 consecutive valid columns
    int nwg;                        // workgroups of four units (set by the launcher)
    int dbg;
};
static_assert(sizeof(HColArgs) <= 4096, "kernel arguments are limited to 4 KiB");
// dx of a block + dz of the block below it in one launch (wn_col2.hip)
constexpr int kCol2MaxK = 48;                       // dx's k-steps (<= 40) + the dS segment of dz (<= 8)
struct HCol2Args {
    const char* wstream1;           // the upper block's KB slab (dx weights)
    const char* wstream2;           // the lower block's KAP slab (dz weights, the dx segment in accumulator order)
    const char* kbase[kCol2MaxK];   // as HColArgs: dx's operands (da, dg per tap [, dr]) then dS
    long long kustride[kCol2MaxK];
    HDst dx;                        // out: the upper block's input gradient = the lower block's dr
    HDst z, sg, da, dg;             // the lower block's saved activations (in) and gate gradients (out)
    unsigned* flag;
    char* dump;
    float oscale1, oscale2;
    int nt, hasdr;                  // row tiles of 32 channels (the same for both products); the upper block has a dr operand
    int B, L, ld, halo, nunit, nwg;
};
static_assert(sizeof(HCol2Args) <= 4096, "kernel arguments are limited to 4 KiB");
hipError_t launch_hcol2(int prec, const HCol2Args& a, hipStream_t st);
hipError_t launch_hcol(int prec, int epi, const HColArgs& a, hipStream_t st);        // dz (HEPI_DGATE), dx (HEPI_STORE; HEPI_LEAKY: masked)
hipError_t launch_hcol_conv(int prec, bool backward, const HColArgs& a, hipStream_t st);   // 1x1 convs in the series with LeakyReLU epilogues
hipError_t launch_hcol_skipsum(int prec, const HColArgs& a, hipStream_t st);                // skips_sum of <= 16 equally wide blocks -> leaky series
static_assert(sizeof(HWgradArgs) <= 4096 && sizeof(HGemmArgs) <= 4096 && sizeof(HFusedArgs) <= 4096, "kernel arguments are limited to 4 KiB");

hipError_t launch_hgemm(int prec, int MT, int epi, const HGemmArgs& a, hipStream_t st);
hipError_t launch_hpack(const HPackArgs& a, hipStream_t st);
// every job of a device-resident table in one launch: jobs[j] covers thread blocks block0[j] .. block0[j + 1] - 1
hipError_t launch_hpack_table(const HPackArgs* jobs, const int* block0, int njobs, int nblocks, const HPackDyn& dyn, hipStream_t st);
inline long long hpack_threads(const HPackArgs& a) {
    const long long nb = (long long)a.nslab * a.rows;
    return a.total_units > nb ? a.total_units : nb;
}
hipError_t launch_hload(const HLoadArgs& a, hipStream_t st);
hipError_t launch_hwgrad(int prec, const HWgradArgs& a, hipStream_t st);

}  // namespace wn
