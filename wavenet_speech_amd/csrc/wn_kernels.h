// Internal kernel-side declarations shared by the .hip translation units.
// Not part of the C ABI (see include/wavenet_amd.h for that).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// (the dilated taps x[t+off] start at arbitrary, only 4-byte aligned columns: the kernels fetch them with 16-byte
// buffer loads, which need dword alignment only)

constexpr int kMaxSeg = 32;   // K-segments of one GEMM  (>= 1 + 2*WN_MAX_TAPS and >= WN_MAX_STACK_GROUP)
constexpr int kMaxSlab = 32;  // M-slabs of one GEMM     (WN_MAX_CHANNELS*2 / 64)
constexpr int kMaxDst = 2;
constexpr int kColTile = 128; // time steps per wave tile (4 MFMA N-tiles of 32, lane n <-> columns 4n..4n+3)

// ---------------------------------------------------------------------------------------------
// "series GEMM":   out[M x (B*L)] = Wpacked[M x K] * Bop[K x (B*L)]
// The K rows of Bop are a concatenation of SEGMENTS; segment s is `cp` channel rows of one
// padded series tensor read at a column offset `off` (the dilated tap).  M is cut into SLABS of
// MT*32 rows; one wavefront (= one 64-thread workgroup) owns one slab x one 128-column tile and
// keeps its MT x 4 accumulator tiles (32x32 each) in AGPRs.
// ---------------------------------------------------------------------------------------------
struct GemmSeg {
    const float* base;  // series buffer [B][cp][ld]
    int cp;             // channel rows per batch element (multiple of 8)
    int off;            // column offset added to t
    int nkb;            // cp / 8 k-blocks
    int _pad;
};

struct GemmSlab {
    long long woff;  // float offset of this slab's packed weights
    int nseg;        // number of leading segments this slab contracts over
    int dst;         // EPI_LINEAR: destination index
    int row0;        // first destination row / channel of the slab
    int boff;        // float offset of this slab's packed bias (MT*32 floats)
};

struct GemmDst {
    float* base;     // series buffer [B][cp][ld]
    int cp;
    int rows;        // valid rows (C)
    int accumulate;  // 1: out += result
    int _pad;
};

struct GemmArgs {
    const float* wpacked;
    const float* bias;  // packed, may be nullptr (=0)
    GemmSeg seg[kMaxSeg];
    GemmSlab slab[kMaxSlab];
    GemmDst dst[kMaxDst];
    // gate epilogues (channel-indexed series with cp = gate_cp)
    float* sg;
    float* z;         // EPI_GATE out
    float* da;        // EPI_DGATE out
    float* dg;
    int gate_cp;
    int gate_rows;    // Co
    int nslab;
    int B, L, ld, halo;
    int tiles_per_row;  // ceil(L / column tile)
    int ncol;           // B * tiles_per_row
    unsigned long long* stamps;  // diagnostic build (-DWN_STAMPS) only: 8 words per workgroup
};

enum { EPI_LINEAR = 0, EPI_GATE = 1, EPI_DGATE = 2, EPI_ACCUM = 3 };  // ACCUM: dst += result (+ bias)

// ---------------------------------------------------------------------------------------------
// weight packing: logical matrix element (slab, tile m, row i, seg, c) -> source tensors
// ---------------------------------------------------------------------------------------------
struct PackSrc {          // one K-segment of one source set
    const float* ptr;     // nullptr => zeros
    int rows, cols;       // valid extent
    int stride_r, stride_c;
};
struct PackSet {
    PackSrc seg[kMaxSeg];
    const float* bias0;   // summed biases (nullable)
    const float* bias1;
    int bias_rows;
    int _pad;
};
struct PackTile { int set; int row0; };  // row0 < 0 => all-zero tile
struct PackArgs {
    PackSet set[2];
    PackTile tile[kMaxSlab * 4];   // [slab*MT + m]
    int seg_nkb[kMaxSeg];          // k-blocks per segment (same for every slab)
    long long slab_woff[kMaxSlab];
    int slab_nseg[kMaxSlab];
    int slab_boff[kMaxSlab];
    int nslab, MT;
    float* wpacked;
    float* bias;
    long long total;               // packed weight floats
};

// ---------------------------------------------------------------------------------------------
// weight-gradient kernel: out_p[M x N] = sum_{b,t} A_p[b][m][t] * B_p[b][n][t + off_p]
// ---------------------------------------------------------------------------------------------
constexpr int kMaxPair = 2 * 8 + 3;  // 2*WN_MAX_TAPS + 3
constexpr int kMaxReduceDst = 56;    // destinations of one reduction launch (a composite pair of the half path has up to four; several blocks per launch)
struct WgradPair {
    const float* A; const float* Bm;
    int a_cp, b_cp;       // rows per batch element
    int off;              // column offset on the B side
    int mt, nt;           // workgroup tiles along M / N
    int tile0;            // first workgroup-tile index of this pair
    long long slab_off;   // float offset of this pair's [Mp x Np] block inside one split's slab
    int Mp, Np;           // padded dims (multiples of the workgroup tile)
    int rowsum;           // also emit row sums of A (bias gradient)
    int rs_off;           // float offset of the row sums inside the split's row-sum area
};
struct WgradArgs {
    WgradPair pair[kMaxPair];
    int npair;
    int ntile_total;      // sum over pairs of mt*nt
    int nsplit;
    int B, L, ld, halo;
    int chunks_per_row;   // ceil(L/32)
    int nchunk;           // B * chunks_per_row
    float* slab;          // [nsplit][slab_floats]
    float* rowsum;        // [nsplit][rs_floats]
    long long slab_floats;
    int rs_floats;
    int xcd_map;          // 1: nsplit % 8 == 0 and all tiles of a split are placed on one XCD
};

struct ReduceDst {
    float* w;             // destination of the [M x N] matrix: w[m*sm + n*sn]
    int M, N, sm, sn;
    long long slab_off; int Np;
    float* b0; float* b1; // row-sum destinations (nullable)
    int rs_off;
    float post;           // the matrix is multiplied by this on the way out (1 in the fp32 path; 1/input scale in the half path)
};
struct ReduceArgs {   // <= 4 KiB: it travels as a kernel argument
    ReduceDst d[kMaxReduceDst];
    int npair, nsplit;
    const float* slab; const float* rowsum;
    long long slab_floats; int rs_floats;
    const float* dyn_inv; // optional device scalar multiplied into every output (1 / dynamic gradient scale of the half path)
};

static_assert(sizeof(ReduceArgs) <= 4096 && sizeof(WgradArgs) <= 4096, "kernel arguments are limited to 4 KiB");

// launchers (defined in the .hip files)
hipError_t launch_gemm(int MT, int epi, const GemmArgs& a, hipStream_t st);
hipError_t launch_pack(const PackArgs& a, hipStream_t st);
hipError_t launch_wgrad(int WT, const WgradArgs& a, hipStream_t st);
hipError_t launch_wgrad_reduce(const ReduceArgs& a, hipStream_t st);
hipError_t launch_nll_forward(const float* logits, const long long* target, float* lse, float* partial, int* bad_targets, int B,
                              int C, int L, hipStream_t st);
hipError_t launch_nll_backward(const float* logits, const long long* target, const float* lse, const float* gscale,
                               float* dlogits, int B, int C, int L, hipStream_t st);

}  // namespace wn
