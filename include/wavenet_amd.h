/*
 * wavenet_amd.h -- C ABI of libwavenet_amd.so: the MI355X (gfx950) implementation of the
 * WaveNet dilated residual-block hot path of paultsw/wavenet-speech.
 *
 * The reference has no FFI layer (it is pure Python on stock PyTorch ops); the
 * boundary it exposes for this path is the nn.Module surface
 *     modules/conv_ops.py:8-79   CausalConv1d / NonCausalConv1d
 *     modules/block.py:15-82     ResidualBlock(in,out,k,d,causal).forward(seq) -> (residual_out, skip_out)
 *     modules/wavenet.py:98-100  the per-layer loop  out,skip = convolutions[l](out); skips_sum += bottlenecks[l](skip)
 * Each entry point below names the reference code it replaces.  The Python host side
 * (wavenet_speech_amd/) binds these with ctypes and re-creates that nn.Module surface on top.
 *
 * Conventions
 *  - plain C: raw DEVICE pointers, ints, a stream handle (hipStream_t passed as void*).  No torch types.
 *  - the library never allocates or frees device memory and keeps no device state: every buffer
 *    (activations, packed weights, saved tensors, workspaces) is owned by the caller.  Host-side state it does keep:
 *    the measurement hooks' event lists (wn_prof_*, off by default, process-wide behind a mutex) and the text of the last
 *    HIP error per thread (wn_last_hip_error).
 *  - every function only enqueues work on `stream` and returns 0 or a negative wn_status code;
 *    no exceptions cross the ABI.  wn_strerror() gives the text.
 *  - re-entrant; one process per GPU for data parallelism.
 *  - dtype: the wn_block_* / wn_conv_* / wn_skipsum_* entry points are fp32 storage and fp32 arithmetic
 *    (v_mfma_f32_32x32x2_f32, exact fp32 fma chains); the wn_h* entry points further down are the half-precision-MFMA modes
 *    (wn_precision: f16x3 = two fp16 planes per operand and three products, f16, bf16; fp32 accumulation, their own
 *    "half series" layout).
 *
 * Padded series layout (all activation tensors -- "series" of B utterances x C channels x L steps)
 *    float buf[B][Cp][ld],  Cp = wn_round_up(C, 8),  ld = halo + wn_round_up(L,128) + halo
 *    sample (b, c, t) lives at buf[(b*Cp + c)*ld + halo + t];   halo = wn_round_up(max |tap offset|, 4)
 *    EVERYTHING outside the valid [C][L] window (pad rows, both halos, the tail up to the next
 *    multiple of 128) MUST be zero on input and is kept zero on output.  This is what lets the
 *    kernels read the dilated taps x[t-d] with unmasked, time-coalesced 16-byte loads.
 *    wn_series_layout() computes (ld, halo, Cp) so host code never hard-codes the rule.
 */
#ifndef WAVENET_AMD_H
#define WAVENET_AMD_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WN_VERSION 300 /* 0.3.0: round-3 entry points (fused forward, block-group weight gradients, pack tables, series convs, front-ends) */

typedef void* wn_stream_t; /* hipStream_t */

typedef enum wn_status {
    WN_OK = 0,
    WN_ERR_BAD_SHAPE = -1,   /* non-positive / inconsistent dimension, or layout (ld, halo) too small for the taps */
    WN_ERR_UNSUPPORTED = -2, /* kernel_width > WN_MAX_TAPS, channels > WN_MAX_CHANNELS, or channels*ld*4 >= 2^32 */
    WN_ERR_NULL = -3,        /* a required pointer is NULL */
    WN_ERR_HIP = -4,         /* a HIP runtime call or kernel launch failed (see wn_last_hip_error) */
    WN_ERR_WORKSPACE = -5    /* workspace smaller than wn_*_workspace_bytes(), or not 16-byte aligned */
} wn_status;

#define WN_MAX_TAPS 8
#define WN_MAX_CHANNELS 1024

/* Shape of one residual block call.  Mirrors ResidualBlock.__init__ (modules/block.py:22-51)
 * plus the batch geometry.  `skip_rows` is the row count of the skip projection: Co for the
 * stand-alone block (conv1x1_skip), or out_dim when the host has folded the stack's bottleneck
 * 1x1 into it (W = bottleneck.W @ conv1x1_skip.W, modules/wavenet.py:99-100). */
typedef struct wn_block_shape {
    int batch;        /* B */
    int length;       /* L, valid time steps */
    int in_channels;  /* Ci */
    int out_channels; /* Co */
    int skip_rows;    /* Ms */
    int kernel_width; /* k  (1..WN_MAX_TAPS) */
    int dilation;     /* d */
    int causal;       /* 1: CausalConv1d taps (j-(k-1))*d ; 0: NonCausalConv1d taps j*d - autopad(k,d) */
    int ld;           /* row pitch of every series buffer, floats */
    int halo;         /* zero columns before t=0 (and after the 128-rounded tail) */
} wn_block_shape;

/* The ten parameter tensors of a ResidualBlock in PyTorch's native layouts
 * (state_dict keys of modules/block.py:42-48).  Used for parameters and for their gradients. */
typedef struct wn_block_params {
    float* w_tanh;    /* conv_tanh.conv1d.weight     [Co][Ci][k] */
    float* b_tanh;    /* conv_tanh.conv1d.bias       [Co]        */
    float* w_sigmoid; /* conv_sigmoid.conv1d.weight  [Co][Ci][k] */
    float* b_sigmoid; /* conv_sigmoid.conv1d.bias    [Co]        */
    float* w_res;     /* conv1x1_residual.weight     [Co][Co](1) */
    float* b_res;     /* conv1x1_residual.bias       [Co]        */
    float* w_skip;    /* conv1x1_skip.weight         [Ms][Co](1)  (or the folded bottleneck*skip matrix) */
    float* b_skip;    /* conv1x1_skip.bias           [Ms]        */
    float* w_proj;    /* residual_proj.weight        [Co][Ci]    */
    float* b_proj;    /* residual_proj.bias          [Co]        */
} wn_block_params;

int wn_version(void);
const char* wn_strerror(int status);
/* text of the last HIP error seen by this thread's calls (empty string if none) */
const char* wn_last_hip_error(void);

int wn_round_up(int x, int multiple);
/* modules/conv_ops.py:104-116 autopad and the tap offsets of both conv flavours (off[j], j<k). */
int wn_autopad(int kernel_width, int dilation);
int wn_tap_offsets(int kernel_width, int dilation, int causal, int* off /* [k] */);
/* Padded series layout for L steps and taps reaching at most max_abs_offset columns away. */
int wn_series_layout(int length, int max_abs_offset, int* ld, int* halo);
/* floats in one series buffer of `channels` channels: B * round_up(C,8) * ld */
size_t wn_series_floats(int batch, int channels, int ld);

/* ---- weights: repack PyTorch-layout parameters into MFMA-fragment order -------------------
 * Once per optimizer step (weights are constant across the batch).  `packed` must hold
 * wn_block_packed_bytes() bytes; it is consumed by wn_block_forward / wn_block_backward_data. */
size_t wn_block_packed_bytes(const wn_block_shape* s);
int wn_block_pack(const wn_block_shape* s, const wn_block_params* p, void* packed, wn_stream_t stream);

/* ---- forward: replaces ResidualBlock.forward (modules/block.py:54-82) ----------------------
 *   a = conv_tanh(x), g = conv_sigmoid(x); ta = tanh(a), sg = sigmoid(g); z = ta*sg
 *   r    = W_res z + b_res + W_proj x + b_proj                       -> r_out (may be NULL: not needed)
 *   skip = W_skip z + b_skip                                         -> skip  (skip_accumulate=0)
 *   skip += W_skip z + b_skip   (the stack's running skips_sum)       -> skip  (skip_accumulate=1)
 *   skip == NULL: the skip product is left to wn_skipsum_forward (below)
 *   z (always written) and sg (may be NULL for inference) are what the backward pass needs: the tanh is recovered as z / sg
 *   there, which saves one tensor per block in the forward launch's stores and in the state held until backward.
 * x: [B][Ci8][ld]; r_out, sg, z: [B][Co8][ld]; skip: [B][Ms8][ld]. */
int wn_block_forward(const wn_block_shape* s, const void* packed, const float* x,
                     float* r_out, float* skip, int skip_accumulate,
                     float* sg, float* z, wn_stream_t stream);

/* ---- skips_sum of a whole stack as ONE long-K product (training, where every block's z is kept anyway) ----
 *   skip (+)= sum_l W_skip_l z_l + bias_total        == the sum over l of modules/wavenet.py:100
 * Call wn_block_forward with skip == NULL for each block (it then skips the per-block accumulation) and this once
 * per group of <= WN_MAX_STACK_GROUP blocks: K = sum_l C_l amortises the per-wave costs that dominate the per-block
 * K = C_l product and removes 2 HBM passes over skips_sum per block.
 * w_skip / z are HOST arrays of nblocks DEVICE pointers ([Ms][C_l] matrices and [B][C_l 8][ld] series). */
#define WN_MAX_STACK_GROUP 32
typedef struct wn_skipsum_shape {
    int batch, length, skip_rows, nblocks, ld, halo;
    int channels[WN_MAX_STACK_GROUP]; /* C_l */
} wn_skipsum_shape;
size_t wn_skipsum_packed_bytes(const wn_skipsum_shape* s);
int wn_skipsum_pack(const wn_skipsum_shape* s, const float* const* w_skip, const float* bias_total /* [Ms] or NULL */,
                    void* packed, wn_stream_t stream);
int wn_skipsum_forward(const wn_skipsum_shape* s, const void* packed, const float* const* z, float* skip,
                       int accumulate, wn_stream_t stream);

/* ---- backward (data): what autograd computes through modules/block.py:54-82 ----------------
 *   dz = W_res^T dr + W_skip^T dskip ;  da = dz*sg*(1-ta^2) ;  dg = dz*ta*sg*(1-sg), ta = z/sg     -> da, dg
 *   dx[t] = W_proj^T dr[t] + sum_j (W_tanh_j^T da + W_sigmoid_j^T dg)[t - off_j]         -> dx (may be NULL)
 * dr may be NULL (residual output unused, e.g. the last block of the stack).
 * dr, da, dg: [B][Co8][ld]; dskip: [B][Ms8][ld]; dx: [B][Ci8][ld]. */
int wn_block_backward_data(const wn_block_shape* s, const void* packed,
                           const float* dr, const float* dskip, const float* z, const float* sg,
                           float* da, float* dg, float* dx, wn_stream_t stream);

/* ---- backward (weights): time/batch-summed outer products -> gradients in PyTorch layouts ---
 *   dW_tanh[:,:,j] = sum da[t] x[t+off_j]^T, dW_sigmoid likewise with dg, dW_res = sum dr z^T,
 *   dW_proj = sum dr x^T, dW_skip = sum dskip z^T, biases = row sums.  Deterministic (split-K
 *   partial slabs reduced in a fixed order).  Gradients are OVERWRITTEN, not accumulated.
 *   dr may be NULL (no consumer of the residual output: the last block of a stack).  The four residual-path
 *   gradient pointers (w_res, b_res, w_proj, b_proj) may then be NULL too -- "no gradient", as autograd leaves
 *   them in the reference -- and any of them that is given is written as zeros. */
size_t wn_block_wgrad_workspace_bytes(const wn_block_shape* s);
int wn_block_backward_weights(const wn_block_shape* s, const float* x, const float* z,
                              const float* da, const float* dg, const float* dr, const float* dskip,
                              const wn_block_params* grads, void* workspace, size_t workspace_bytes,
                              wn_stream_t stream);

/* ---- stand-alone dilated conv: CausalConv1d / NonCausalConv1d (modules/conv_ops.py:8-79) ---
 * and, with kernel_width=1, any 1x1 Conv1d.  y = sum_j W[:,:,j] x[t+off_j] + b. */
typedef struct wn_conv_shape {
    int batch, length, in_channels, out_channels, kernel_width, dilation, causal, ld, halo;
} wn_conv_shape;
size_t wn_conv_packed_bytes(const wn_conv_shape* s);
int wn_conv_pack(const wn_conv_shape* s, const float* weight /*[Co][Ci][k]*/, const float* bias /*[Co] or NULL*/,
                 void* packed, wn_stream_t stream);
int wn_conv_forward(const wn_conv_shape* s, const void* packed, const float* x, float* y, wn_stream_t stream);
int wn_conv_backward_data(const wn_conv_shape* s, const void* packed, const float* dy, float* dx, wn_stream_t stream);
size_t wn_conv_wgrad_workspace_bytes(const wn_conv_shape* s);
int wn_conv_backward_weights(const wn_conv_shape* s, const float* x, const float* dy,
                             float* dweight, float* dbias /* may be NULL */,
                             void* workspace, size_t workspace_bytes, wn_stream_t stream);

/* ---- next-sample NLL head: replaces the L-iteration CrossEntropyLoss loop of Loss.py:38-43 /
 * legacy_code/train.py:37-39.  logits: dense [B][C][L] floats; target: [B][L] int64 class indices.
 *   forward : lse[b][t] = logsumexp_c logits[b][c][t];  partial[i] = sum over workgroup i of (lse - logits[target])
 *             (wn_nll_partials(B, L) floats; the caller sums them in order and divides by B: deterministic)
 *             A target outside [0, C) is never used as an index: it is counted in *bad_targets (one DEVICE int the caller
 *             zeroed; may be NULL) and makes its workgroup's partial NaN (torch asserts on the device in that case).
 *   backward: dlogits = (exp(logits - lse) - onehot(target)) * gscale[0]      (gscale: one DEVICE float, = dloss / B) */
size_t wn_nll_partials(int batch, int length);
int wn_nll_forward(const float* logits, const long long* target, float* lse, float* partial, int* bad_targets,
                   int batch, int classes, int length, wn_stream_t stream);
int wn_nll_backward(const float* logits, const long long* target, const float* lse, const float* gscale, float* dlogits,
                    int batch, int classes, int length, wn_stream_t stream);

/* ---- entry conv on quantised levels: replaces entry_conv1d(one_hot(levels)) of
 * modules/wavenet.py:54,93 + modules/fns.py:6-15 without materialising the [B][classes][L] one-hot.
 *   forward : y[b][co][t] = bias[co] + sum_j W[co][levels[b][t + j - (k-1)]][j]   (causal, dilation 1; y dense [B][Co][L])
 *   backward: no entry point of its own: dW[co][c][j] = sum over (b,t) with levels[b][t + j - (k-1)] == c of dy[b][co][t] is
 *             formed by wn_conv_backward_weights against a one-hot the caller builds for the duration of the backward call
 *             (the host mirror does; a gather-form kernel without it was measured 3.5x slower and removed in round 3).
 * levels: [B][L] int64 in [0, classes); anything else is counted in *bad_levels (DEVICE int, caller-zeroed, may be NULL)
 * and contributes nothing.  classes <= 512. */
int wn_embed_forward(const long long* levels, const float* weight, const float* bias, float* y, int batch, int length,
                     int classes, int out_channels, int kernel_width, int* bad_levels, wn_stream_t stream);
/* ---- synthetic reads on the device: the reference's on-line generator
 * utils/gaussian_kmer_model.py:53-104 (gaussian_model_fn :53-73, quantize_fn :79-86, one_hot_fn :89-97), float64 like its
 * numpy arithmetic.  All pointers are DEVICE pointers.
 *   wn_synth_bases     nucleotides 1..4 [B][nbases] int64 from a counter-based Philox4x32-10 stream of `seed`
 *   wn_synth_signal    picoamps[b][t] = means[k] + stdvs[k] * z,  k = 5-mer index of bases[b][p+2 .. p+6], p = t / upsampling
 *                      (scipy generic_filter's centred window after the [4:-4] trim: n bases give (n - 8) * upsampling
 *                      samples), z ~ N(0,1) from a second Philox stream of `seed`, or noise[b][t] when `noise` is not NULL
 *                      (the deterministic stages can then be checked against fixtures).  means / stdvs: 1024 doubles.
 *                      A base outside 1..4 is counted in *bad_bases (DEVICE int, caller-zeroed, may be NULL).
 *                      Also leaves per-tile (sum, min, max) partials in `workspace` for the next call.
 *   wn_synth_quantize  per read (x - mean) / (max - min), mu-law with mu = num_levels, np.digitize against `edges`
 *                      (num_levels ascending doubles: linspace(-1, 1, num_levels)); levels [B][L] int64 and, unless
 *                      one_hot is NULL, the dense one-hot [B][num_levels][L] fp32.  The per-read mean is a fixed-order sum:
 *                      results are bit-reproducible. */
size_t wn_synth_workspace_bytes(int batch, int length);
int wn_synth_bases(unsigned long long seed, int batch, int nbases, long long* bases, wn_stream_t stream);
int wn_synth_signal(const long long* bases, int batch, int nbases, int length, int upsampling, const double* means,
                    const double* stdvs, unsigned long long seed, const double* noise /* may be NULL */, double* picoamps,
                    void* workspace, size_t workspace_bytes, int* bad_bases /* may be NULL */, wn_stream_t stream);
int wn_synth_quantize(const double* picoamps, const void* workspace, size_t workspace_bytes, int batch, int length,
                      int num_levels, const double* edges, long long* levels, float* one_hot /* may be NULL */,
                      wn_stream_t stream);

/* ---- CTC on the device: replaces the warp-ctc call of Loss.py:49-53 (legacy_code/train.py:46,
 * pretrain_tnt.py:145,159, which copies the activations to the CPU every step).  warp-ctc semantics: softmax over the
 * classes inside the loss, one negative log likelihood per utterance (the caller sums them), and the gradient with respect
 * to the activations comes back with the loss.  Layout is the stack's own: logits / dlogits dense [B][C][T], time fastest --
 * nothing is permuted.  All pointers are DEVICE pointers.
 *   labels         [B][max_label_len] int64, utterance b uses the first label_lengths[b]; values in [0, C) and != blank
 *   label_lengths  [B] int64;  input_lengths [B] int64 or NULL (= every utterance has `length` frames)
 *   nll            [B] fp32: -log p(labels | logits); +inf when no alignment fits (its gradient rows are zero)
 *   dlogits        d nll[b] / d logits[b], or NULL for the loss alone
 * A label outside [0, C), equal to blank, or a length outside its range poisons that utterance (nll = NaN, zero gradient)
 * and is counted in *bad_labels (DEVICE int, caller-zeroed, may be NULL).  The recursions run in float64.
 * Limits: C <= 64, max_label_len <= 2047. */
size_t wn_ctc_workspace_bytes(int batch, int classes, int length, int max_label_len);
int wn_ctc_loss(const float* logits, const long long* labels, const long long* label_lengths,
                const long long* input_lengths /* may be NULL */, int batch, int classes, int length, int max_label_len,
                int blank, float* nll, float* dlogits /* may be NULL */, void* workspace, size_t workspace_bytes,
                int* bad_labels /* may be NULL */, wn_stream_t stream);

/* ======================================================================================================================
 * Half-precision-MFMA modes of the same path (opt-in; the entry points above stay exact fp32).
 *
 *   WN_F16X3  every operand is split into two fp16 planes (hi, lo); each product is hi*hi + hi*lo + lo*hi on
 *             v_mfma_f32_32x32x16_f16 with fp32 accumulation: 22-bit operands at 3/16 of the fp32 MFMA cost (within 1e-4 of the
 *             fp32 path on conditioned models; 3-4x further from fp64 than fp32 on the reference's ill-conditioned random init).
 *   WN_F16 / WN_BF16   one plane of fp16 / bf16 storage, one MFMA per product, fp32 accumulation
 *             (the dtypes BASELINE.json configs[4] / configs[1] name).
 *
 * "Half series" layout of every activation:  T buf[B][P][G][ld][8],  P = planes (2 for F16X3), G = round_up(C,32)/8;
 *   sample (b, c, t) of plane p at ((((b*P + p)*G + c/8)*ld + halo + t)*8 + c%8.  Eight channels of one time step are one
 *   16-byte unit = one MFMA operand fragment; a dilated tap is the same unit stream at another start.  Halos, the tail up to
 *   ld (= 2*halo + round_up(L, 256)) and the pad channels MUST be zero on input and are kept zero.
 * Power-of-two scales (exact): the residual stream (x, r_out) is stored as value * wn_hseries_residual_scale() (= 1/16, so
 *   fp16 holds |r| up to 1e6); ta, sg, z as is; every gradient series as value * s, where s is one DEVICE scalar per
 *   backward call chosen by the caller (pass 1/s as dyn_inv_scale where results leave the half domain).
 * fp16 stores that overflow (|v| > 65504) set *overflow_flag (a DEVICE unsigned the caller zeroed; may be NULL) to 1.
 * Everything else (ownership, streams, status codes, dr == NULL for the last block) is as for the fp32 entry points, whose
 * reference counterparts (modules/block.py:54-82, modules/wavenet.py:98-100) these replace in the same way. */
typedef enum wn_precision { WN_F32 = 0, WN_F16X3 = 1, WN_F16 = 2, WN_BF16 = 3 } wn_precision;

int wn_hseries_layout(int length, int max_abs_offset, int* ld, int* halo);
size_t wn_hseries_bytes(int precision, int batch, int channels, int ld);
float wn_hseries_residual_scale(void);
/* dense fp32 [B][C][L] -> half series, multiplied by scale * (dyn_scale ? *dyn_scale : 1).  Only the valid window is written. */
int wn_hseries_load(int precision, const float* dense, void* series, int batch, int channels, int length, int ld, int halo,
                    float scale, const float* dyn_scale, unsigned* overflow_flag, wn_stream_t stream);

size_t wn_hblock_packed_bytes(const wn_block_shape* s, int precision);
int wn_hblock_pack(const wn_block_shape* s, int precision, const wn_block_params* p, void* packed, wn_stream_t stream);
/* the same; *overflow_flag (DEVICE unsigned, caller-zeroed, may be NULL) is set when a weight leaves fp16's range after its
 * built-in scale (256 * w / input scale: |w| >= 16 for gate / projection weights) -- the gate would saturate silently otherwise */
int wn_hblock_pack_checked(const wn_block_shape* s, int precision, const wn_block_params* p, void* packed, unsigned* overflow_flag,
                           wn_stream_t stream);
/* x, r_out (nullable), sg (nullable: inference), z: half series.  z = tanh(a) sigmoid(g) and sg = sigmoid(g) are what the
 * backward pass needs (the tanh is recovered as z / sg: one tensor less to write and to keep).  skip_dense (nullable): dense
 * fp32 [B][Ms][L] that receives (skip_accumulate: += ) W_skip z + b_skip -- the per-block form used for inference.
 * Blocks of <= 128 channels with two taps run as ONE launch in the one-plane modes (wn_hblock_forward_is_fused() == 1: gate,
 * z and both products fused, z never leaves the chip between them -- the span modules/block.py:65-79 of the reference); there z
 * may be NULL when sg is NULL (inference: nothing needs it).  The fused kernel writes to a 1 KiB scratch line inside `packed`. */
int wn_hblock_forward_is_fused(const wn_block_shape* s, int precision);
int wn_hblock_forward(const wn_block_shape* s, int precision, const void* packed, const void* x, void* r_out,
                      float* skip_dense, int skip_accumulate, void* sg, void* z, unsigned* overflow_flag,
                      wn_stream_t stream);
size_t wn_hskipsum_packed_bytes(const wn_skipsum_shape* s, int precision);
int wn_hskipsum_pack(const wn_skipsum_shape* s, int precision, const float* const* w_skip, const float* bias_total,
                     void* packed, wn_stream_t stream);
/* z[l]: half series of every block; skip_dense: dense fp32 [B][Ms][L] */
int wn_hskipsum_forward(const wn_skipsum_shape* s, int precision, const void* packed, const void* const* z,
                        float* skip_dense, int accumulate, wn_stream_t stream);
/* dr (nullable), dskip, z, sg (the forward pass's), da, dg: half series (gradients carry the scale s).  The input gradient goes
 * either to the half series dx (scaled by s, the next block's dr) or to dense fp32 dx_dense [B][Ci][L] multiplied by
 * *dyn_inv_scale.  Blocks that take the fused forward and whose skip path is as wide as the block (<= 128 channels, one-plane
 * modes) run dz and the series dx as column-owner streaming kernels (hcol_kernel, csrc/wn_col_dev.h: the autograd of the
 * reference's modules/block.py:65-79 with the activations read straight from the series into registers); same arguments, same
 * packed weights.  The 1x1 wn_hconv_*_series calls below do likewise. */
int wn_hblock_backward_data(const wn_block_shape* s, int precision, const void* packed, const void* dr, const void* dskip,
                            const void* z, const void* sg, void* da, void* dg, void* dx, float* dx_dense,
                            const float* dyn_inv_scale, unsigned* overflow_flag, wn_stream_t stream);
size_t wn_hblock_wgrad_workspace_bytes(const wn_block_shape* s, int precision);
/* gradients in PyTorch layouts, fp32, multiplied by *dyn_inv_scale (and by the residual-stream scale where x is an operand) */
int wn_hblock_backward_weights(const wn_block_shape* s, int precision, const void* x, const void* z, const void* da,
                               const void* dg, const void* dr, const void* dskip, const wn_block_params* grads,
                               const float* dyn_inv_scale, void* workspace, size_t workspace_bytes, wn_stream_t stream);

/* The convolutions AROUND the block stack kept in the half series (the callers either side of the path; reference
 * modules/wavenet.py:67-71,103 output_stack, raw_ctcnet.py:57-61,89-93,128,148 feature_layer / output_block): LeakyReLU and the
 * 1x1 convs run without dense fp32 round trips between them.
 *   wn_hskipsum_forward_series     out = leaky(skips_sum) * out_scale as a half series (instead of the dense fp32 of
 *                                  wn_hskipsum_forward): the activated input of the output block's first conv
 *   wn_hconv_forward_series        y = leaky(conv(x) + b) * out_scale, series in, series out (leaky_slope 1 = no activation)
 *   wn_hconv_backward_data_series  dx = (W^T dy) * leaky'(act): act = the conv's stored (activated) input, NULL = no activation
 * Weight gradients: wn_hconv_backward_weights (series operands already).  out_scale is the scale the stored tensor carries
 * (wn_hseries_residual_scale() by convention, so that fp16 cannot overflow); pack the consumer with that input_scale. */
int wn_hskipsum_forward_series(const wn_skipsum_shape* s, int precision, const void* packed, const void* const* z, void* out_series,
                               float out_scale, float leaky_slope, unsigned* overflow_flag, wn_stream_t stream);
int wn_hconv_forward_series(const wn_conv_shape* s, int precision, const void* packed, const void* x, void* y_series, float out_scale,
                            float leaky_slope, unsigned* overflow_flag, wn_stream_t stream);
int wn_hconv_backward_data_series(const wn_conv_shape* s, int precision, const void* packed, const void* dy, const void* act,
                                  float leaky_slope, void* dx_series, unsigned* overflow_flag, wn_stream_t stream);

/* dx of `upper` and dz (-> da, dg) of `lower`, the block directly below it in the stack, in ONE launch: dz is pointwise in time
 * and its dr operand is the dx tile the same wave has just computed, so it goes from the MFMA result registers straight back into
 * the MFMA (csrc/wn_col2.hip: the autograd of two consecutive `out, skip = block(out)` steps of modules/wavenet.py:98-100).
 * dx_upper is still written (the lower block's weight gradients read it as their dr).  Only for pairs for which
 * wn_hblock_backward_pair_is_fused() == 1 (both blocks take the column-owner kernels, equal widths and geometry); otherwise
 * WN_ERR_UNSUPPORTED and the caller uses wn_hblock_backward_data per block.  dr_upper nullable (the top block of a stack).
 * wn_hblock_backward_input: the input gradient of a block whose gate gradients exist already (the bottom of such a chain). */
int wn_hblock_backward_pair_is_fused(const wn_block_shape* upper, const wn_block_shape* lower, int precision);
int wn_hblock_backward_pair(const wn_block_shape* upper, const void* packed_upper, const wn_block_shape* lower, const void* packed_lower,
                            int precision, const void* dr_upper, const void* da_upper, const void* dg_upper, const void* dskip,
                            const void* z_lower, const void* sg_lower, void* dx_upper, void* da_lower, void* dg_lower,
                            unsigned* overflow_flag, wn_stream_t stream);
int wn_hblock_backward_input(const wn_block_shape* s, int precision, const void* packed, const void* dr, const void* da, const void* dg,
                             void* dx, float* dx_dense, const float* dyn_inv_scale, const void* x_act, float leaky_slope,
                             unsigned* overflow_flag, wn_stream_t stream);

/* The first block of a stack whose input x is the ACTIVATED output of a front-end conv kept in the series (x_act = leaky(.) as
 * stored): dx = (input gradient) * leaky'(x_act), written to the half series dx -- wn_hblock_backward_data plus the LeakyReLU
 * backward of the layer in front, in one epilogue. */
int wn_hblock_backward_data_masked(const wn_block_shape* s, int precision, const void* packed, const void* dr, const void* dskip,
                                   const void* z, const void* sg, void* da, void* dg, void* dx, const void* x_act, float leaky_slope,
                                   unsigned* overflow_flag, wn_stream_t stream);

/* Front-ends that are not GEMM-shaped, written straight into the stack's layouts (csrc/wn_front.hip):
 *   wn_hfeature_forward            RawCTCNet.feature_layer[0..1] (reference modules/raw_ctcnet.py:57-61,128): Conv1d(1 -> F, k,
 *                                  padding k - 1) + LeakyReLU of the raw signal x [B][length] -> half series of length length + k - 1,
 *                                  stored * out_scale (k multiply-adds per element: elementwise work, not a GEMM with 31/32 of K zero)
 *   wn_hfeature_backward_weights   dW [F][1][k], db [F] from the series gradient of that layer's output (* dyn_inv_scale / dy_scale);
 *                                  deterministic (per-slab partial sums reduced in slab order)
 *   wn_hseries_load_pooled /       WaveNetClassifier.mean_pool (reference modules/classifier.py:53,102): AvgPool1d(pool) fused into
 *   wn_series_load_pooled          the load of the stack's input (half series / fp32 padded series of length length / pool)
 *   wn_pool_backward               dx[b][c][t] = dpooled[b][c][t / pool] / pool (0 for a dropped tail), both dense fp32 */
int wn_hfeature_forward(int precision, const float* x, const float* weight, const float* bias, void* y_series, int batch, int length,
                        int features, int kernel_width, int ld, int halo, float out_scale, float leaky_slope, unsigned* overflow_flag,
                        wn_stream_t stream);
size_t wn_hfeature_wgrad_workspace_bytes(int batch, int length, int features, int kernel_width);
int wn_hfeature_backward_weights(int precision, const float* x, const void* dy_series, float dy_scale, float* dweight, float* dbias,
                                 int batch, int length, int features, int kernel_width, int ld, int halo, const float* dyn_inv_scale,
                                 void* workspace, size_t workspace_bytes, wn_stream_t stream);
int wn_hseries_load_pooled(int precision, const float* dense, void* series, int batch, int channels, int length, int pool, int ld, int halo,
                           float scale, const float* dyn_scale, unsigned* overflow_flag, wn_stream_t stream);
int wn_series_load_pooled(const float* dense, float* series, int batch, int channels, int length, int pool, int ld, int halo,
                          wn_stream_t stream);
int wn_pool_backward(const float* dpooled, float* dx, int batch, int channels, int length, int pool, wn_stream_t stream);
/* the dynamic gradient scale of a backward call in the fp16 modes: scale_and_inverse[0] = 2^floor(log2(target / max|x|)) (clamped to
 * 2^+-100), [1] = its reciprocal -- device floats, no host synchronisation.  `accumulator`: one DEVICE unsigned, zero before the
 * first call (the call leaves it zero again).  x 16-byte aligned. */
int wn_grad_scale(const float* x, long long count, float target, float* scale_and_inverse, unsigned* accumulator, wn_stream_t stream);

/* The weight gradients of SEVERAL blocks of one series geometry in one launch (+ one reduction): blocks of <= 128 channels are
 * two or three gradient tiles each, so per-block launches are short split-K jobs dominated by their partial slabs; keep the
 * operands of up to wn_hblocks_wgrad_group_max() blocks (after their wn_hblock_backward_data calls) and hand them over
 * together.  Arrays are indexed by block; dr[l] may be NULL (a last block).  Same results as per-block calls up to the
 * summation order over time splits (deterministic either way). */
int wn_hblocks_wgrad_group_max(const wn_block_shape* s, int precision);
size_t wn_hblocks_wgrad_workspace_bytes(const wn_block_shape* shapes, int nblocks, int precision);
int wn_hblocks_backward_weights(const wn_block_shape* shapes, int nblocks, int precision, const void* const* x, const void* const* z,
                                const void* const* da, const void* const* dg, const void* const* dr, const void* const* dskip,
                                const wn_block_params* grads, const float* dyn_inv_scale, void* workspace, size_t workspace_bytes,
                                wn_stream_t stream);

/* Every weight-pack job of a stack of blocks in ONE launch.  Training repacks all weights after each optimizer step (five
 * launches per block through wn_hblock_pack, one per group through wn_hskipsum_pack); the jobs' arguments depend only on
 * shapes, precision and pointers, so they are built ONCE into a table: wn_hstack_pack_table_build fills `table_host` (host
 * memory of wn_hstack_pack_table_bytes(nblocks) bytes), the caller copies it to the device and keeps it; each step
 * wn_hstack_pack_run(table_dev, ...) packs into `packed`, a device buffer of *packed_total bytes that may be a different
 * allocation every time.  Block l's packed weights (what wn_hblock_forward / backward_* take) then start at
 * packed + block_offsets[l]; with_skipsum != 0 also packs the long-K skips_sum weights of wn_hskipsum_forward from the
 * blocks' w_skip / b_skip, group g (WN_MAX_STACK_GROUP blocks each) at packed + skipsum_offsets[g] -- the b_skip vectors
 * must then be equally spaced in memory (WN_ERR_UNSUPPORTED otherwise: use the per-block entry points).
 * `dynamic` (<= 3 ranges): parameter tensors inside one of these address ranges are re-allocated between steps; the table
 * stores them as offsets and wn_hstack_pack_run receives the current bases, in the same order.  Rebuild the table when
 * any other pointer, a shape or the precision changes.  (No reference counterpart: the reference has no packed weights.) */
typedef struct wn_mem_range { const void* base; size_t bytes; } wn_mem_range;
size_t wn_hstack_pack_table_bytes(int nblocks);
int wn_hstack_pack_table_build(const wn_block_shape* shapes, const wn_block_params* params, int nblocks, int precision,
                               int with_skipsum, const wn_mem_range* dynamic, int ndynamic, void* table_host, size_t table_bytes,
                               size_t* block_offsets, size_t* skipsum_offsets, size_t* packed_total, int* njobs,
                               int* launch_blocks);
int wn_hstack_pack_run(const void* table_dev, int nblocks, int njobs, int launch_blocks, const void* const* dynamic_bases,
                       int ndynamic, void* packed, unsigned* overflow_flag /* as wn_hblock_pack_checked */, wn_stream_t stream);

/* ---- measurement hooks (bench.py): HIP-event timing of every kernel on its launch stream ----
 * Kernel classes: index into wn_prof_kernel_name().  wn_prof_collect() synchronises the recorded
 * events and adds them to the per-class totals; wn_prof_get() reads them. */
int wn_prof_enable(int on);
int wn_prof_reset(void);
int wn_prof_collect(void);
int wn_prof_num_kernels(void);
const char* wn_prof_kernel_name(int kernel_class);
int wn_prof_get(int kernel_class, double* total_ms, long long* launches, double* flops);

/* ---- stand-alone dilated conv in the half-precision modes: the half-series counterpart of wn_conv_* (CausalConv1d /
 * NonCausalConv1d of modules/conv_ops.py:8-79 and the 1x1 Conv1d members of the output stacks), so that a model switched to a
 * half mode runs ALL of its convolutions on the half kernels.  x and dy are half series (wn_hseries_load; x stored as
 * x * input_scale, dy as dy * the call's gradient scale); y, dx, dweight, dbias are dense fp32.
 *   forward          y[b][co][t]  = bias[co] + sum_j W[co][:][j] x[b][:][t + off_j]
 *   backward_data    dx[b][ci][t] = sum_j W[:][ci][j]^T dy[b][:][t - off_j]            (* *dyn_inv_scale when given)
 *   backward_weights dW[co][ci][j] = sum_{b,t} dy[b][co][t] x[b][ci][t + off_j] / input_scale;  db = row sums of dy   (same) */
size_t wn_hconv_packed_bytes(const wn_conv_shape* s, int precision);
int wn_hconv_pack(const wn_conv_shape* s, int precision, const float* weight /*[Co][Ci][k]*/, const float* bias /* may be NULL */,
                  float input_scale, void* packed, wn_stream_t stream);
int wn_hconv_forward(const wn_conv_shape* s, int precision, const void* packed, const void* x, float* y_dense, wn_stream_t stream);
int wn_hconv_backward_data(const wn_conv_shape* s, int precision, const void* packed, const void* dy, float* dx_dense,
                           const float* dyn_inv_scale /* DEVICE scalar, may be NULL */, wn_stream_t stream);
size_t wn_hconv_wgrad_workspace_bytes(const wn_conv_shape* s, int precision);
int wn_hconv_backward_weights(const wn_conv_shape* s, int precision, const void* x, const void* dy, float input_scale,
                              float* dweight, float* dbias /* may be NULL */, const float* dyn_inv_scale /* may be NULL */,
                              void* workspace, size_t workspace_bytes, wn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* WAVENET_AMD_H */
