// Connectionist temporal classification on the device.  The reference hands its transcription logits
// to warp-ctc (SeanNaren/warp-ctc `pytorch_bindings`, unpinned; call sites Loss.py:49-53, legacy_code/train.py:46,
// pretrain_tnt.py:145,159 -- the last one moves the activations to the CPU every step): softmax over the labels inside the
// loss, blank = 0, negative log likelihoods summed over the batch, gradient with respect to the activations returned with
// the loss.  This file restates the published algorithm (Graves et al. 2006, eqs. 5-16) on the stack's own layout,
// logits [B][C][T] with time fastest, so nothing is permuted or copied:
//
//   ctc_lse_kernel     lse[b][t] = log sum_c exp(x[b][c][t])                                       (parallel over b, t)
//   ctc_pass_kernel    alpha (forward) and beta (backward) over the 2 S + 1 blank-extended states, float64 mantissas with a
//                      separate integer exponent per state: one workgroup per (utterance, direction); T sequential steps,
//                      one barrier each, the row of the previous step in LDS, every row written to HBM.  beta is alpha on
//                      reversed time and labels.
//   ctc_grad_kernel    d nll / d x[b][c][t] = softmax(x)[c] - sum_{s : l'_s = c} alpha_t(s) beta~_t(s) / sum_s alpha_t(s) beta~_t(s)
//                      one wave per time step, per-lane private class bins in LDS folded in lane order: deterministic.
//
// float64 because a rounding error per step compounds over thousands of frames (torch's fp32 GPU ctc_loss is 2e-2 off in
// the gradient at 4098 frames, tools/ctc_bench.py).  The passes are latency-bound chains (T steps), not roofline work:
// measured next to torch's own ctc_loss in profiles/.
#include "../../include/wavenet_amd.h"
#include "wn_kernels.h"

namespace wn {

constexpr int kCtcThreads = 512;    // states in flight per pass workgroup
constexpr int kCtcMaxNS = 8;        // states per thread: up to 4096 extended states (2047 labels)
constexpr int kCtcChunk = 32;       // time steps of log-probabilities staged per refill
constexpr int kCtcMaxClasses = 64;

__global__ __launch_bounds__(256) void ctc_lse_kernel(const float* __restrict__ x, double* __restrict__ lse, int B, int C, int T) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)B * T) return;
    const int b = (int)(i / T), t = (int)(i - (long long)b * T);
    const float* p = x + (long long)b * C * T + t;
    float m = -__builtin_huge_valf();
    for (int c = 0; c < C; ++c) m = fmaxf(m, p[(long long)c * T]);
    double s = 0.0;
    for (int c = 0; c < C; ++c) s += exp((double)p[(long long)c * T] - (double)m);
    lse[i] = (double)m + log(s);
}

struct CtcArgs {
    const float* x;                 // [B][C][T]
    const long long* labels;        // [B][Lmax]
    const long long* label_len;     // [B]
    const long long* input_len;     // [B] or nullptr (= T)
    const double* lse;              // [B][T]
    unsigned long long* alpha; unsigned long long* beta;    // [B][T][Sp] {float mantissa, int32 exponent}
    double* nll;                    // [B] (float64 copy used by the gradient kernel)
    float* nll_out;                 // [B]
    float* dx;                      // [B][C][T] or nullptr
    int* bad;
    int B, C, T, Lmax, Sp, blank;
};

// dir 0: alpha.  dir 1: beta, computed as alpha on reversed time and reversed labels and stored at the mirrored state.
//
// Number format of the recursions: every state value is a pair (m, e), value = m * 2^e, m a float64 in [1, 2) (or m = 0),
// e a 32-bit integer -- float64 precision with an unlimited exponent range:
//     sum  = ldexp(m0, e0 - emax) + ldexp(m1, e1 - emax) [+ ldexp(m2, e2 - emax)],  emax = max(e0, e1, e2)
//     v    = sum * y_t(l'_s);   e' = emax + ilogb(v),  m' = ldexp(v, -ilogb(v))
// about a dozen hardware instructions (v_ldexp_f64, v_frexp_*) per state and step.  A log-space update costs three exp and
// one log, which in float64 are software sequences of ~100 instructions each: that version of this kernel spent 2.4 us per
// time step in them.  Rescaling whole rows by one common factor (Rabiner's scaled forward-backward) is cheaper still but
// WRONG for long utterances: the states an alignment actually passes through can lie more than e^-709 below the row's
// largest forward value (measured: 2000 frames x 256 labels on random logits) and vanish from a plain float64 row.
// Rows go to HBM as {float mantissa, int32 exponent} (8 bytes per state); beta is stored WITHOUT the frame's own emission,
// so the gradient kernel multiplies alpha_t(s) * beta~_t(s) and never divides by y_t.
constexpr int kCtcZeroExp = -(1 << 28);                 // exponent of the value 0 (its mantissa is 0: any exponent is right)

__device__ __forceinline__ unsigned long long ctc_pack(double m, int e) {
    return ((unsigned long long)(unsigned)e << 32) | (unsigned long long)__float_as_uint((float)m);
}

__global__ __launch_bounds__(kCtcThreads) void ctc_pass_kernel(const CtcArgs a) {
    extern __shared__ double sh[];
    const int b = blockIdx.x, rev = blockIdx.y, tid = threadIdx.x;
    const int C = a.C, T = a.T;
    long long Tb = a.input_len ? a.input_len[b] : T;
    long long Lb = a.label_len[b];
    bool bad = Tb < 0 || Tb > T || Lb < 0 || Lb > a.Lmax;
    if (bad) { Tb = 0; Lb = 0; }
    const int S = 2 * (int)Lb + 1;
    const long long* lab = a.labels + (long long)b * a.Lmax;
    const int RS = a.Sp + 2;                            // row stride: two guard entries (value 0) in front of each row
    double* mrow = sh;                                  // [2][RS] mantissas
    double* yc = sh + 2 * RS;                           // [C][kCtcChunk] softmax probabilities of the staged steps
    int* erow = reinterpret_cast<int*>(yc + C * kCtcChunk);   // [2][RS] exponents
    int* lbad = erow + 2 * RS;
    if (tid == 0) *lbad = bad ? 1 : 0;
    __syncthreads();

    // this thread's states
    int cls[kCtcMaxNS];
    bool skip[kCtcMaxNS];
#pragma unroll
    for (int i = 0; i < kCtcMaxNS; ++i) {
        const int s = tid + i * kCtcThreads;
        cls[i] = a.blank; skip[i] = false;
        if (s < S && (s & 1)) {
            const int j = s >> 1;                                        // label index in pass order
            const long long l = lab[rev ? Lb - 1 - j : j];
            if (l < 0 || l >= C || l == a.blank) { atomicOr(lbad, 1); cls[i] = a.blank; }
            else cls[i] = (int)l;
            if (j >= 1) {
                const long long lprev = lab[rev ? Lb - j : j - 1];
                skip[i] = l != lprev;
            }
        }
    }
    for (int i = tid; i < 2 * RS; i += kCtcThreads) { mrow[i] = 0.0; erow[i] = kCtcZeroExp; }
    __syncthreads();
    const bool poisoned = *lbad != 0;
    if (poisoned && tid == 0 && rev == 0) {
        if (a.bad) atomicAdd(a.bad, 1);
        a.nll[b] = __builtin_nan("");
        a.nll_out[b] = __builtin_nanf("");
    }
    if (poisoned || Tb == 0) {
        if (!poisoned && tid == 0 && rev == 0) {
            // no frames: only the empty labelling has probability 1
            const double v = Lb == 0 ? 0.0 : __builtin_huge_val();
            a.nll[b] = v; a.nll_out[b] = (float)v;
        }
        return;
    }
    unsigned long long* out = (rev ? a.beta : a.alpha) + (long long)b * T * a.Sp;
    const float* xb = a.x + (long long)b * C * T;
    const double* lseb = a.lse + (long long)b * T;

    for (int k = 0; k < (int)Tb; ++k) {
        const int kc = k % kCtcChunk;
        if (kc == 0) {
            // stage y of the next kCtcChunk steps: [c][kk] for pass steps k .. k+chunk-1
            __syncthreads();                                             // the previous chunk is no longer read
            for (int i = tid; i < C * kCtcChunk; i += kCtcThreads) {
                const int c = i / kCtcChunk, kk = i - c * kCtcChunk;
                const int ks = k + kk;
                double v = 0.0;
                if (ks < (int)Tb) {
                    const int t = rev ? (int)Tb - 1 - ks : ks;
                    v = exp((double)xb[(long long)c * T + t] - lseb[t]);
                }
                yc[i] = v;
            }
        }
        __syncthreads();                                                 // previous row complete, chunk visible
        const int po = ((k + 1) & 1) * RS + 2, co = (k & 1) * RS + 2;    // row written at step k-1 / row of this step
        const int t = rev ? (int)Tb - 1 - k : k;
        unsigned long long* orow = out + (long long)t * a.Sp;
#pragma unroll
        for (int i = 0; i < kCtcMaxNS; ++i) {
            const int s = tid + i * kCtcThreads;
            if (s < S) {
                double pre; int pe;
                if (k == 0) {
                    pre = s < 2 ? 1.0 : 0.0; pe = s < 2 ? 0 : kCtcZeroExp;   // paths start in the first blank or the first label
                } else {
                    const int e0 = erow[po + s], e1 = erow[po + s - 1];
                    int emax = e0 > e1 ? e0 : e1;
                    int e2 = kCtcZeroExp;
                    if (skip[i]) { e2 = erow[po + s - 2]; emax = e2 > emax ? e2 : emax; }
                    // exponent gaps beyond float64's range contribute 0; clamping keeps the int subtraction away from overflow
                    double sum = ldexp(mrow[po + s], max(e0 - emax, -2200)) + ldexp(mrow[po + s - 1], max(e1 - emax, -2200));
                    if (skip[i]) sum += ldexp(mrow[po + s - 2], max(e2 - emax, -2200));
                    if (sum > 0.0) {
                        const int ex = ilogb(sum);
                        pre = ldexp(sum, -ex); pe = emax + ex;
                    } else {
                        pre = 0.0; pe = kCtcZeroExp;
                    }
                }
                double v = pre * yc[cls[i] * kCtcChunk + kc];
                int ve = pe;
                if (v > 0.0) {
                    const int ex = ilogb(v);
                    v = ldexp(v, -ex); ve = pe + ex;
                } else {
                    v = 0.0; ve = kCtcZeroExp;
                }
                mrow[co + s] = v; erow[co + s] = ve;
                orow[rev ? S - 1 - s : s] = rev ? ctc_pack(pre, pe) : ctc_pack(v, ve);   // beta is stored WITHOUT this frame's emission
            }
        }
    }
    __syncthreads();
    if (tid == 0 && rev == 0) {
        const int lo = (((int)Tb - 1) & 1) * RS + 2;
        // paths end in the last blank or the last label
        double m = mrow[lo + S - 1]; int e = erow[lo + S - 1];
        if (S >= 2) {
            const double m2 = mrow[lo + S - 2]; const int e2 = erow[lo + S - 2];
            const int emax = e > e2 ? e : e2;
            m = ldexp(m, max(e - emax, -2200)) + ldexp(m2, max(e2 - emax, -2200));
            e = emax;
        }
        const double nll = m > 0.0 ? -(log(m) + (double)e * 0.69314718055994530942) : __builtin_huge_val();
        a.nll[b] = nll;
        a.nll_out[b] = (float)nll;
    }
}

// one wave per time step
__global__ __launch_bounds__(256) void ctc_grad_kernel(const CtcArgs a, int steps_per_wave) {
    extern __shared__ double bins[];                    // [4 waves][64 lanes][C]
    const int b = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int C = a.C, T = a.T;
    long long Tb = a.input_len ? a.input_len[b] : T;
    long long Lb = a.label_len[b];
    if (Tb < 0 || Tb > T || Lb < 0 || Lb > a.Lmax) { Tb = 0; Lb = 0; }
    const int S = 2 * (int)Lb + 1;
    const double nll = a.nll[b];
    const bool usable = nll == nll && nll != __builtin_huge_val();   // not poisoned, not infeasible
    const long long* lab = a.labels + (long long)b * a.Lmax;
    double* mine = bins + ((long long)wave * 64 + lane) * C;
    const float* xb = a.x + (long long)b * C * T;
    float* dxb = a.dx + (long long)b * C * T;
    const int t_begin = (blockIdx.x * 4 + wave) * steps_per_wave;
    for (int i = 0; i < steps_per_wave; ++i) {           // uniform trip count: the barriers below are reached by every thread
        const int t = t_begin + i;
        const bool inside = t < T;
        const bool compute = inside && t < (int)Tb && usable;
        if (inside && !compute) {                        // frames past the utterance, infeasible or rejected labellings: no gradient
            for (int c = lane; c < C; c += 64) dxb[(long long)c * T + t] = 0.0f;
        }
        double total = 0.0;
        if (compute) {
            for (int c = 0; c < C; ++c) mine[c] = 0.0;
            const unsigned long long* ar = a.alpha + ((long long)b * T + t) * a.Sp;
            const unsigned long long* br = a.beta + ((long long)b * T + t) * a.Sp;
            // the frame's largest alpha * beta~ exponent (the occupancies are normalised per frame, so any common factor will do)
            int emax = 2 * kCtcZeroExp;
            for (int s = lane; s < S; s += 64) {
                const unsigned long long pa = ar[s], pb = br[s];
                if ((unsigned)pa != 0u && (unsigned)pb != 0u) {
                    const int e = (int)(pa >> 32) + (int)(pb >> 32);
                    emax = e > emax ? e : emax;
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const int v = __shfl_xor(emax, o); emax = v > emax ? v : emax; }
            for (int s = lane; s < S; s += 64) {
                int c = a.blank;
                if (s & 1) {
                    const long long l = lab[s >> 1];
                    c = (l >= 0 && l < C) ? (int)l : a.blank;
                }
                const unsigned long long pa = ar[s], pb = br[s];
                const double mm = (double)__uint_as_float((unsigned)pa) * (double)__uint_as_float((unsigned)pb);
                const int e = (int)(pa >> 32) + (int)(pb >> 32);
                const double g = mm > 0.0 ? ldexp(mm, max(e - emax, -2200)) : 0.0;   // proportional to the occupancy of state s
                mine[c] += g;
                total += g;
            }
            // the frame's normaliser: the wave's total in a fixed shuffle order
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o);
        }
        __syncthreads();                                 // private bins complete
        if (compute) {
            const double lse_t = a.lse[(long long)b * T + t];
            const double inv = total > 0.0 ? 1.0 / total : 0.0;
            for (int c = lane; c < C; c += 64) {
                double occ = 0.0;
                for (int l = 0; l < 64; ++l) occ += bins[((long long)wave * 64 + l) * C + c];   // lane order: deterministic
                const double y = exp((double)xb[(long long)c * T + t] - lse_t);
                dxb[(long long)c * T + t] = (float)(y - occ * inv);
            }
        }
        __syncthreads();                                 // bins free for the next step
    }
}

}  // namespace wn

namespace wn {
int hip_fail_shared(hipError_t e, const char* what);
struct ProfScopeShared { void* impl; ProfScopeShared(int kc, double flops, hipStream_t st); ~ProfScopeShared(); };
}
using namespace wn;
static const int KC_CTC = 19;   // index into wn_api.hip's kernel-class table

static int check_ctc(int batch, int classes, int length, int max_label_len, int blank) {
    if (batch <= 0 || classes <= 1 || length <= 0 || max_label_len <= 0) return WN_ERR_BAD_SHAPE;
    if (blank < 0 || blank >= classes) return WN_ERR_BAD_SHAPE;
    if (classes > kCtcMaxClasses || 2 * (long long)max_label_len + 1 > (long long)kCtcMaxNS * kCtcThreads || batch > 65535)
        return WN_ERR_UNSUPPORTED;
    if ((double)batch * (double)length >= 2147483648.0) return WN_ERR_UNSUPPORTED;
    return WN_OK;
}
static int ctc_sp(int max_label_len) { return (2 * max_label_len + 1 + 63) / 64 * 64; }

size_t wn_ctc_workspace_bytes(int batch, int classes, int length, int max_label_len) {
    if (check_ctc(batch, classes, length, max_label_len, 0) != WN_OK) return 0;
    const size_t rows = (size_t)batch * (size_t)length;
    return rows * 8 /* lse */ + 2 * rows * (size_t)ctc_sp(max_label_len) * 8 /* alpha, beta */ + (size_t)batch * 8 /* nll */ + 256;
}

int wn_ctc_loss(const float* logits, const long long* labels, const long long* label_lengths, const long long* input_lengths,
                int batch, int classes, int length, int max_label_len, int blank, float* nll, float* dlogits, void* workspace,
                size_t workspace_bytes, int* bad_labels, wn_stream_t stream) {
    int rc = check_ctc(batch, classes, length, max_label_len, blank);
    if (rc != WN_OK) return rc;
    if (!logits || !labels || !label_lengths || !nll || !workspace) return WN_ERR_NULL;
    if (workspace_bytes < wn_ctc_workspace_bytes(batch, classes, length, max_label_len)) return WN_ERR_WORKSPACE;
    if (reinterpret_cast<uintptr_t>(workspace) & 7) return WN_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const size_t rows = (size_t)batch * (size_t)length;
    const int Sp = ctc_sp(max_label_len);
    CtcArgs a;
    a.x = logits; a.labels = labels; a.label_len = label_lengths; a.input_len = input_lengths;
    double* w = reinterpret_cast<double*>(workspace);
    a.lse = w; a.alpha = reinterpret_cast<unsigned long long*>(w + rows); a.beta = a.alpha + rows * Sp;
    a.nll = reinterpret_cast<double*>(a.beta + rows * Sp);
    a.nll_out = nll; a.dx = dlogits; a.bad = bad_labels;
    a.B = batch; a.C = classes; a.T = length; a.Lmax = max_label_len; a.Sp = Sp; a.blank = blank;
    ProfScopeShared prof(KC_CTC, 0.0, st);
    hipLaunchKernelGGL(ctc_lse_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, logits, w, batch, classes, length);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail_shared(e, "ctc_lse");
    const size_t pass_lds = (size_t)(2 * (Sp + 2) + classes * kCtcChunk) * 8 + (size_t)(2 * (Sp + 2)) * 4 + 16;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(ctc_pass_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pass_lds);
    if (e != hipSuccess) return hip_fail_shared(e, "ctc_pass attribute");
    hipLaunchKernelGGL(ctc_pass_kernel, dim3(batch, dlogits ? 2 : 1), dim3(kCtcThreads), pass_lds, st, a);
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail_shared(e, "ctc_pass");
    if (dlogits) {
        const int steps_per_wave = 4;
        const size_t grad_lds = (size_t)4 * 64 * classes * 8;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(ctc_grad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)grad_lds);
        if (e != hipSuccess) return hip_fail_shared(e, "ctc_grad attribute");
        hipLaunchKernelGGL(ctc_grad_kernel, dim3((unsigned)((length + 4 * steps_per_wave - 1) / (4 * steps_per_wave)), batch), dim3(256),
                           grad_lds, st, a, steps_per_wave);
        e = hipGetLastError();
        if (e != hipSuccess) return hip_fail_shared(e, "ctc_grad");
    }
    return WN_OK;
}
