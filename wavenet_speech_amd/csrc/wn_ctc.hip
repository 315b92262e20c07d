// Connectionist temporal classification on the device.  The reference hands its transcription logits
// to warp-ctc (SeanNaren/warp-ctc `pytorch_bindings`, unpinned; call sites Loss.py:49-53, legacy_code/train.py:46,
// pretrain_tnt.py:145,159 -- the last one moves the activations to the CPU every step): softmax over the labels inside the
// loss, blank = 0, negative log likelihoods summed over the batch, gradient with respect to the activations returned with
// the loss.  This file restates the published algorithm (Graves et al. 2006, eqs. 5-16) on the stack's own layout,
// logits [B][C][T] with time fastest, so nothing is permuted or copied:
//
//   ctc_lse_kernel     lse[b][t] = log sum_c exp(x[b][c][t])                                       (parallel over b, t)
//   ctc_pass_kernel    log-space alpha (forward) and beta (backward) over the 2 S + 1 blank-extended states, float64:
//                      one workgroup per (utterance, direction); T sequential steps, one barrier each, the row of the
//                      previous step in LDS, every row written to HBM.  beta is alpha on reversed time and labels.
//   ctc_grad_kernel    d nll / d x[b][c][t] = softmax(x)[c] - sum_{s : l'_s = c} exp(alpha_t(s) + beta_t(s) - logp_t(l'_s) + nll)
//                      one wave per time step, per-lane private class bins in LDS folded in lane order: deterministic.
//
// float64 in the recursions because log-likelihoods of thousands of frames reach magnitudes where fp32 resolves only
// ~1e-3, which is the relative error the occupancies would inherit.  The passes are latency-bound chains (T steps), not
// roofline work: measured next to torch's own ctc_loss in profiles/.
#include "../../include/wavenet_amd.h"
#include "wn_kernels.h"

namespace wn {

constexpr int kCtcThreads = 512;    // states in flight per pass workgroup
constexpr int kCtcMaxNS = 8;        // states per thread: up to 4096 extended states (2047 labels)
constexpr int kCtcChunk = 32;       // time steps of log-probabilities staged per refill
constexpr int kCtcMaxClasses = 64;

#define WN_NEG_INF (-__builtin_huge_val())

__device__ __forceinline__ double lse2(double a, double b) {
    const double m = fmax(a, b);
    if (m == WN_NEG_INF) return WN_NEG_INF;
    return m + log(exp(a - m) + exp(b - m));
}
__device__ __forceinline__ double lse3(double a, double b, double c) {
    const double m = fmax(a, fmax(b, c));
    if (m == WN_NEG_INF) return WN_NEG_INF;
    return m + log(exp(a - m) + exp(b - m) + exp(c - m));
}

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
    double* alpha; double* beta;    // [B][T][Sp]
    double* nll;                    // [B] (float64 copy used by the gradient kernel)
    float* nll_out;                 // [B]
    float* dx;                      // [B][C][T] or nullptr
    int* bad;
    int B, C, T, Lmax, Sp, blank;
};

// dir 0: alpha.  dir 1: beta, computed as alpha on reversed time and reversed labels and stored at the mirrored state.
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
    double* row0 = sh;                                  // [2][Sp + 2]: two guard entries (-inf) in front of each row
    double* lp = sh + 2 * (a.Sp + 2);                   // [C][kCtcChunk] log-probabilities of the staged steps
    int* lbad = reinterpret_cast<int*>(lp + C * kCtcChunk);
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
    for (int i = tid; i < 2 * (a.Sp + 2); i += kCtcThreads) row0[i] = WN_NEG_INF;
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
    double* out = (rev ? a.beta : a.alpha) + (long long)b * T * a.Sp;
    const float* xb = a.x + (long long)b * C * T;
    const double* lseb = a.lse + (long long)b * T;

    for (int k = 0; k < (int)Tb; ++k) {
        const int kc = k % kCtcChunk;
        if (kc == 0) {
            // stage log p of the next kCtcChunk steps: [c][kk] for pass steps k .. k+chunk-1
            __syncthreads();                                             // the previous chunk is no longer read
            for (int i = tid; i < C * kCtcChunk; i += kCtcThreads) {
                const int c = i / kCtcChunk, kk = i - c * kCtcChunk;
                const int ks = k + kk;
                double v = 0.0;
                if (ks < (int)Tb) {
                    const int t = rev ? (int)Tb - 1 - ks : ks;
                    v = (double)xb[(long long)c * T + t] - lseb[t];
                }
                lp[i] = v;
            }
        }
        __syncthreads();                                                 // previous row complete, chunk visible
        const double* prev = row0 + ((k + 1) & 1) * (a.Sp + 2) + 2;      // row written at step k-1
        double* cur = row0 + (k & 1) * (a.Sp + 2) + 2;
        const int t = rev ? (int)Tb - 1 - k : k;
        double* orow = out + (long long)t * a.Sp;
#pragma unroll
        for (int i = 0; i < kCtcMaxNS; ++i) {
            const int s = tid + i * kCtcThreads;
            if (s < S) {
                double v;
                if (k == 0) {
                    v = s < 2 ? 0.0 : WN_NEG_INF;                        // paths start in the first blank or the first label
                } else {
                    v = skip[i] ? lse3(prev[s], prev[s - 1], prev[s - 2]) : lse2(prev[s], prev[s - 1]);
                }
                v += lp[cls[i] * kCtcChunk + kc];
                cur[s] = v;
                orow[rev ? S - 1 - s : s] = v;
            }
        }
    }
    __syncthreads();
    if (tid == 0 && rev == 0) {
        const double* last = row0 + (((int)Tb - 1) & 1) * (a.Sp + 2) + 2;
        const double ll = S >= 2 ? lse2(last[S - 1], last[S - 2]) : last[S - 1];   // paths end in the last blank or the last label
        a.nll[b] = -ll;
        a.nll_out[b] = (float)(-ll);
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
        double lse_t = 0.0;
        if (compute) {
            for (int c = 0; c < C; ++c) mine[c] = 0.0;
            lse_t = a.lse[(long long)b * T + t];
            const double* ar = a.alpha + ((long long)b * T + t) * a.Sp;
            const double* br = a.beta + ((long long)b * T + t) * a.Sp;
            for (int s = lane; s < S; s += 64) {
                int c = a.blank;
                if (s & 1) {
                    const long long l = lab[s >> 1];
                    c = (l >= 0 && l < C) ? (int)l : a.blank;
                }
                const double lpv = (double)xb[(long long)c * T + t] - lse_t;
                mine[c] += exp(ar[s] + br[s] - lpv + nll);   // occupancy of state s at time t (<= 1; the states of a step sum to 1)
            }
        }
        __syncthreads();                                 // private bins complete
        if (compute) {
            for (int c = lane; c < C; c += 64) {
                double occ = 0.0;
                for (int l = 0; l < 64; ++l) occ += bins[((long long)wave * 64 + l) * C + c];   // lane order: deterministic
                const double y = exp((double)xb[(long long)c * T + t] - lse_t);
                dxb[(long long)c * T + t] = (float)(y - occ);
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
    a.lse = w; a.alpha = w + rows; a.beta = a.alpha + rows * Sp; a.nll = a.beta + rows * Sp;
    a.nll_out = nll; a.dx = dlogits; a.bad = bad_labels;
    a.B = batch; a.C = classes; a.T = length; a.Lmax = max_label_len; a.Sp = Sp; a.blank = blank;
    ProfScopeShared prof(KC_CTC, 0.0, st);
    hipLaunchKernelGGL(ctc_lse_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, logits, w, batch, classes, length);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail_shared(e, "ctc_lse");
    const size_t pass_lds = (size_t)(2 * (Sp + 2) + classes * kCtcChunk) * 8 + 16;
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
