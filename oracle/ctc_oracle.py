"""
CPU ORACLE for the CTC loss.  TEST INFRASTRUCTURE ONLY (same rules as wavenet_oracle.py: only tests/, smoke() and the
cpu_baseline leg of bench.py may import it, as the checker).

The reference does not implement CTC: it calls warp-ctc (SeanNaren/warp-ctc, branch `pytorch_bindings`, unpinned,
README.md:15-16; call sites Loss.py:49-53, legacy_code/train.py:46, pretrain_tnt.py:145,159), a dependency that is
absent from /root/reference and not installable here.  This file restates the published algorithm -- Graves, Fernandez,
Gomez, Schmidhuber, "Connectionist Temporal Classification", ICML 2006, eqs. 5-8 (forward variables), 9-11 (backward
variables), 14-16 (gradient) -- with warp-ctc's conventions as the reference uses them: activations are unnormalised (a
softmax over the labels is applied inside), blank = 0, the losses of a batch are summed, and the gradient is taken with
respect to the activations.

PINNED by the one known answer the reference holds for warp-ctc, tests/test_classifier.py:53-59: activations
[[0.1, 0.6, 0.1, 0.1, 0.1], [0.1, 0.1, 0.6, 0.1, 0.1]] (T = 2, one utterance), labels [1, 2] -> "approximately 2.4628"
(tests/test_ctc.py asserts it), and cross-checked there against torch.nn.functional.ctc_loss on the CPU.
Plain numpy float64, loops over time: small cases only.
"""
import numpy as np


def _logsumexp(values):
    m = np.max(values)
    if not np.isfinite(m):
        return m
    return m + np.log(np.sum(np.exp(values - m)))


def log_softmax(acts):
    """acts [C, T] -> log softmax over C (warp-ctc applies the softmax itself)"""
    m = acts.max(axis=0, keepdims=True)
    return acts - m - np.log(np.exp(acts - m).sum(axis=0, keepdims=True))


def extended(labels, blank):
    ext = [blank]
    for l in labels:
        ext += [int(l), blank]
    return ext


def ctc_nll_and_grad(acts, labels, blank=0):
    """acts [C, T] float64 activations of ONE utterance, labels: sequence of ints (no blanks).
    Returns (nll, d nll / d acts [C, T]).  nll = +inf (gradient zeros) when no alignment fits in T frames."""
    acts = np.asarray(acts, dtype=np.float64)
    C, T = acts.shape
    logp = log_softmax(acts)
    ext = extended(labels, blank)
    S = len(ext)
    NEG = -np.inf
    alpha = np.full((T, S), NEG)
    beta = np.full((T, S), NEG)
    # eq. 6-8: alpha_t(s) = y_t(l'_s) * (alpha_{t-1}(s) + alpha_{t-1}(s-1) [+ alpha_{t-1}(s-2) if l'_s != blank, != l'_{s-2}])
    alpha[0, 0] = logp[blank, 0]
    if S > 1:
        alpha[0, 1] = logp[ext[1], 0]
    for t in range(1, T):
        for s in range(S):
            terms = [alpha[t - 1, s]]
            if s >= 1:
                terms.append(alpha[t - 1, s - 1])
            if s >= 2 and ext[s] != blank and ext[s] != ext[s - 2]:
                terms.append(alpha[t - 1, s - 2])
            alpha[t, s] = _logsumexp(np.array(terms)) + logp[ext[s], t]
    # eq. 9-11 (with the emission of frame t included, as in eq. 10)
    beta[T - 1, S - 1] = logp[blank, T - 1]
    if S > 1:
        beta[T - 1, S - 2] = logp[ext[S - 2], T - 1]
    for t in range(T - 2, -1, -1):
        for s in range(S):
            terms = [beta[t + 1, s]]
            if s + 1 < S:
                terms.append(beta[t + 1, s + 1])
            if s + 2 < S and ext[s] != blank and ext[s] != ext[s + 2]:
                terms.append(beta[t + 1, s + 2])
            beta[t, s] = _logsumexp(np.array(terms)) + logp[ext[s], t]
    tail = [alpha[T - 1, S - 1]] + ([alpha[T - 1, S - 2]] if S > 1 else [])
    ll = _logsumexp(np.array(tail))                       # eq. 8
    if not np.isfinite(ll):
        return np.inf, np.zeros_like(acts)
    # eq. 16: d(-ln p)/d u_t(k) = y_t(k) - 1/(p y_t(k)) sum_{s in lab(k)} alpha_t(s) beta_t(s)
    grad = np.exp(logp)
    for t in range(T):
        for s in range(S):
            v = alpha[t, s] + beta[t, s]
            if np.isfinite(v):
                grad[ext[s], t] -= np.exp(v - logp[ext[s], t] - ll)
    return -ll, grad


def ctc_total(acts, labels, label_lengths, blank=0, input_lengths=None):
    """acts [B, C, T]; labels [B, Lmax] padded; returns (sum of nll, gradient [B, C, T]) -- Loss.py:52's ctc_loss_fn value"""
    acts = np.asarray(acts, dtype=np.float64)
    B, C, T = acts.shape
    total, grads = 0.0, np.zeros_like(acts)
    for b in range(B):
        tb = T if input_lengths is None else int(input_lengths[b])
        nll, g = ctc_nll_and_grad(acts[b][:, :tb], [int(v) for v in labels[b][:int(label_lengths[b])]], blank)
        total += nll
        grads[b][:, :tb] = g
    return total, grads
