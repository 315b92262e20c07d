"""
The training step the reference intends (legacy_code/train.py:24-61, Loss.py:18-58; train_tnt.py / Model.py cannot be
executed as shipped), on top of the MI355X modules:

    wavenet_pred  = wavenet(sig[:, :, :-1])                 # next-sample logits          (HIP path)
    transcription = ctcnet(wavenet_pred)                    # label logits per frame      (HIP path)
    xe  = sum_t CE(wavenet_pred[:, :, t], argmax(sig[:, :, t+1]))    # reference: an L-iteration Python loop
    ctc = CTC(transcription, labels + 1)                    # reference: warp-ctc on the CPU, blank = 0
    (xe / L + ctc / L').backward(); opt.step()

Differences, all outside the hot path: the per-timestep CE loop is one vectorised cross_entropy call (same value: CE
averages over the batch at each step and the steps are summed), and CTC stays on the device using
torch.nn.functional.ctc_loss(log_softmax(.), reduction="sum"), which reproduces the one known answer the reference
holds for warp-ctc (tests/test_classifier.py:59 -> 2.4628).
"""
import torch
import torch.nn.functional as F


def sequence_nll(pred, target):
    """sum over time of the batch-averaged cross entropy (Loss.py:38-43, legacy_code/train.py:37-39).
    pred: [B, C, L] logits, target: [B, L] int64.  Device tensors go through the fused HIP kernel
    (functional.sequence_nll); the torch expression is the CPU form used by the tests."""
    if pred.is_cuda:
        from . import functional as HF
        return HF.sequence_nll(pred, target)
    return F.cross_entropy(pred, target, reduction="sum") / pred.shape[0]


def ctc_total(transcription, labels, label_lengths, blank=0):
    """warp-ctc semantics (Loss.py:49-53): softmax applied internally, negative log likelihoods summed over the batch.
    transcription: [B, labels, T] logits; labels: [B, S] already offset so that 0 is <BLANK>."""
    logp = F.log_softmax(transcription.permute(2, 0, 1), dim=2)          # (T, B, C)
    T, B = logp.shape[0], logp.shape[1]
    in_lengths = torch.full((B,), T, dtype=torch.long, device=logp.device)
    return F.ctc_loss(logp, labels, in_lengths, label_lengths, blank=blank, reduction="sum", zero_infinity=False)


def joint_losses(wavenet, ctcnet, sig, seq, lengths):
    """sig: one-hot [B, 256, L] signal; seq: [B, S] int labels in 0..num_labels-2; lengths: [B] label lengths.
    Returns (avg_xe, avg_ctc, avg_joint) exactly as legacy_code/train.py:50-61 averages them."""
    pred = wavenet(sig[:, :, :-1])
    transcription = ctcnet(pred)
    dense = sig[:, :, 1:].argmax(dim=1)
    xe = sequence_nll(pred, dense)
    ctc = ctc_total(transcription, seq.long() + 1, lengths.long())
    avg_xe = xe / sig.shape[2]
    avg_ctc = ctc / transcription.shape[2]
    return avg_xe, avg_ctc, avg_xe + avg_ctc


def train_step(wavenet, ctcnet, sig, seq, lengths, opt, sync=None):
    """one optimisation step; `sync` is an optional parallel.FlatGradAllReduce for data parallelism"""
    if sync is not None:
        sync.zero()
    else:
        opt.zero_grad(set_to_none=True)
    avg_xe, avg_ctc, avg_joint = joint_losses(wavenet, ctcnet, sig, seq, lengths)
    avg_joint.backward()
    if sync is not None:
        sync.reduce()
    opt.step()
    return float(avg_xe.detach()), float(avg_ctc.detach()), float(avg_joint.detach())
