"""
The training step the reference intends (legacy_code/train.py:24-61, Loss.py:18-58; train_tnt.py / Model.py cannot be
executed as shipped), on top of the MI355X modules:

    wavenet_pred  = wavenet(sig[:, :, :-1])                 # next-sample logits          (HIP path)
    transcription = ctcnet(wavenet_pred)                    # label logits per frame      (HIP path)
    xe  = sum_t CE(wavenet_pred[:, :, t], argmax(sig[:, :, t+1]))    # reference: an L-iteration Python loop
    ctc = CTC(transcription, labels + 1)                    # reference: warp-ctc on the CPU, blank = 0
    (xe / L + ctc / L').backward(); opt.step()

Differences, all outside the hot path: the per-timestep CE loop is one vectorised cross_entropy call (same value: CE
averages over the batch at each step and the steps are summed), and CTC stays on the device: wn_ctc_loss (csrc/wn_ctc.hip)
reads the [B, labels, T] logits in place and returns loss and gradient, checked against the one known answer the
reference holds for warp-ctc (tests/test_classifier.py:59 -> 2.4628) and against oracle/ctc_oracle.py.
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


class _CTCFn(torch.autograd.Function):
    """wn_ctc_loss (csrc/wn_ctc.hip): like warp-ctc the gradient with respect to the activations is produced together with
    the loss; backward only scales it."""

    @staticmethod
    def forward(ctx, transcription, labels, label_lengths, input_lengths, blank):
        import ctypes
        from . import _lib
        from .functional import _p, _stream
        lib = _lib.load()
        x = transcription.detach().contiguous().float()
        B, C, T = x.shape
        dev = x.device
        labels = labels.to(device=dev, dtype=torch.int64).contiguous()
        label_lengths = label_lengths.to(device=dev, dtype=torch.int64).contiguous()
        if labels.dim() != 2 or labels.shape[0] != B or label_lengths.shape != (B,):
            raise RuntimeError("wavenet_speech_amd: ctc labels must be [B, S] with lengths [B]")
        in_len = None if input_lengths is None else input_lengths.to(device=dev, dtype=torch.int64).contiguous()
        lmax = max(int(labels.shape[1]), 1)
        if labels.shape[1] == 0:
            labels = torch.zeros(B, 1, dtype=torch.int64, device=dev)
        with torch.cuda.device(dev):
            ws_bytes = lib.wn_ctc_workspace_bytes(B, C, T, lmax)
            if ws_bytes == 0:
                _lib.check(lib.wn_ctc_loss(None, None, None, None, B, C, T, lmax, blank, None, None, None, 0, None, None),
                           "wn_ctc_loss")
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            nll = torch.empty(B, dtype=torch.float32, device=dev)
            need_grad = ctx.needs_input_grad[0]
            dx = torch.empty_like(x) if need_grad else None
            bad = torch.zeros(1, dtype=torch.int32, device=dev)
            _lib.check(lib.wn_ctc_loss(_p(x), _p(labels), _p(label_lengths), _p(in_len), B, C, T, lmax, int(blank), _p(nll),
                                       _p(dx), _p(ws), ws_bytes, _p(bad), _stream()), "wn_ctc_loss")
            from . import _flags
            _flags.WATCH.poll()
            _flags.WATCH.note(bad, lambda n, C=C, blank=blank: "wavenet_speech_amd: ctc labels outside [0, %d), equal to the blank "
                              "(%d), or lengths out of range in %d utterance(s)" % (C, blank, n), at_once=not need_grad)
        ctx.dx = dx
        return nll.sum()

    @staticmethod
    def backward(ctx, grad_out):
        dx, ctx.dx = ctx.dx, None
        return (None if dx is None else dx * grad_out), None, None, None, None


def ctc_total(transcription, labels, label_lengths, blank=0, input_lengths=None):
    """warp-ctc semantics (Loss.py:49-53): softmax applied internally, negative log likelihoods summed over the batch.
    transcription: [B, labels, T] logits; labels: [B, S] already offset so that 0 is <BLANK>.
    Device tensors go through the HIP kernels (wn_ctc_loss: the logits stay on the GPU and are read in place, where the
    reference copies them to the CPU for warp-ctc every step, pretrain_tnt.py:159); the torch expression below is the CPU
    form used by the tests."""
    if transcription.is_cuda:
        return _CTCFn.apply(transcription, labels, label_lengths, input_lengths, int(blank))
    logp = F.log_softmax(transcription.permute(2, 0, 1), dim=2)          # (T, B, C)
    T, B = logp.shape[0], logp.shape[1]
    in_lengths = torch.full((B,), T, dtype=torch.long) if input_lengths is None else input_lengths.long()
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
