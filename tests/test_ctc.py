"""CTC.  CPU: oracle/ctc_oracle.py against the one known answer the reference holds for warp-ctc
(tests/test_classifier.py:53-59 -> "approximately 2.4628") and against torch's CPU ctc_loss.  GPU: the HIP kernels
(csrc/wn_ctc.hip through the C ABI, wavenet_speech_amd.training.ctc_total) against the oracle: loss and gradient."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ctc_oracle as CO
from wavenet_speech_amd import training as T


def _kat():
    acts = np.array([[0.1, 0.6, 0.1, 0.1, 0.1], [0.1, 0.1, 0.6, 0.1, 0.1]]).T[None]     # [B=1, C=5, T=2]
    return acts, np.array([[1, 2]]), np.array([2])


def _random_case(seed, B, C, Tn, lmax, repeats=False):
    rng = np.random.default_rng(seed)
    acts = rng.normal(size=(B, C, Tn)) * 1.5
    lens = rng.integers(0 if seed % 3 == 0 else 1, lmax + 1, size=B)
    labels = rng.integers(1, 2 if repeats else C, size=(B, lmax))       # repeats: one symbol only -> every neighbour repeats
    return acts, labels, lens


def test_oracle_reproduces_the_reference_known_answer():
    acts, labels, lens = _kat()
    nll, grad = CO.ctc_total(acts, labels, lens)
    assert abs(nll - 2.4628) < 1e-4, nll                 # the reference prints "approximately 2.4628"
    assert abs(grad.sum(axis=1)).max() < 1e-12           # each frame's gradient sums to zero over the classes


@pytest.mark.parametrize("seed,B,C,Tn,lmax,repeats", [(1, 3, 5, 12, 4, False), (2, 2, 7, 30, 9, False), (3, 2, 5, 9, 4, True),
                                                      (4, 1, 4, 3, 3, False), (6, 2, 3, 20, 10, True)])
def test_oracle_agrees_with_torch_cpu_ctc(seed, B, C, Tn, lmax, repeats):
    acts, labels, lens = _random_case(seed, B, C, Tn, lmax, repeats)
    nll, grad = CO.ctc_total(acts, labels, lens)
    x = torch.tensor(acts, dtype=torch.float64, requires_grad=True)
    logp = F.log_softmax(x.permute(2, 0, 1), dim=2)
    ref = F.ctc_loss(logp, torch.tensor(labels), torch.full((B,), Tn, dtype=torch.long), torch.tensor(lens), blank=0,
                     reduction="sum", zero_infinity=False)
    if np.isinf(nll):
        assert torch.isinf(ref)                           # no alignment fits (e.g. repeats need separating blanks)
        return
    assert abs(float(ref) - nll) < 1e-9 * max(1.0, abs(nll))
    ref.backward()
    assert np.abs(x.grad.numpy() - grad).max() < 1e-9


def test_cpu_form_of_ctc_total_is_the_same_quantity():
    acts, labels, lens = _random_case(5, 3, 5, 25, 6)
    got = T.ctc_total(torch.tensor(acts, dtype=torch.float32), torch.tensor(labels), torch.tensor(lens))
    assert abs(float(got) - CO.ctc_total(acts, labels, lens)[0]) < 1e-4 * abs(float(got))


@pytest.mark.gpu
def test_hip_ctc_known_answer_on_the_device():
    acts, labels, lens = _kat()
    x = torch.tensor(acts, dtype=torch.float32, device="cuda:0", requires_grad=True)
    loss = T.ctc_total(x, torch.tensor(labels, device="cuda:0"), torch.tensor(lens, device="cuda:0"))
    assert abs(float(loss) - 2.4628) < 1e-4
    loss.backward()
    assert np.abs(x.grad.cpu().numpy() - CO.ctc_total(acts, labels, lens)[1]).max() < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("seed,B,C,Tn,lmax,repeats", [(1, 3, 5, 12, 4, False), (2, 2, 7, 30, 9, False), (3, 2, 5, 9, 4, True),
                                                      (4, 1, 4, 3, 3, False), (6, 2, 3, 20, 10, True), (7, 4, 5, 200, 40, False),
                                                      (8, 2, 64, 70, 33, False), (9, 3, 5, 130, 64, False),
                                                      (10, 1, 2, 1, 1, False), (12, 2, 3, 2, 1, True), (13, 1, 5, 1, 3, False)])
def test_hip_ctc_matches_the_oracle(seed, B, C, Tn, lmax, repeats):
    acts, labels, lens = _random_case(seed, B, C, Tn, lmax, repeats)
    acts = acts.astype(np.float32).astype(np.float64)                   # the device reads fp32 activations
    nll_ref, grad_ref = 0.0, np.zeros_like(acts)
    per = []
    for b in range(B):
        n, g = CO.ctc_nll_and_grad(acts[b], [int(v) for v in labels[b][:lens[b]]])
        per.append(n)
        grad_ref[b] = g
    x = torch.tensor(acts, dtype=torch.float32, device="cuda:0", requires_grad=True)
    loss = T.ctc_total(x, torch.tensor(labels, device="cuda:0"), torch.tensor(lens, device="cuda:0"))
    total = float(np.sum(per))
    if np.isinf(total):
        assert torch.isinf(loss)
    else:
        assert abs(float(loss) - total) < 2e-6 * max(1.0, abs(total)), (float(loss), total)
    loss.backward()
    g = x.grad.cpu().numpy()
    assert np.isfinite(g).all()
    assert np.abs(g - grad_ref).max() < 2e-6, np.abs(g - grad_ref).max()
    # an upstream factor scales the gradient (the training step divides by the frame count, Loss.py:53)
    x2 = x.detach().clone().requires_grad_(True)
    (T.ctc_total(x2, torch.tensor(labels, device="cuda:0"), torch.tensor(lens, device="cuda:0")) / 7.0).backward()
    if np.isfinite(total):
        assert np.abs(x2.grad.cpu().numpy() * 7.0 - g).max() < 1e-6


@pytest.mark.gpu
def test_hip_ctc_input_lengths_bad_labels_and_cfg2_shape():
    dev = "cuda:0"
    # per-utterance frame counts: frames past the length get no gradient, the loss is that of the truncated utterance
    acts, labels, lens = _random_case(11, 3, 5, 40, 6)
    acts = acts.astype(np.float32).astype(np.float64)
    in_len = np.array([40, 25, 31])
    ref, gref = CO.ctc_total(acts, labels, lens, input_lengths=in_len)
    x = torch.tensor(acts, dtype=torch.float32, device=dev, requires_grad=True)
    loss = T.ctc_total(x, torch.tensor(labels, device=dev), torch.tensor(lens, device=dev), input_lengths=torch.tensor(in_len, device=dev))
    assert abs(float(loss) - ref) < 2e-6 * abs(ref)
    loss.backward()
    assert np.abs(x.grad.cpu().numpy() - gref).max() < 2e-6
    assert float(x.grad[1, :, 25:].abs().max()) == 0.0
    # a label equal to the blank (or outside the classes) is refused, not read out of bounds
    bad = torch.tensor(labels, device=dev)
    bad[0, 0] = 0
    lens_t = torch.tensor(np.maximum(lens, 1), device=dev)
    with pytest.raises(RuntimeError, match="ctc labels"):
        T.ctc_total(x.detach(), bad, lens_t)
    bad[0, 0] = 9
    with pytest.raises(RuntimeError, match="ctc labels"):
        T.ctc_total(x.detach(), bad, lens_t)
    # BASELINE configs[1]'s shape: 32 utterances x 5 labels x 4098 frames, ~400 bases each; against torch's float64 CPU ctc_loss
    g = torch.Generator().manual_seed(3)
    B, C, Tn, lmax = 32, 5, 4098, 420
    x = (torch.randn(B, C, Tn, generator=g) * 2.0)
    labels = torch.randint(1, C, (B, lmax), generator=g)
    lens = torch.randint(300, lmax + 1, (B,), generator=g)
    xd = x.double().requires_grad_(True)
    ref = F.ctc_loss(F.log_softmax(xd.permute(2, 0, 1), dim=2), labels, torch.full((B,), Tn, dtype=torch.long), lens, blank=0,
                     reduction="sum")
    ref.backward()
    xg = x.to(dev).requires_grad_(True)
    loss = T.ctc_total(xg, labels.to(dev), lens.to(dev))
    assert abs(float(loss) - float(ref)) < 1e-6 * abs(float(ref)), (float(loss), float(ref))
    loss.backward()
    err = float((xg.grad.cpu().double() - xd.grad).abs().max())
    assert err < 5e-6, err
    # deterministic: the same call gives the same bits
    xg2 = x.to(dev).requires_grad_(True)
    T.ctc_total(xg2, labels.to(dev), lens.to(dev)).backward()
    assert torch.equal(xg2.grad, xg.grad)
