"""The synthetic signal generator (wavenet_speech_amd/synthetic.py) and, on the GPU, BASELINE.json configs[0]: the
reference's tests/wavenet_overfit_test.py scenario (tiny WaveNet, one synthetic one-hot signal, per-step NLL, Adadelta
with ReduceLROnPlateau) -- 'you should see a decreasing loss'."""
import numpy as np
import pytest
import torch

from wavenet_speech_amd import synthetic as S


def test_generator_shapes_and_determinism():
    g = torch.Generator().manual_seed(5)
    levels, one_hot, bases = S.gaussian_kmer_signal(3, 100, num_levels=32, generator=g)
    assert levels.shape == (3, 100) and one_hot.shape == (3, 32, 100) and bases.shape[0] == 3
    assert int(levels.min()) >= 0 and int(levels.max()) <= 31
    assert torch.equal(one_hot.sum(1), torch.ones(3, 100)) and torch.equal(one_hot.argmax(1), levels)
    assert int(bases.min()) >= 1 and int(bases.max()) <= 4
    g2 = torch.Generator().manual_seed(5)
    levels2, _, _ = S.gaussian_kmer_signal(3, 100, num_levels=32, generator=g2)
    assert torch.equal(levels, levels2)
    # each k-mer is held for `upsampling` samples: the signal is piecewise stationary, so neighbouring samples inside a
    # segment are closer than across the whole read
    lv = levels.float()
    assert float((lv[:, 1:] - lv[:, :-1]).abs().mean()) < float((lv - lv.mean(1, keepdim=True)).abs().mean()) * 1.5


def test_quantisation_matches_numpy_digitize():
    x = torch.linspace(-0.999, 0.999, 1001)
    mapped = S.mu_law(x, 256.0)
    edges = torch.linspace(-1.0, 1.0, 256)
    ours = torch.bucketize(mapped, edges, right=True)
    ref = np.digitize(mapped.numpy(), edges.numpy())              # utils/gaussian_kmer_model.py:86
    assert np.array_equal(ours.numpy(), ref)
    assert torch.all(mapped[1:] >= mapped[:-1])                    # mu-law is monotone


def _check_against_reference_fixture(device):
    """tests/golden/generator_00.npz: outputs of the reference's own k-mer window, quantize_fn and one_hot_fn"""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "generator_00.npz"), allow_pickle=False)
    table = (torch.from_numpy(z["table.means"]).to(device), torch.from_numpy(z["table.stdvs"]).to(device))
    for case in range(3):
        pre = "case%d." % case
        ups, nl = int(z[pre + "upsampling"]), int(z[pre + "num_levels"])
        bases = torch.from_numpy(z[pre + "bases"]).to(device)
        kmers = S.kmer_indices(bases, ups)
        assert np.array_equal(kmers.cpu().numpy(), z[pre + "kmer_seq"]), "k-mer window / trim / upsampling"
        assert np.array_equal(table[0][kmers].cpu().numpy(), z[pre + "kmer_means"])
        pico = table[0][kmers] + table[1][kmers] * torch.from_numpy(z[pre + "noise"]).to(device)
        assert np.allclose(pico.cpu().numpy(), z[pre + "picoamps"], rtol=1e-14, atol=0)
        q = S.quantize(torch.from_numpy(z[pre + "picoamps"]).to(device), nl)
        assert np.array_equal(q.cpu().numpy(), z[pre + "quantized"]), "normalise / mu-law / digitize"
        oh = S.one_hot(q, nl)
        assert oh.dtype == torch.float32 and np.array_equal(oh.cpu().numpy(), z[pre + "one_hot"])


def test_deterministic_stages_match_the_reference_generator():
    _check_against_reference_fixture("cpu")


def test_gaussian_stage_statistics():
    """the one random stage: per k-mer the samples have the table's mean and standard deviation"""
    means, stdvs = S.standin_kmer_table()
    kmers = torch.tensor([5, 700, 1023]).repeat_interleave(20000)
    x = S.gaussian_picoamps(kmers, (means.double(), stdvs.double()), torch.Generator().manual_seed(3)).view(3, 20000)
    for i, k in enumerate((5, 700, 1023)):
        assert abs(float(x[i].mean()) - float(means[k])) < 0.1 and abs(float(x[i].std()) - float(stdvs[k])) < 0.1


@pytest.mark.gpu
def test_generator_on_the_device_matches_the_reference_fixture_and_draws_there():
    _check_against_reference_fixture("cuda:0")
    g = torch.Generator(device="cuda:0").manual_seed(7)
    levels, oh, bases = S.gaussian_kmer_signal(2, 300, generator=g, device="cuda:0")
    assert levels.is_cuda and oh.is_cuda and bases.is_cuda
    g2 = torch.Generator(device="cuda:0").manual_seed(7)
    levels2, _, _ = S.gaussian_kmer_signal(2, 300, generator=g2, device="cuda:0")
    assert torch.equal(levels, levels2)
    assert int(levels.min()) >= 1 and int(levels.max()) <= 255


@pytest.mark.gpu
def test_hip_generator_matches_the_reference_fixture():
    """csrc/wn_synth.hip through the C ABI against tests/golden/generator_00.npz (outputs of the reference's own
    gaussian_model_fn / quantize_fn / one_hot_fn): k-mer window and lookup exactly, the signal with the fixture's noise to
    1e-14, quantisation and one-hot of the fixture's signal exactly"""
    import os
    dev = "cuda:0"
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "generator_00.npz"), allow_pickle=False)
    table = (torch.from_numpy(z["table.means"]).to(dev), torch.from_numpy(z["table.stdvs"]).to(dev))
    for case in range(3):
        pre = "case%d." % case
        ups, nl = int(z[pre + "upsampling"]), int(z[pre + "num_levels"])
        bases = torch.from_numpy(z[pre + "bases"]).to(dev).long().reshape(1, -1)
        L = int(z[pre + "kmer_seq"].shape[0])
        zero = torch.zeros(1, L, dtype=torch.float64, device=dev)
        _, _, pico0 = S.hip_signal(bases, L, nl, ups, table, noise=zero, want_one_hot=False)
        assert np.array_equal(pico0.cpu().numpy()[0], z[pre + "kmer_means"]), "k-mer window / trim / upsampling / lookup"
        noise = torch.from_numpy(z[pre + "noise"]).to(dev).reshape(1, L)
        _, _, pico = S.hip_signal(bases, L, nl, ups, table, noise=noise, want_one_hot=False)
        assert np.allclose(pico.cpu().numpy()[0], z[pre + "picoamps"], rtol=1e-14, atol=0)
        given = torch.from_numpy(z[pre + "picoamps"]).to(dev).reshape(1, L)
        levels, oh, _ = S.hip_signal(bases, L, nl, ups, table, picoamps=given)
        assert np.array_equal(levels.cpu().numpy()[0], z[pre + "quantized"]), "normalise / mu-law / digitize"
        assert oh.dtype == torch.float32 and np.array_equal(oh.cpu().numpy()[0], z[pre + "one_hot"])


@pytest.mark.gpu
def test_hip_generator_random_stages_and_full_size():
    dev = "cuda:0"
    # nucleotides: uniform over 1..4, reproducible from the seed
    b1 = S.hip_bases(8, 50000, 1234, dev)
    assert torch.equal(b1, S.hip_bases(8, 50000, 1234, dev)) and not torch.equal(b1, S.hip_bases(8, 50000, 1235, dev))
    assert int(b1.min()) == 1 and int(b1.max()) == 4
    counts = torch.bincount(b1.flatten(), minlength=5)[1:].double() / b1.numel()
    assert float((counts - 0.25).abs().max()) < 0.005
    # Gaussian stage: one k-mer held for the whole read -> the samples have the table's mean and standard deviation,
    # no skew, normal kurtosis
    means, stdvs = S.standin_kmer_table(device=dev)
    for kmer_bases in ((1, 1, 1, 1, 1), (4, 2, 3, 1, 4)):
        bases = torch.tensor([kmer_bases[0]] * 2 + list(kmer_bases) + [kmer_bases[-1]] * 2, device=dev).repeat(1, 1)
        k = sum((nt - 1) * w for nt, w in zip(kmer_bases, S.KMER_WEIGHTS))
        _, _, x = S.hip_signal(bases, 200000, 256, 200000, (means, stdvs), seed=99, want_one_hot=False)
        x = x[0]
        zs = (x - float(means[k])) / float(stdvs[k])
        assert abs(float(zs.mean())) < 0.01 and abs(float(zs.std()) - 1.0) < 0.01
        assert abs(float((zs ** 3).mean())) < 0.03 and abs(float((zs ** 4).mean()) - 3.0) < 0.08
        assert abs(float((zs[1:] * zs[:-1]).mean())) < 0.01                 # neighbouring samples are independent
    # the whole generator at the bench shape: reproducible, in range, one-hot consistent, and the HIP quantiser agrees
    # with the torch restatement on the same signal (the per-read mean is summed in another order: an exact tie may move)
    g = torch.Generator(device=dev).manual_seed(7)
    levels, oh, bases = S.gaussian_kmer_signal(16, 16000, generator=g, device=dev)
    g2 = torch.Generator(device=dev).manual_seed(7)
    levels2, _, _ = S.gaussian_kmer_signal(16, 16000, generator=g2, device=dev, want_one_hot=False)
    assert torch.equal(levels, levels2)
    assert int(levels.min()) >= 1 and int(levels.max()) <= 255 and oh.shape == (16, 256, 16000)
    assert torch.equal(oh.argmax(1), levels) and float(oh.sum()) == 16 * 16000
    _, _, pico = S.hip_signal(bases, 16000, 256, 3, None, seed=5, want_one_hot=False)
    lv_hip, _, _ = S.hip_signal(bases, 16000, 256, 3, None, picoamps=pico, want_one_hot=False)
    lv_ref = S.quantize(pico, 256)
    diff = (lv_hip - lv_ref).abs()
    assert int(diff.max()) <= 1 and int((diff != 0).sum()) <= 3, (int(diff.max()), int((diff != 0).sum()))
    kmers = S.kmer_indices(bases, 3)[:, :16000]
    assert float((pico - means.double()[kmers]).abs().max()) < 6.5 * float(stdvs.max())   # every sample within 6.5 sigma
    # smallest and ragged shapes: one read of two samples (a span of zero would divide by zero, as in the reference), a length
    # that is not a multiple of the upsampling or of the kernel's tile, no one-hot
    for B_, L_ in ((1, 2), (3, 7), (2, 257)):
        lv, oh_, bs = S.gaussian_kmer_signal(B_, L_, generator=torch.Generator(device=dev).manual_seed(L_), device=dev)
        assert lv.shape == (B_, L_) and oh_.shape == (B_, 256, L_) and int(lv.min()) >= 0 and int(lv.max()) <= 255
        assert torch.equal(oh_.argmax(1), lv)
        _, _, pc = S.hip_signal(bs, L_, 256, 3, None, seed=1, want_one_hot=False)
        lv2, _, _ = S.hip_signal(bs, L_, 256, 3, None, picoamps=pc, want_one_hot=False)
        assert int((lv2 - S.quantize(pc, 256).clamp(0, 255)).abs().max()) <= 1


@pytest.mark.gpu
def test_config0_wavenet_overfit_script():
    from torch.optim.lr_scheduler import ReduceLROnPlateau
    from wavenet_speech_amd import training as T
    from wavenet_speech_amd.modules.wavenet import WaveNet
    dev = "cuda:0"
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(1)
    _, signal, _ = S.gaussian_kmer_signal(1, 401, num_levels=32, generator=g, device=dev)   # generated on the GPU, RNG included
    source, target = signal[:, :, :-1].contiguous(), signal[:, :, 1:].argmax(1)
    net = WaveNet(32, 2, [(32, 32, 2, d) for d in (1, 2, 4, 8, 16)], 32, softmax=False).to(dev)
    opt = torch.optim.Adadelta(net.parameters(), lr=1.0, rho=0.9, weight_decay=1e-4)   # wavenet_overfit_test.py:33-37
    sched = ReduceLROnPlateau(opt, patience=5)
    losses = []
    for step in range(80):
        opt.zero_grad()
        loss = T.sequence_nll(net(source), target)               # the reference sums CE over time steps (:49-50)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()) / target.shape[1])
        if step % 20 == 0:
            sched.step(losses[-1])
    assert losses[0] > 2.5                                        # ~ln(32) at initialisation
    assert losses[-1] < 0.7 * losses[0], (losses[0], losses[-1])
