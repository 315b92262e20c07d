"""
CPU: pin the oracle (oracle/wavenet_oracle.py) to the golden vectors captured from the
reference's own modules (tests/golden/make_golden.py).  Tolerance: the path is fp32
floating point; north_star allows 1e-4 relative, the oracle is held to 2e-5 here
(observed ~1e-6; summation order differs from ATen's conv).
"""
import pytest
import torch

from oracle import wavenet_oracle as O
from tests import goldenio

TOL = 2e-5


def _leafify(sd):
    return {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}


def _check_grads(g, sd, x, tol=TOL):
    for k, ref in g.grads.items():
        if not g.hasgrad[k]:
            continue
        got = sd[k].grad
        assert got is not None, k
        assert O.rel_err(got, ref) < tol, (g.name, k, O.rel_err(got, ref))
    if "x" in g.grad_inputs:
        assert O.rel_err(x.grad, g.grad_inputs["x"]) < tol


@pytest.mark.parametrize("name", goldenio.names("conv_"))
@pytest.mark.parametrize("impl", ["taps", "aten"])
def test_conv(name, impl):
    g = goldenio.load(name)
    m = g.meta
    sd = _leafify(g.sd)
    x = g.inputs["x"].clone().requires_grad_(True)
    y = O.dilated_conv(x, sd["conv1d.weight"], sd["conv1d.bias"], m["d"], m["causal"], impl)
    assert y.shape == g.outs[0].shape
    assert O.rel_err(y, g.outs[0]) < TOL
    (y * g.cots[0]).sum().backward()
    _check_grads(g, sd, x)


@pytest.mark.parametrize("name", goldenio.names("block_"))
def test_block_forward_backward(name):
    g = goldenio.load(name)
    m = g.meta
    sd = _leafify(g.sd)
    x = g.inputs["x"].clone().requires_grad_(True)
    r, s = O.residual_block(x, sd, m["d"], m["causal"])
    assert O.rel_err(r, g.outs[0]) < TOL and O.rel_err(s, g.outs[1]) < TOL
    ((r * g.cots[0]).sum() + (s * g.cots[1]).sum()).backward()
    _check_grads(g, sd, x)


@pytest.mark.parametrize("name", goldenio.names("block_"))
def test_block_manual_backward_matches_reference(name):
    """The hand-derived backward (the formulae the HIP kernels implement)."""
    g = goldenio.load(name)
    m = g.meta
    with torch.no_grad():
        dx, grads = O.residual_block_backward(g.inputs["x"], g.sd, m["d"], m["causal"], g.cots[0], g.cots[1])
    assert O.rel_err(dx, g.grad_inputs["x"]) < TOL
    for k, ref in g.grads.items():
        assert O.rel_err(grads[k], ref) < TOL, (name, k)


def test_block_manual_backward_fp64_gradcheck():
    torch.manual_seed(3)
    p = {k: torch.randn(s, dtype=torch.float64) * 0.3 for k, s in [
        ("conv_tanh.conv1d.weight", (5, 4, 3)), ("conv_tanh.conv1d.bias", (5,)),
        ("conv_sigmoid.conv1d.weight", (5, 4, 3)), ("conv_sigmoid.conv1d.bias", (5,)),
        ("conv1x1_residual.weight", (5, 5, 1)), ("conv1x1_residual.bias", (5,)),
        ("conv1x1_skip.weight", (5, 5, 1)), ("conv1x1_skip.bias", (5,)),
        ("residual_proj.weight", (5, 4)), ("residual_proj.bias", (5,))]}
    for causal in (True, False):
        x = torch.randn(2, 4, 23, dtype=torch.float64)
        dr, ds = torch.randn(2, 5, 23, dtype=torch.float64), torch.randn(2, 5, 23, dtype=torch.float64)
        pl = _leafify(p)
        xl = x.clone().requires_grad_(True)
        r, s = O.residual_block(xl, pl, 3, causal)
        ((r * dr).sum() + (s * ds).sum()).backward()
        dx, grads = O.residual_block_backward(x, p, 3, causal, dr, ds)
        assert O.rel_err(dx, xl.grad) < 1e-12
        for k in p:
            assert O.rel_err(grads[k], pl[k].grad) < 1e-12, k


@pytest.mark.parametrize("name", goldenio.names("wavenet_"))
def test_wavenet(name):
    g = goldenio.load(name)
    m = g.meta
    sd = _leafify(g.sd)
    x = g.inputs["x"].clone().requires_grad_(True)
    y = O.wavenet(x, sd, m["layers"], m["softmax"])
    assert y.shape == g.outs[0].shape
    assert O.rel_err(y, g.outs[0]) < TOL
    (y * g.cots[0]).sum().backward()
    _check_grads(g, sd, x, tol=1e-4)


@pytest.mark.parametrize("name", goldenio.names("rawctc_"))
def test_raw_ctcnet(name):
    g = goldenio.load(name)
    m = g.meta
    sd = _leafify(g.sd)
    x = g.inputs["x"].clone().requires_grad_(True)
    y = O.raw_ctcnet(x, sd, m["layers"], m["feature_kwidth"], m["input_dilation"], m["positions"],
                     m["softmax"], m["causal"])
    assert y.shape == g.outs[0].shape == (m["B"], m["num_labels"], m["L"] + m["feature_kwidth"] - 1)
    assert O.rel_err(y, g.outs[0]) < TOL
    (y * g.cots[0]).sum().backward()
    _check_grads(g, sd, x, tol=1e-4)


@pytest.mark.parametrize("name", goldenio.names("classifier_"))
def test_classifier(name):
    g = goldenio.load(name)
    m = g.meta
    sd = _leafify(g.sd)
    x = g.inputs["x"].clone().requires_grad_(True)
    y = O.wavenet_classifier(x, sd, m["layers"], m["pool_kernel_size"], m["input_dilation"], m["softmax"])
    assert O.rel_err(y, g.outs[0]) < TOL
    (y * g.cots[0]).sum().backward()
    _check_grads(g, sd, x, tol=1e-4)


def test_tap_offsets_match_survey_probe():
    # probed tap offsets of the reference's non-causal conv
    assert O.tap_offsets(2, 1, False) == [-1, 0]
    assert O.tap_offsets(2, 2, False) == [-1, 1]
    assert O.tap_offsets(2, 3, False) == [-2, 1]
    assert O.tap_offsets(2, 4, False) == [-2, 2]
    assert O.tap_offsets(2, 512, False) == [-256, 256]
    assert O.tap_offsets(3, 5, False) == [-5, 0, 5]
    assert O.tap_offsets(2, 7, True) == [-7, 0]
    assert O.tap_offsets(3, 2, True) == [-4, -2, 0]
