"""GPU: the stand-alone conv of the half-precision modes (wn_hconv_* through functional.dilated_conv(..., precision=...))
against the oracle's conv, forward and all gradients; f16x3 at the fp32 path's 1e-4 bar."""
import pytest
import torch

from oracle import wavenet_oracle as O
from wavenet_speech_amd import functional as HF

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("case", [(2, 32, 32, 1, 1, True, 300), (2, 32, 32, 2, 1, True, 300), (1, 24, 40, 3, 5, False, 130),
                                  (3, 1, 48, 3, 1, True, 77), (2, 256, 256, 1, 1, True, 512), (1, 64, 5, 1, 1, True, 1000),
                                  (2, 40, 72, 2, 300, True, 260), (1, 8, 8, 2, 1, True, 1), (2, 16, 24, 3, 2, False, 7)])
@pytest.mark.parametrize("precision", ["f16x3", "f16", "bf16"])
def test_half_conv_matches_the_oracle(case, precision):
    B, Ci, Co, k, d, causal, L = case
    g = torch.Generator().manual_seed(B * 1000 + Ci + Co + k + L)
    x = torch.randn(B, Ci, L, generator=g) * 3.0
    w = torch.randn(Co, Ci, k, generator=g) * (1.0 / (Ci * k) ** 0.5)
    b = torch.randn(Co, generator=g) * 0.1
    cot = torch.randn(B, Co, L, generator=g)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    y0 = O.dilated_conv(xr, wr, br, d, causal)
    (y0 * cot).sum().backward()
    xg, wg, bg = (t.clone().to(DEV).requires_grad_(True) for t in (x, w, b))
    y1 = HF.dilated_conv(xg, wg, bg, d, causal, precision)
    (y1 * cot.to(DEV)).sum().backward()
    tol = {"f16x3": 1e-4, "f16": 4e-3, "bf16": 3e-2}[precision]
    errs = {"y": O.rel_err(y1.detach().cpu(), y0), "dx": O.rel_err(xg.grad.cpu(), xr.grad),
            "dw": O.rel_err(wg.grad.cpu(), wr.grad), "db": O.rel_err(bg.grad.cpu(), br.grad)}
    print(precision, case, {k_: "%.1e" % v for k_, v in errs.items()})
    assert max(errs.values()) < tol, errs
    # inference call: same forward
    with torch.no_grad():
        y2 = HF.dilated_conv(xg, wg, bg, d, causal, precision)
    assert torch.equal(y2, y1.detach())
