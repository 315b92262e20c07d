#!/usr/bin/env python3
"""
Generate the golden input/output vectors under tests/golden/ from the REFERENCE's own
modules (paultsw/wavenet-speech, mounted read-only at /root/reference).

Run in the build container only:   python tests/golden/make_golden.py
The reference never travels to the GPU box; only the small .npz fixtures written here do.
Each fixture holds: inputs, the reference module's state_dict, its forward outputs, and
the gradients autograd produced for a fixed random cotangent
(loss = sum(out * cot)), w.r.t. the input and every parameter.

Nothing from the reference is copied: the modules are imported, run, and their numbers saved.
"""
import json
import os
import sys
import warnings

import numpy as np
import torch

REF = os.environ.get("WN_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
warnings.filterwarnings("ignore")

from modules.block import ResidualBlock  # noqa: E402
from modules.classifier import WaveNetClassifier  # noqa: E402
from modules.conv_ops import CausalConv1d, NonCausalConv1d  # noqa: E402
from modules.raw_ctcnet import RawCTCNet  # noqa: E402
from modules.wavenet import WaveNet  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def _np(t):
    return t.detach().cpu().numpy().copy()


def run_and_save(name, module, inputs, meta, perturb_bias=True, seed=0):
    """inputs: dict name->tensor (first one requires grad). Saves fixture `name`.npz."""
    g = torch.Generator().manual_seed(1000 + seed)
    if perturb_bias:
        # the reference zero-initialises most biases; make them non-trivial so parity sees them
        with torch.no_grad():
            for p in module.parameters():
                if p.dim() == 1:
                    p.add_(0.1 * torch.randn(p.shape, generator=g))
    xs = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in inputs.items()}
    outs = module(*xs.values())
    if not isinstance(outs, tuple):
        outs = (outs,)
    cots = [torch.randn(o.shape, generator=g) for o in outs]
    loss = sum((o * c).sum() for o, c in zip(outs, cots))
    loss.backward()
    blob = {"meta": np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)}
    for k, v in xs.items():
        blob["in." + k] = _np(v)
        if v.grad is not None:
            blob["grad_in." + k] = _np(v.grad)
    for i, (o, c) in enumerate(zip(outs, cots)):
        blob["out.%d" % i] = _np(o)
        blob["cot.%d" % i] = _np(c)
    for k, v in module.state_dict().items():
        blob["sd." + k] = _np(v)
    for k, p in module.named_parameters():
        blob["grad." + k] = _np(p.grad) if p.grad is not None else np.zeros(tuple(p.shape), np.float32)
        blob["hasgrad." + k] = np.array(p.grad is not None)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **blob)
    print("%-40s %8.1f KB" % (name, os.path.getsize(path) / 1024))


def main():
    torch.manual_seed(20260101)
    torch.set_num_threads(4)

    # ---- conv ops (tests/test_conv_ops.py uses k=5, d=3) --------------------------------
    conv_cases = [
        # cin, cout, k, d, causal, L, B
        (3, 4, 5, 3, True, 20, 2),
        (3, 4, 5, 3, False, 20, 2),
        (4, 4, 2, 1, True, 16, 1),
        (4, 6, 2, 3, False, 17, 2),   # even k, odd d: asymmetric taps [-2,+1]
        (4, 6, 2, 4, False, 17, 2),   # [-2,+2]
        (5, 3, 3, 2, False, 19, 2),   # k=3 symmetric
        (4, 4, 2, 32, True, 10, 2),   # dilation >= L
    ]
    for i, (ci, co, k, d, causal, L, B) in enumerate(conv_cases):
        cls = CausalConv1d if causal else NonCausalConv1d
        m = cls(ci, co, k, dilation=d)
        x = torch.randn(B, ci, L)
        run_and_save("conv_%02d" % i, m, {"x": x},
                     dict(kind="conv", cin=ci, cout=co, k=k, d=d, causal=causal, L=L, B=B), seed=i)

    # ---- residual blocks (tests/test_block.py: 4->5 ch, k=2, d=2, L=12, B=3) -------------
    block_cases = [
        (4, 5, 2, 2, True, 12, 3),
        (4, 5, 2, 2, False, 12, 3),
        (8, 8, 2, 1, True, 33, 2),
        (8, 8, 2, 3, False, 40, 2),
        (6, 7, 3, 4, False, 50, 2),
        (6, 7, 3, 4, True, 50, 2),
        (4, 4, 2, 16, True, 10, 2),        # d >= L: the shifted tap reads only zeros
        (16, 16, 2, 512, True, 1100, 1),   # the largest dilation of the 1..512 cycle
        (32, 32, 2, 8, True, 300, 2),
        (40, 24, 2, 5, False, 131, 1),     # channel counts that are not multiples of 8/32
    ]
    for i, (ci, co, k, d, causal, L, B) in enumerate(block_cases):
        m = ResidualBlock(ci, co, k, d, causal=causal)
        x = torch.randn(B, ci, L)
        run_and_save("block_%02d" % i, m, {"x": x},
                     dict(kind="block", cin=ci, cout=co, k=k, d=d, causal=causal, L=L, B=B), seed=100 + i)

    # ---- WaveNet ------------------------------------------------------------------------
    # config 1 (BASELINE.json configs[0]): tiny WaveNet, 32 ch, 5 dilated blocks, batch 1
    layers = [(32, 32, 2, d) for d in (1, 2, 4, 8, 16)]
    for sm in (False, True):
        m = WaveNet(32, 2, layers, 32, softmax=sm)
        q = torch.randint(0, 32, (1, 200))
        x = torch.zeros(1, 32, 200).scatter_(1, q.unsqueeze(1), 1.)
        run_and_save("wavenet_cfg1_softmax%d" % int(sm), m, {"x": x},
                     dict(kind="wavenet", in_dim=32, entry_kwidth=2, layers=layers, out_dim=32, softmax=sm,
                          L=200, B=1), seed=200 + int(sm))
    # tests/test_wavenet.py shape: 11-dim, dilations cycling 1..512 with L=14 (d >> L), B=5 (fewer blocks here)
    layers = [(11, 11, 2, 2 ** (i % 10)) for i in range(12)]
    m = WaveNet(11, 2, layers, 11, softmax=False)
    run_and_save("wavenet_small_L14", m, {"x": torch.randn(5, 11, 14)},
                 dict(kind="wavenet", in_dim=11, entry_kwidth=2, layers=layers, out_dim=11, softmax=False,
                      L=14, B=5), seed=210)
    # mixed channel widths and out_dim != C
    layers = [(8, 16, 2, 1), (16, 16, 2, 2), (16, 24, 2, 4), (24, 8, 3, 8)]
    m = WaveNet(6, 3, layers, 12, softmax=False)
    run_and_save("wavenet_mixed", m, {"x": torch.randn(2, 6, 77)},
                 dict(kind="wavenet", in_dim=6, entry_kwidth=3, layers=layers, out_dim=12, softmax=False,
                      L=77, B=2), seed=220)

    # ---- RawCTCNet ----------------------------------------------------------------------
    layers = [(16, 16, 2, d) for d in (1, 2, 4, 8)]
    for i, (causal, positions, sm) in enumerate([(False, False, False), (True, False, True), (False, True, False)]):
        m = RawCTCNet(16, 3, 5, layers, 16, input_kernel_size=2, input_dilation=1,
                      positions=positions, softmax=sm, causal=causal)
        run_and_save("rawctc_%02d" % i, m, {"x": torch.randn(2, 1, 90)},
                     dict(kind="rawctc", num_features=16, feature_kwidth=3, num_labels=5, layers=layers, out_dim=16,
                          input_kernel_size=2, input_dilation=1, positions=positions, softmax=sm, causal=causal,
                          L=90, B=2), perturb_bias=False, seed=300 + i)

    # ---- WaveNetClassifier --------------------------------------------------------------
    layers = [(16, 16, 2, d) for d in (1, 2, 4)]
    m = WaveNetClassifier(12, 5, layers, 16, pool_kernel_size=3, input_kernel_size=2, input_dilation=1, softmax=False)
    run_and_save("classifier_00", m, {"x": torch.randn(2, 12, 100)},
                 dict(kind="classifier", in_dim=12, num_labels=5, layers=layers, out_dim=16, pool_kernel_size=3,
                      input_kernel_size=2, input_dilation=1, softmax=False, L=100, B=2), seed=400)


if __name__ == "__main__":
    main()
