"""
CPU ORACLE for the WaveNet dilated residual-block path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch CPU restatement (plain torch tensor ops, fp32 or
fp64) of the arithmetic of the reference's hot path.  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import
it, and there only as the checker -- never as the thing measured or shipped.
The product path (`wavenet_speech_amd`) never imports this module and raises
when its HIP library is missing.

Parity is PINNED: every function below is checked in
`tests/test_oracle_golden.py` against golden vectors captured from the
reference's own modules (imported from /root/reference in the build container
by `tests/golden/make_golden.py`; the reference holds no numeric fixtures of
its own for this path -- only shape asserts).

All file:line citations are relative to the reference repository
(paultsw/wavenet-speech).

Parameters are addressed by the reference's state_dict key names, e.g.
``convolutions.3.conv_tanh.conv1d.weight`` so that a reference checkpoint is a
valid input to every function here.
"""
import math

import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------
# conv_ops
# --------------------------------------------------------------------------


def autopad(k, d):
    """Padding used by the non-causal conv.  modules/conv_ops.py:104-116:
    total = (k-1)*d; odd totals round UP ((total-1)/2 + 1), even totals halve."""
    total = (k - 1) * d
    return (total - 1) // 2 + 1 if total % 2 == 1 else total // 2


def tap_offsets(k, d, causal):
    """Time offset of tap j relative to the output sample: y[t] = sum_j W[:,:,j] x[t + off_j].

    causal  (modules/conv_ops.py:16-44): Conv1d(padding=(k-1)d both sides) then keep the
            first L outputs  =>  off_j = (j - (k-1)) * d   (k=2: [-d, 0]).
    non-causal (modules/conv_ops.py:51-79): padding p = autopad(k,d) both sides, keep the
            first L outputs  =>  off_j = j*d - p           (k=2,d=2: [-1,+1]; k=2,d=3: [-2,+1]).
    """
    p = (k - 1) * d if causal else autopad(k, d)
    return [j * d - p for j in range(k)]


def shifted(x, off):
    """x[..., t + off] with zeros outside [0, L)."""
    L = x.shape[-1]
    if off == 0:
        return x
    out = torch.zeros_like(x)
    if abs(off) >= L:
        return out
    if off < 0:
        out[..., -off:] = x[..., : L + off]
    else:
        out[..., : L - off] = x[..., off:]
    return out


def dilated_conv(x, weight, bias, d, causal, impl="taps"):
    """CausalConv1d / NonCausalConv1d forward (modules/conv_ops.py:39-44, 73-79).

    impl="taps": explicit restatement  y[t] = b + sum_j W[:,:,j] @ x[t + off_j]
    impl="aten": the ATen op the reference itself calls (conv1d with symmetric
                 padding, then slice to the first L steps) -- used for the CPU
                 baseline timing and as a cross-check of "taps".
    """
    k = weight.shape[2]
    L = x.shape[2]
    if impl == "aten":
        p = (k - 1) * d if causal else autopad(k, d)
        return F.conv1d(x, weight, bias, stride=1, padding=p, dilation=d)[:, :, 0:L]
    y = None
    for j, off in enumerate(tap_offsets(k, d, causal)):
        term = torch.einsum("oc,bcl->bol", weight[:, :, j], shifted(x, off))
        y = term if y is None else y + term
    if bias is not None:
        y = y + bias.view(1, -1, 1)
    return y


# --------------------------------------------------------------------------
# ResidualBlock
# --------------------------------------------------------------------------

BLOCK_KEYS = (
    "conv_tanh.conv1d.weight", "conv_tanh.conv1d.bias",
    "conv_sigmoid.conv1d.weight", "conv_sigmoid.conv1d.bias",
    "conv1x1_residual.weight", "conv1x1_residual.bias",
    "conv1x1_skip.weight", "conv1x1_skip.bias",
    "residual_proj.weight", "residual_proj.bias",
)


def block_params(sd, prefix=""):
    """Pick the ten ResidualBlock tensors (modules/block.py:42-48) out of a state dict."""
    return {k: sd[prefix + k] for k in BLOCK_KEYS}


def residual_block(x, p, d, causal, impl="taps", return_saved=False):
    """ResidualBlock.forward (modules/block.py:54-82):
        a = conv_tanh(x); g = conv_sigmoid(x)            (:65-66)
        z = tanh(a) * sigmoid(g)                         (:69, :184-185)
        r = conv1x1_residual(z) + residual_proj(x)       (:73, :77-79)  proj is an nn.Linear, NOT identity
        s = conv1x1_skip(z)                              (:74)
    returns (r, s)."""
    a = dilated_conv(x, p["conv_tanh.conv1d.weight"], p["conv_tanh.conv1d.bias"], d, causal, impl)
    g = dilated_conv(x, p["conv_sigmoid.conv1d.weight"], p["conv_sigmoid.conv1d.bias"], d, causal, impl)
    ta, sg = torch.tanh(a), torch.sigmoid(g)
    z = ta * sg
    Wr, Wk, Wp = p["conv1x1_residual.weight"][:, :, 0], p["conv1x1_skip.weight"][:, :, 0], p["residual_proj.weight"]
    r = (torch.einsum("oc,bcl->bol", Wr, z) + p["conv1x1_residual.bias"].view(1, -1, 1)
         + torch.einsum("oc,bcl->bol", Wp, x) + p["residual_proj.bias"].view(1, -1, 1))
    s = torch.einsum("oc,bcl->bol", Wk, z) + p["conv1x1_skip.bias"].view(1, -1, 1)
    if return_saved:
        return r, s, (ta, sg, z)
    return r, s


def residual_block_backward(x, p, d, causal, dr, ds):
    """Hand-derived backward of `residual_block` (what autograd computes for
    modules/block.py:54-82).  This is the set of formulae the HIP kernels
    implement; tests check it against torch.autograd.

        dz = Wr^T dr + Wk^T ds
        da = dz * sg * (1 - ta^2)        dg = dz * ta * sg * (1 - sg)
        dx[t] = Wp^T dr[t] + sum_j (Wt_j^T da + Ws_j^T dg)[t - off_j]
        dWt_j = sum_{b,t} da[t] x[t+off_j]^T   (same for Ws with dg)
        dWr = sum dr z^T, dWk = sum ds z^T, dWp = sum dr x^T, biases = row sums.
    Returns (dx, grads dict keyed like BLOCK_KEYS)."""
    Wt, Ws = p["conv_tanh.conv1d.weight"], p["conv_sigmoid.conv1d.weight"]
    Wr, Wk, Wp = p["conv1x1_residual.weight"][:, :, 0], p["conv1x1_skip.weight"][:, :, 0], p["residual_proj.weight"]
    k = Wt.shape[2]
    _, _, (ta, sg, z) = residual_block(x, p, d, causal, return_saved=True)
    dz = torch.einsum("oc,bol->bcl", Wr, dr) + torch.einsum("oc,bol->bcl", Wk, ds)
    da = dz * sg * (1 - ta * ta)
    dg = dz * z * (1 - sg)
    dx = torch.einsum("oc,bol->bcl", Wp, dr)
    dWt, dWs = torch.zeros_like(Wt), torch.zeros_like(Ws)
    for j, off in enumerate(tap_offsets(k, d, causal)):
        dx = dx + shifted(torch.einsum("oc,bol->bcl", Wt[:, :, j], da)
                          + torch.einsum("oc,bol->bcl", Ws[:, :, j], dg), -off)
        xs = shifted(x, off)
        dWt[:, :, j] = torch.einsum("bol,bcl->oc", da, xs)
        dWs[:, :, j] = torch.einsum("bol,bcl->oc", dg, xs)
    grads = {
        "conv_tanh.conv1d.weight": dWt, "conv_tanh.conv1d.bias": da.sum((0, 2)),
        "conv_sigmoid.conv1d.weight": dWs, "conv_sigmoid.conv1d.bias": dg.sum((0, 2)),
        "conv1x1_residual.weight": torch.einsum("bol,bcl->oc", dr, z).unsqueeze(2),
        "conv1x1_residual.bias": dr.sum((0, 2)),
        "conv1x1_skip.weight": torch.einsum("bol,bcl->oc", ds, z).unsqueeze(2),
        "conv1x1_skip.bias": ds.sum((0, 2)),
        "residual_proj.weight": torch.einsum("bol,bcl->oc", dr, x),
        "residual_proj.bias": dr.sum((0, 2)),
    }
    return dx, grads


# --------------------------------------------------------------------------
# Stack of blocks + bottlenecks (the loop shared by all three models)
# --------------------------------------------------------------------------


def conv1x1(x, weight, bias):
    """nn.Conv1d(kernel_size=1): y = W[:,:,0] @ x + b."""
    return torch.einsum("oc,bcl->bol", weight[:, :, 0], x) + bias.view(1, -1, 1)


def block_stack(out, skips_sum, sd, layers, causal, impl="taps",
                conv_prefix="convolutions.", bott_prefix="bottlenecks."):
    """for l: out, skip = convolutions[l](out); skips_sum = skips_sum + bottlenecks[l](skip)
    modules/wavenet.py:98-100, modules/raw_ctcnet.py:143-145, modules/classifier.py:110-112."""
    for l, (_ci, _co, _k, d) in enumerate(layers):
        out, skip = residual_block(out, block_params(sd, "%s%d." % (conv_prefix, l)), d, causal, impl)
        skips_sum = skips_sum + conv1x1(skip, sd["%s%d.weight" % (bott_prefix, l)], sd["%s%d.bias" % (bott_prefix, l)])
    return out, skips_sum


def _leaky(x, slopes, key):
    """LeakyReLU(0.01).  `slopes` (optional dict key -> tensor of per-element slopes 1 / 0.01) pins the activation
    pattern: LeakyReLU is not differentiable at 0, and two fp32 evaluations of the same network can put a
    pre-activation that is ~0 on opposite sides of the kink, which changes that element's gradient by 99 %.
    Tests that compare gradients of deep models capture the pattern of the implementation under test and replay it
    here, so that the comparison is about arithmetic, not about which side of a tie was taken."""
    if slopes is not None and key in slopes:
        return x * slopes[key]
    return F.leaky_relu(x, 0.01)


def channel_softmax(x):
    """reshape_in -> F.softmax (implicit dim=1 on a 2-D tensor = channels) -> reshape_out
    (modules/wavenet.py:108-109, modules/conv_ops.py:91-101)."""
    return torch.softmax(x, dim=1)


def wavenet(signal, sd, layers, softmax, impl="taps", slopes=None):
    """WaveNet.forward (modules/wavenet.py:88-111): entry CausalConv1d(d=1) -> causal block stack ->
    LeakyReLU(0.01), 1x1, LeakyReLU(0.01), 1x1 on skips_sum (:67-71,:103) -> optional softmax."""
    out = dilated_conv(signal, sd["entry_conv1d.conv1d.weight"], sd["entry_conv1d.conv1d.bias"], 1, True, impl)
    out_dim = sd["bottlenecks.0.weight"].shape[0]
    skips = torch.zeros(signal.shape[0], out_dim, signal.shape[2], dtype=signal.dtype)
    _, skips = block_stack(out, skips, sd, layers, True, impl)
    y = _leaky(skips, slopes, "output_stack.0")
    y = conv1x1(y, sd["output_stack.1.weight"], sd["output_stack.1.bias"])
    y = _leaky(y, slopes, "output_stack.2")
    y = conv1x1(y, sd["output_stack.3.weight"], sd["output_stack.3.bias"])
    return channel_softmax(y) if softmax else y


def _input_block_and_stack(out, sd, layers, causal, impl):
    """input_block + input_skip_bottleneck, then the stack
    (modules/raw_ctcnet.py:137-145, modules/classifier.py:104-112)."""
    out_dim = sd["input_skip_bottleneck.weight"].shape[0]
    skips = torch.zeros(out.shape[0], out_dim, out.shape[2], dtype=out.dtype)
    in_d = sd["__input_dilation__"]
    out, skip = residual_block(out, block_params(sd, "input_block."), in_d, causal, impl)
    skips = skips + conv1x1(skip, sd["input_skip_bottleneck.weight"], sd["input_skip_bottleneck.bias"])
    _, skips = block_stack(out, skips, sd, layers, causal, impl)
    return skips


def _output_block(skips, sd, slopes=None):
    y = _leaky(skips, slopes, "output_block.0")
    y = conv1x1(y, sd["output_block.1.weight"], sd["output_block.1.bias"])
    y = _leaky(y, slopes, "output_block.2")
    return conv1x1(y, sd["output_block.3.weight"], sd["output_block.3.bias"])


def raw_ctcnet(seq, sd, layers, feature_kwidth, input_dilation=1, positions=False,
               softmax=True, causal=False, impl="taps", slopes=None):
    """RawCTCNet.forward (modules/raw_ctcnet.py:117-153).
    feature_layer = Conv1d(1,F,k,padding=k-1) [length grows to L+k-1, :57-61,:128], LeakyReLU, 1x1, LeakyReLU;
    optional positions: out += Hardtanh(Conv1x1(arange(L'))) (:131-135)."""
    kf = feature_kwidth
    out = F.conv1d(seq, sd["feature_layer.0.weight"], sd["feature_layer.0.bias"], padding=kf - 1)
    out = _leaky(out, slopes, "feature_layer.1")
    out = _leaky(conv1x1(out, sd["feature_layer.2.weight"], sd["feature_layer.2.bias"]), slopes, "feature_layer.3")
    if positions:
        pos = torch.arange(0., out.shape[2], dtype=out.dtype).view(1, 1, -1)
        out = out + F.hardtanh(conv1x1(pos, sd["positions_conv1x1.0.weight"], sd["positions_conv1x1.0.bias"]))
    sd = dict(sd)
    sd["__input_dilation__"] = input_dilation
    y = _output_block(_input_block_and_stack(out, sd, layers, causal, impl), sd, slopes)
    return channel_softmax(y) if softmax else y


def wavenet_classifier(seq, sd, layers, pool_kernel_size=2, input_dilation=1, softmax=True, impl="taps", slopes=None):
    """WaveNetClassifier.forward (modules/classifier.py:91-120): AvgPool1d(pool) -> non-causal stack."""
    out = F.avg_pool1d(seq, pool_kernel_size)
    sd = dict(sd)
    sd["__input_dilation__"] = input_dilation
    y = _output_block(_input_block_and_stack(out, sd, layers, False, impl), sd, slopes)
    return channel_softmax(y) if softmax else y


# --------------------------------------------------------------------------
# helpers for tests / bench
# --------------------------------------------------------------------------


def one_hot_encoding(seq, num_indices):
    """modules/fns.py:6-15: (B, L) long -> (B, num_indices, L) float one-hot."""
    return torch.zeros(seq.size(0), num_indices, seq.size(1)).scatter_(1, seq.unsqueeze(1), 1.)


def capture_leaky_slopes(model):
    """Register forward-pre-hooks on every nn.LeakyReLU of `model`; returns (slopes dict, remove()).  After a forward
    pass slopes["<module name>"] holds the per-element slope (1 where the input was > 0, else 0.01) on the CPU.

    In the half-precision modes the HIP package runs the output block (and RawCTCNet's feature layer) INSIDE its stack
    function (series layout): the LeakyReLU modules are then never called.  The un-fused form is the same forward computation
    (tests/test_gpu_head.py: bitwise for the output block), so the pattern is taken from one extra no_grad forward in that form
    (WN_SERIES_HEAD=0, WN_SERIES_FRONT=0), run by a hook on the model itself just before the real, fused forward."""
    import os
    import torch.nn as nn
    slopes, handles = {}, []
    for name, mod in model.named_modules():
        if isinstance(mod, nn.LeakyReLU):
            def hook(_m, inp, name=name, ns=mod.negative_slope):
                xin = inp[0].detach()
                slopes[name] = torch.where(xin > 0, torch.ones_like(xin), torch.full_like(xin, ns)).cpu()
            handles.append(mod.register_forward_pre_hook(hook))
    state = {"busy": False}

    def unfused_first(m, inp):
        if state["busy"]:
            return
        state["busy"] = True
        old = {k: os.environ.get(k) for k in ("WN_SERIES_HEAD", "WN_SERIES_FRONT")}
        for k in old:
            os.environ[k] = "0"
        try:
            with torch.no_grad():
                m(*[t.detach() if isinstance(t, torch.Tensor) else t for t in inp])
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
            state["busy"] = False
    if any(getattr(m, "stack_state", None) is not None for m in model.modules()):
        handles.append(model.register_forward_pre_hook(unfused_first))
    return slopes, lambda: [h.remove() for h in handles]


def rel_err(a, b):
    """max |a-b| / max |b|  -- the parity metric used throughout tests (tolerance 1e-4, north_star)."""
    a, b = a.detach(), b.detach()
    denom = float(b.abs().max())
    return float((a - b).abs().max()) / (denom if denom > 0 else 1.0)


def random_wavenet_state(in_dim, entry_kwidth, layers, out_dim, seed=0, dtype=torch.float32):
    """A state dict with the reference's key names/shapes and kaiming-uniform-like values
    (modules/wavenet.py:74-85); used to build synthetic models on the GPU box where the
    reference is absent.  Values are NOT bit-identical to the reference initialiser."""
    g = torch.Generator().manual_seed(seed)

    def ku(*shape):
        fan_in = shape[1] * (shape[2] if len(shape) > 2 else 1)
        bound = math.sqrt(6.0 / fan_in)
        return ((torch.rand(*shape, generator=g, dtype=torch.float64) * 2 - 1) * bound).to(dtype)

    sd = {"entry_conv1d.conv1d.weight": ku(layers[0][0], in_dim, entry_kwidth),
          "entry_conv1d.conv1d.bias": torch.zeros(layers[0][0], dtype=dtype)}
    for l, (ci, co, k, _d) in enumerate(layers):
        pre = "convolutions.%d." % l
        sd[pre + "conv_tanh.conv1d.weight"] = ku(co, ci, k)
        sd[pre + "conv_tanh.conv1d.bias"] = 0.01 * ku(co, 1)[:, 0]
        sd[pre + "conv_sigmoid.conv1d.weight"] = ku(co, ci, k)
        sd[pre + "conv_sigmoid.conv1d.bias"] = 0.01 * ku(co, 1)[:, 0]
        sd[pre + "conv1x1_residual.weight"] = ku(co, co, 1)
        sd[pre + "conv1x1_residual.bias"] = 0.01 * ku(co, 1)[:, 0]
        sd[pre + "conv1x1_skip.weight"] = ku(co, co, 1)
        sd[pre + "conv1x1_skip.bias"] = 0.01 * ku(co, 1)[:, 0]
        sd[pre + "residual_proj.weight"] = ku(co, ci)
        sd[pre + "residual_proj.bias"] = 0.01 * ku(co, 1)[:, 0]
    for l, (_ci, co, _k, _d) in enumerate(layers):
        sd["bottlenecks.%d.weight" % l] = ku(out_dim, co, 1)
        sd["bottlenecks.%d.bias" % l] = torch.zeros(out_dim, dtype=dtype)
    sd["output_stack.1.weight"] = ku(out_dim, out_dim, 1)
    sd["output_stack.1.bias"] = torch.zeros(out_dim, dtype=dtype)
    sd["output_stack.3.weight"] = ku(out_dim, out_dim, 1)
    sd["output_stack.3.bias"] = torch.zeros(out_dim, dtype=dtype)
    return sd


def random_rawctcnet_state(num_features, feature_kwidth, num_labels, layers, out_dim, input_kernel_size=2, seed=0,
                           dtype=torch.float32):
    """State dict with the key names/shapes of RawCTCNet (modules/raw_ctcnet.py:57-93, positions=False) and values
    following its init rules (:95-115: kaiming-uniform weights, ~0 biases, identity + 1e-4 noise bottlenecks).
    Values are NOT bit-identical to the reference initialiser; used by bench.py's cpu_baseline leg."""
    g = torch.Generator().manual_seed(seed)

    def ku(*shape):
        fan_in = shape[1] * (shape[2] if len(shape) > 2 else 1)
        bound = math.sqrt(6.0 / fan_in)
        return ((torch.rand(*shape, generator=g, dtype=torch.float64) * 2 - 1) * bound).to(dtype)

    def nz(n):
        return (1e-4 * torch.randn(n, generator=g, dtype=torch.float64)).to(dtype)

    def eye(co, ci):
        return (torch.eye(co, ci, dtype=torch.float64) + 1e-4 * torch.randn(co, ci, generator=g, dtype=torch.float64)
                ).to(dtype).unsqueeze(2)

    F_, c0 = num_features, layers[0][0]
    sd = {"feature_layer.0.weight": ku(F_, 1, feature_kwidth), "feature_layer.0.bias": nz(F_),
          "feature_layer.2.weight": ku(F_, F_, 1), "feature_layer.2.bias": nz(F_)}

    def block(pre, ci, co, k):
        sd[pre + "conv_tanh.conv1d.weight"] = ku(co, ci, k)
        sd[pre + "conv_tanh.conv1d.bias"] = nz(co)
        sd[pre + "conv_sigmoid.conv1d.weight"] = ku(co, ci, k)
        sd[pre + "conv_sigmoid.conv1d.bias"] = nz(co)
        sd[pre + "conv1x1_residual.weight"] = ku(co, co, 1)
        sd[pre + "conv1x1_residual.bias"] = nz(co)
        sd[pre + "conv1x1_skip.weight"] = ku(co, co, 1)
        sd[pre + "conv1x1_skip.bias"] = nz(co)
        sd[pre + "residual_proj.weight"] = ku(co, ci)
        sd[pre + "residual_proj.bias"] = nz(co)

    block("input_block.", F_, c0, input_kernel_size)
    sd["input_skip_bottleneck.weight"] = ku(out_dim, c0, 1)
    sd["input_skip_bottleneck.bias"] = nz(out_dim)
    for l, (ci, co, k, _d) in enumerate(layers):
        block("convolutions.%d." % l, ci, co, k)
    for l, (_ci, co, _k, _d) in enumerate(layers):
        sd["bottlenecks.%d.weight" % l] = eye(out_dim, co)
        sd["bottlenecks.%d.bias" % l] = nz(out_dim)
    sd["output_block.1.weight"] = ku(out_dim, out_dim, 1)
    sd["output_block.1.bias"] = nz(out_dim)
    sd["output_block.3.weight"] = ku(num_labels, out_dim, 1)
    sd["output_block.3.bias"] = nz(num_labels)
    return sd
