"""
Host side of the hot path: torch.autograd.Function wrappers that drive libwavenet_amd.so through its
C ABI.  torch supplies device memory, the current HIP stream and autograd bookkeeping -- all arithmetic
(dilated convs, gate, 1x1 products, every gradient) runs in the HIP kernels.

There is no CPU path: CPU tensors raise.
"""
import ctypes
import os

import torch
from torch.autograd.function import once_differentiable

from . import _flags, _lib
from .series import Lease, SeriesLayout, fresh_series, load_series, window

PARAMS_PER_BLOCK = 10  # order = _lib.BlockParams fields


class BlockSpec(object):
    """Static description of one residual block (modules/block.py:22-51 in the reference) as the C ABI sees it."""
    __slots__ = ("ci", "co", "ms", "k", "d", "causal")

    def __init__(self, ci, co, ms, k, d, causal):
        self.ci, self.co, self.ms, self.k, self.d, self.causal = int(ci), int(co), int(ms), int(k), int(d), bool(causal)

    def offsets(self):
        return _lib.tap_offsets(self.k, self.d, self.causal)

    def reach(self):
        return max(abs(o) for o in self.offsets())


def _on_device_of_first_tensor(fn):
    """Run an autograd.Function forward/backward with the CUDA/HIP device of its first tensor argument current, so
    that torch.cuda.current_stream() and the kernel launches of the C ABI target the device that owns the buffers."""
    import functools

    @functools.wraps(fn)
    def wrapper(ctx, *args):
        dev = next((a.device for a in args if isinstance(a, torch.Tensor) and a.is_cuda), None)
        if dev is None:
            return fn(ctx, *args)
        with torch.cuda.device(dev):
            return fn(ctx, *args)
    return wrapper


def _require_device(t, what):
    if not t.is_cuda:
        raise RuntimeError("wavenet_speech_amd: %s is a CPU tensor; the HIP path needs ROCm device tensors "
                           "(there is no CPU fallback)" % what)
    if t.dtype != torch.float32:
        raise RuntimeError("wavenet_speech_amd: %s must be float32, got %s" % (what, t.dtype))


def _own(view):
    """Fresh contiguous copy of a window of a pooled buffer.  `.contiguous()` is NOT enough: when halo == 0, L is a
    multiple of 128 and C a multiple of 8 the padded layout is already dense and it would return the alias."""
    return view.clone(memory_format=torch.contiguous_format)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(x):
    """device pointer of a tensor / Lease / None"""
    if x is None:
        return None
    if isinstance(x, int):
        return ctypes.c_void_p(x)
    if isinstance(x, Lease):
        return ctypes.c_void_p(x.ptr)
    return ctypes.c_void_p(x.data_ptr())


def _shape(spec, batch, layout):
    return _lib.BlockShape(batch, layout.length, spec.ci, spec.co, spec.ms, spec.k, spec.d, int(spec.causal),
                           layout.ld, layout.halo)


def _params_struct(tensors):
    return _lib.BlockParams(*[_p(t) for t in tensors])


def _prep_params(tensors, spec):
    """contiguous fp32 device copies (no-ops for ordinary nn.Parameters) + shape checks"""
    want = [(spec.co, spec.ci, spec.k), (spec.co,), (spec.co, spec.ci, spec.k), (spec.co,),
            (spec.co, spec.co), (spec.co,), (spec.ms, spec.co), (spec.ms,), (spec.co, spec.ci), (spec.co,)]
    out = []
    for t, w in zip(tensors, want):
        _require_device(t, "parameter")
        if t.dim() == 3 and len(w) == 2:  # 1x1 Conv1d weights arrive as [Co][Ci][1]
            t = t[:, :, 0]
        if tuple(t.shape) != w:
            raise RuntimeError("wavenet_speech_amd: parameter shape %s, expected %s" % (tuple(t.shape), w))
        out.append(t.detach().contiguous())
    return out


def _pack_block(lib, shape, params, device):
    nbytes = lib.wn_block_packed_bytes(ctypes.byref(shape))
    if nbytes == 0:
        _lib.check(-1, "wn_block_packed_bytes")
    packed = torch.empty(nbytes, dtype=torch.uint8, device=device)
    ps = _params_struct(params)
    _lib.check(lib.wn_block_pack(ctypes.byref(shape), ctypes.byref(ps), _p(packed), _stream()), "wn_block_pack")
    return packed


def _block_backward(lib, shape, spec, packed, x, sg, z, dr, dskip, want_dx, layout, batch, device):
    """bwd-data + bwd-weights of one block; returns (dx lease or None, [10 gradient tensors])"""
    da, dg = Lease(batch, spec.co, layout, device), Lease(batch, spec.co, layout, device)
    dx = Lease(batch, spec.ci, layout, device) if want_dx else None
    _lib.check(lib.wn_block_backward_data(ctypes.byref(shape), _p(packed), _p(dr), _p(dskip), _p(z), _p(sg),
                                          _p(da), _p(dg), _p(dx), _stream()), "wn_block_backward_data")
    k = spec.k
    shapes = [(spec.co, spec.ci, k), (spec.co,), (spec.co, spec.ci, k), (spec.co,), (spec.co, spec.co), (spec.co,),
              (spec.ms, spec.co), (spec.ms,), (spec.co, spec.ci), (spec.co,)]
    # the last block of a stack has no consumer of its residual output (dr is None): conv1x1_residual and
    # residual_proj then get NO gradient (None), exactly as autograd leaves them in the reference -- a zero tensor
    # would make weight-decay optimisers decay parameters the reference never touches
    unused = (4, 5, 8, 9) if dr is None else ()
    grads = [None if i in unused else torch.empty(s, dtype=torch.float32, device=device) for i, s in enumerate(shapes)]
    ws_bytes = lib.wn_block_wgrad_workspace_bytes(ctypes.byref(shape))
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=device)
    gs = _params_struct(grads)
    _lib.check(lib.wn_block_backward_weights(ctypes.byref(shape), _p(x), _p(z), _p(da), _p(dg), _p(dr), _p(dskip),
                                             ctypes.byref(gs), _p(ws), ws_bytes, _stream()),
               "wn_block_backward_weights")
    return dx, grads


def _stack_skip_sum(lib, specs, zs, skip_w, skip_b, S, batch, layout, device):
    """S = sum_l W_skip_l z_l + sum_l b_skip_l, in groups of <= MAX_STACK_GROUP blocks (one long-K GEMM each)."""
    bias_total = torch.stack(skip_b).sum(0).contiguous()
    G = _lib.MAX_STACK_GROUP
    for g0 in range(0, len(specs), G):
        idx = range(g0, min(g0 + G, len(specs)))
        n = len(idx)
        shape = _lib.SkipSumShape(batch, layout.length, specs[0].ms, n, layout.ld, layout.halo)
        for i, l in enumerate(idx):
            shape.channels[i] = specs[l].co
        nbytes = lib.wn_skipsum_packed_bytes(ctypes.byref(shape))
        if nbytes == 0:
            _lib.check(-1, "wn_skipsum_packed_bytes")
        packed = torch.empty(nbytes, dtype=torch.uint8, device=device)
        wptrs = (ctypes.c_void_p * n)(*[skip_w[l].data_ptr() for l in idx])
        zptrs = (ctypes.c_void_p * n)(*[zs[l].ptr for l in idx])
        _lib.check(lib.wn_skipsum_pack(ctypes.byref(shape), wptrs, _p(bias_total) if g0 == 0 else None, _p(packed),
                                       _stream()), "wn_skipsum_pack")
        _lib.check(lib.wn_skipsum_forward(ctypes.byref(shape), _p(packed), zptrs, _p(S), 0 if g0 == 0 else 1, _stream()),
                   "wn_skipsum_forward")


class _ResidualStackFn(torch.autograd.Function):
    """The per-layer loop of WaveNet / RawCTCNet / WaveNetClassifier
    (modules/wavenet.py:98-100, raw_ctcnet.py:138-145, classifier.py:105-112):
        for l: out, skip = block_l(out); skips_sum = skips_sum + bottleneck_l(skip)
    with the bottleneck folded into the skip projection by the caller.  Returns skips_sum [B, Ms, L]."""

    @staticmethod
    @_on_device_of_first_tensor
    def forward(ctx, x, specs, grad_enabled, pack_cache, pool, *flat):
        """pool > 1: AvgPool1d(pool) of x (reference modules/classifier.py:53,102) fused into the load of the stack's input series"""
        lib = _lib.load()
        _require_device(x, "input")
        n = len(specs)
        assert len(flat) == n * PARAMS_PER_BLOCK
        B, C0, L = x.shape
        ctx.pool, ctx.in_length = int(pool), L
        if pool > 1:
            L = L // pool
            if L < 1:
                raise RuntimeError("wavenet_speech_amd: sequence shorter than the pooling window")
        if C0 != specs[0].ci:
            raise RuntimeError("wavenet_speech_amd: input has %d channels, first block expects %d" % (C0, specs[0].ci))
        for l in range(1, n):
            if specs[l].ci != specs[l - 1].co:   # the reference raises a conv1d shape error here
                raise RuntimeError("wavenet_speech_amd: block %d expects %d input channels but block %d produces %d"
                                   % (l, specs[l].ci, l - 1, specs[l - 1].co))
        dev = x.device
        layout = SeriesLayout(L, max(s.reach() for s in specs))
        # needs_input_grad reflects requires_grad of the arguments whatever the grad mode, and grad mode is always off
        # inside Function.forward: the caller captures torch.is_grad_enabled() and hands it in
        training = bool(grad_enabled) and any(ctx.needs_input_grad)
        ctx.training = training
        cur = Lease(B, C0, layout, dev)
        if pool > 1:
            _lib.check(lib.wn_series_load_pooled(_p(x.detach().contiguous()), _p(cur), B, C0, ctx.in_length, int(pool), layout.ld,
                                                 layout.halo, _stream()), "wn_series_load_pooled")
        else:
            load_series(cur.t, x.detach(), layout)
        ms = specs[0].ms
        S = fresh_series(B, ms, layout, dev)
        saved, skip_w, skip_b = [], [], []
        zbuf = None
        for l, spec in enumerate(specs):
            if spec.ms != ms:
                raise RuntimeError("wavenet_speech_amd: all blocks of a stack must share out_dim")
            shape = _shape(spec, B, layout)
            params = _prep_params(flat[l * PARAMS_PER_BLOCK:(l + 1) * PARAMS_PER_BLOCK], spec)
            if pack_cache is not None and pack_cache.frozen and not grad_enabled:
                packed = pack_cache.get(l, layout, B)
                if packed is None:
                    packed = pack_cache.put(l, layout, B, _pack_block(lib, shape, params, dev))
            else:
                packed = _pack_block(lib, shape, params, dev)
            r = Lease(B, spec.co, layout, dev) if l + 1 < n else None  # the last residual output is never used
            if training:
                sg, z = (Lease(B, spec.co, layout, dev) for _ in range(2))   # kept for backward; tanh = z / sg
            else:
                sg = None
                if zbuf is None or zbuf.channels != spec.co:
                    zbuf = Lease(B, spec.co, layout, dev)
                z = zbuf
            # training keeps every block's z, so skips_sum is formed afterwards by one long-K product over all
            # blocks (wn_skipsum_forward); inference accumulates per block and keeps a single z scratch buffer
            _lib.check(lib.wn_block_forward(ctypes.byref(shape), _p(packed), _p(cur), _p(r), None if training else _p(S),
                                            1, _p(sg), _p(z), _stream()), "wn_block_forward")
            if training:
                saved.append((cur, sg, z, packed, shape))
                skip_w.append(params[6])
                skip_b.append(params[7])
            cur = r
        if training:
            _stack_skip_sum(lib, specs, [sv[2] for sv in saved], skip_w, skip_b, S, B, layout, dev)
        ctx.specs, ctx.saved, ctx.layout, ctx.batch = specs, saved, layout, B
        ctx.param_shapes = [tuple(t.shape) for t in flat]
        # hand autograd a tensor of its own: views of internal buffers must never escape a custom Function
        return window(S, ms, layout).clone(memory_format=torch.contiguous_format)

    @staticmethod
    @once_differentiable
    @_on_device_of_first_tensor
    def backward(ctx, d_skips):
        lib = _lib.load()
        specs, layout, B = ctx.specs, ctx.layout, ctx.batch
        dev = d_skips.device
        dS = Lease(B, specs[0].ms, layout, dev)
        load_series(dS.t, d_skips, layout)
        dr = None
        grads_flat = [None] * (len(specs) * PARAMS_PER_BLOCK)
        for l in range(len(specs) - 1, -1, -1):
            spec = specs[l]
            x, sg, z, packed, shape = ctx.saved[l]
            want_dx = l > 0 or ctx.needs_input_grad[0]
            dx, grads = _block_backward(lib, shape, spec, packed, x, sg, z, dr, dS, want_dx, layout, B, dev)
            grads_flat[l * PARAMS_PER_BLOCK:(l + 1) * PARAMS_PER_BLOCK] = grads
            dr = dx
            ctx.saved[l] = None  # release this block's activations to the pool
        dx0 = window(dr.t, specs[0].ci, layout).clone(memory_format=torch.contiguous_format) if ctx.needs_input_grad[0] else None
        if dx0 is not None and ctx.pool > 1:
            dx0 = _unpool(lib, dx0, ctx.in_length, ctx.pool)
        # 1x1 Conv1d weights come in as [Co][Ci][1]; hand each gradient back in its parameter's own shape
        grads_flat = [None if g is None else g.view(shp) for g, shp in zip(grads_flat, ctx.param_shapes)]
        return (dx0, None, None, None, None) + tuple(grads_flat)


def _unpool(lib, dpooled, length, pool):
    """backward of the pooled load: each pooled gradient / pool, broadcast to its pool columns (0 for a dropped tail)"""
    B, C, _ = dpooled.shape
    dx = torch.empty(B, C, length, dtype=torch.float32, device=dpooled.device)
    _lib.check(lib.wn_pool_backward(_p(dpooled.contiguous()), _p(dx), B, C, length, int(pool), _stream()), "wn_pool_backward")
    return dx


class PackCache(object):
    """Per-stack packing state.

    * `tables` (always on): device-resident pack-job tables of the half-precision modes (functional_half.StackPackTable) --
      they hold pointers and shapes, never weight VALUES, so every forward still packs the current weights (one launch).
    * `packed` (only when `frozen`, see modules.block.freeze_for_inference): the packed weights themselves, kept across
      no_grad forwards.  The owner calls `validate(params)` before each use; an in-place update through the autograd-visible
      API (optimizer step, load_state_dict, p.mul_()) bumps a parameter's `_version` and empties the cache, so does a
      parameter moving to other storage.  Updates through `p.data` (p.data.mul_(2)) are invisible to torch's version counter:
      that is why keeping packed weights is opt-in."""

    def __init__(self):
        self.key = None
        self.packed = {}
        self.hits = 0
        self.tables = {}           # device-resident pack-job tables (functional_half.StackPackTable), by their key
        self.frozen = False        # keep packed weights / folded bottlenecks across no_grad forwards (opt-in)

    def validate(self, params, extra=()):
        key = tuple((id(p), p._version, p.data_ptr(), p.device) for p in params) + tuple(extra)
        if key != self.key:
            self.key = key
            self.packed = {}
            self.__dict__["_shapes"] = []

    MAX_SHAPES = 4        # distinct (series layout, batch) shapes whose packed weights are kept (variable-length inference would
                          # otherwise keep one packed copy of every block per utterance length: ADVICE r02)

    def get(self, l, layout, batch):
        shape = (layout.key(), batch)
        t = self.packed.get((l,) + shape)
        if t is not None:
            self.hits += 1
            self._touch(shape)
        return t

    def put(self, l, layout, batch, t):
        shape = (layout.key(), batch)
        self._touch(shape)
        self.packed[(l,) + shape] = t
        return t

    def _touch(self, shape):
        order = self.__dict__.setdefault("_shapes", [])
        if shape in order:
            order.remove(shape)
        order.append(shape)
        while len(order) > self.MAX_SHAPES:
            old = order.pop(0)                       # least recently used shape: drop its packed weights
            for k in [k for k in self.packed if k[1:] == old]:
                del self.packed[k]


def residual_stack(x, specs, flat_params, precision="f32", pack_cache=None, head=None, front=None, pool=1):
    """skips_sum of a stack of residual blocks.  flat_params: 10 tensors per block in C-ABI order
    (w_tanh, b_tanh, w_sigmoid, b_sigmoid, w_res [Co,Co,1], b_res, w_skip [Ms,Co], b_skip, w_proj, b_proj).
    precision: "f32" (exact fp32 MFMA, default) or one of the half-precision MFMA modes of functional_half."""
    if precision != "f32":
        from . import functional_half
        return functional_half.residual_stack(x, specs, flat_params, precision, pack_cache, head, front, pool)
    if head is not None or front is not None:
        raise ValueError("the fused output block / feature layer exist in the half-precision modes only")
    return _ResidualStackFn.apply(x, tuple(specs), torch.is_grad_enabled(), pack_cache, int(pool), *flat_params)


class _ResidualBlockFn(torch.autograd.Function):
    """Stand-alone ResidualBlock.forward (modules/block.py:54-82): returns (residual_out, skip_out)."""

    @staticmethod
    @_on_device_of_first_tensor
    def forward(ctx, x, spec, grad_enabled, *params):
        lib = _lib.load()
        _require_device(x, "input")
        B, C0, L = x.shape
        if C0 != spec.ci:
            raise RuntimeError("wavenet_speech_amd: input has %d channels, block expects %d" % (C0, spec.ci))
        dev = x.device
        layout = SeriesLayout(L, spec.reach())
        shape = _shape(spec, B, layout)
        prm = _prep_params(params, spec)
        packed = _pack_block(lib, shape, prm, dev)
        xin = Lease(B, C0, layout, dev)
        load_series(xin.t, x.detach(), layout)
        r = Lease(B, spec.co, layout, dev)
        s = Lease(B, spec.ms, layout, dev)
        training = bool(grad_enabled) and any(ctx.needs_input_grad)
        sg = Lease(B, spec.co, layout, dev) if training else None
        z = Lease(B, spec.co, layout, dev)
        _lib.check(lib.wn_block_forward(ctypes.byref(shape), _p(packed), _p(xin), _p(r), _p(s), 0,
                                        _p(sg), _p(z), _stream()), "wn_block_forward")
        if training:
            ctx.saved = (xin, sg, z, packed, shape)
        ctx.spec, ctx.layout, ctx.batch = spec, layout, B
        ctx.param_shapes = [tuple(t.shape) for t in params]
        return _own(r.view()), _own(s.view())

    @staticmethod
    @once_differentiable
    @_on_device_of_first_tensor
    def backward(ctx, d_r, d_s):
        lib = _lib.load()
        spec, layout, B = ctx.spec, ctx.layout, ctx.batch
        dev = d_r.device
        x, sg, z, packed, shape = ctx.saved
        dr, ds = Lease(B, spec.co, layout, dev), Lease(B, spec.ms, layout, dev)
        load_series(dr.t, d_r, layout)
        load_series(ds.t, d_s, layout)
        dx, grads = _block_backward(lib, shape, spec, packed, x, sg, z, dr, ds, ctx.needs_input_grad[0], layout, B, dev)
        grads = [g.view(shp) for g, shp in zip(grads, ctx.param_shapes)]
        dx0 = _own(dx.view()) if dx is not None else None
        ctx.saved = None
        return (dx0, None, None) + tuple(grads)


def residual_block(x, spec, params):
    """(residual_out, skip_out) of one block; params in C-ABI order with w_res / w_skip as [Co,Co,1] Conv1d weights."""
    return _ResidualBlockFn.apply(x, spec, torch.is_grad_enabled(), *params)


class _DilatedConvFn(torch.autograd.Function):
    """CausalConv1d / NonCausalConv1d forward (modules/conv_ops.py:39-44, 73-79); k=1 gives a 1x1 Conv1d."""

    @staticmethod
    @_on_device_of_first_tensor
    def forward(ctx, x, weight, bias, dilation, causal, grad_enabled):
        lib = _lib.load()
        _require_device(x, "input")
        _require_device(weight, "weight")
        B, Ci, L = x.shape
        Co, Ci_w, k = weight.shape
        if Ci_w != Ci:
            raise RuntimeError("wavenet_speech_amd: input has %d channels, conv expects %d" % (Ci, Ci_w))
        dev = x.device
        reach = max(abs(o) for o in _lib.tap_offsets(k, dilation, causal))
        layout = SeriesLayout(L, reach)
        shape = _lib.ConvShape(B, L, Ci, Co, k, int(dilation), int(bool(causal)), layout.ld, layout.halo)
        nbytes = lib.wn_conv_packed_bytes(ctypes.byref(shape))
        if nbytes == 0:
            _lib.check(-1 if k <= _lib.MAX_TAPS else -2, "wn_conv_packed_bytes")
        packed = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        w = weight.detach().contiguous()
        b = bias.detach().contiguous() if bias is not None else None
        _lib.check(lib.wn_conv_pack(ctypes.byref(shape), _p(w), _p(b), _p(packed), _stream()), "wn_conv_pack")
        xin = Lease(B, Ci, layout, dev)
        load_series(xin.t, x.detach(), layout)
        y = Lease(B, Co, layout, dev)
        _lib.check(lib.wn_conv_forward(ctypes.byref(shape), _p(packed), _p(xin), _p(y), _stream()), "wn_conv_forward")
        if grad_enabled and any(ctx.needs_input_grad):
            ctx.saved = (xin, packed, shape)
        ctx.layout, ctx.dims, ctx.has_bias = layout, (B, Ci, Co, k), bias is not None
        return _own(y.view())

    @staticmethod
    @once_differentiable
    @_on_device_of_first_tensor
    def backward(ctx, d_y):
        lib = _lib.load()
        xin, packed, shape = ctx.saved
        layout = ctx.layout
        B, Ci, Co, k = ctx.dims
        dev = d_y.device
        dy = Lease(B, Co, layout, dev)
        load_series(dy.t, d_y, layout)
        dx0 = None
        if ctx.needs_input_grad[0]:
            dx = Lease(B, Ci, layout, dev)
            _lib.check(lib.wn_conv_backward_data(ctypes.byref(shape), _p(packed), _p(dy), _p(dx), _stream()),
                       "wn_conv_backward_data")
            dx0 = _own(dx.view())
        dw = torch.empty(Co, Ci, k, dtype=torch.float32, device=dev)
        db = torch.empty(Co, dtype=torch.float32, device=dev) if ctx.has_bias else None
        ws_bytes = lib.wn_conv_wgrad_workspace_bytes(ctypes.byref(shape))
        ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=dev)
        _lib.check(lib.wn_conv_backward_weights(ctypes.byref(shape), _p(xin), _p(dy), _p(dw), _p(db), _p(ws), ws_bytes,
                                                _stream()), "wn_conv_backward_weights")
        ctx.saved = None
        return dx0, dw, db, None, None, None


def dilated_conv(x, weight, bias, dilation=1, causal=True, precision="f32"):
    if precision != "f32":
        from . import functional_half
        return functional_half.conv(x, weight, bias, dilation, causal, precision)
    return _DilatedConvFn.apply(x, weight, bias, int(dilation), bool(causal), torch.is_grad_enabled())


class _SequenceNLLFn(torch.autograd.Function):
    """sum_t mean_b CE(logits[:, :, t], target[:, t])  (Loss.py:38-43) as one fused HIP pass each way."""

    @staticmethod
    @_on_device_of_first_tensor
    def forward(ctx, logits, target, grad_enabled=True):
        lib = _lib.load()
        _require_device(logits, "logits")
        if target.dtype != torch.int64 or not target.is_cuda:
            raise RuntimeError("wavenet_speech_amd: target must be an int64 device tensor")
        B, C, L = logits.shape
        x = logits.detach().contiguous()
        tg = target.contiguous()
        lse = torch.empty(B, L, dtype=torch.float32, device=x.device)
        partial = torch.empty(lib.wn_nll_partials(B, L), dtype=torch.float32, device=x.device)
        bad = torch.zeros(1, dtype=torch.int32, device=x.device)
        _lib.check(lib.wn_nll_forward(_p(x), _p(tg), _p(lse), _p(partial), _p(bad), B, C, L, _stream()), "wn_nll_forward")
        # the kernel never indexes the logits with an out-of-range label (it counts it and poisons the loss with NaN); the count
        # becomes an exception without stalling the stream in training calls (see _flags.py; WN_NLL_CHECK=0 skips it)
        if os.environ.get("WN_NLL_CHECK", "1") != "0":
            _flags.WATCH.poll()
            _flags.WATCH.note(bad, lambda n, C=C: "wavenet_speech_amd: sequence_nll got %d target(s) outside [0, %d)" % (n, C),
                              at_once=not (grad_enabled and ctx.needs_input_grad[0]))
        ctx.save_for_backward(x, tg, lse)
        return partial.sum() / B

    @staticmethod
    @once_differentiable
    @_on_device_of_first_tensor
    def backward(ctx, g):
        lib = _lib.load()
        x, tg, lse = ctx.saved_tensors
        B, C, L = x.shape
        gscale = (g / B).to(torch.float32).reshape(1).contiguous()
        dx = torch.empty_like(x)
        _lib.check(lib.wn_nll_backward(_p(x), _p(tg), _p(lse), _p(gscale), _p(dx), B, C, L, _stream()), "wn_nll_backward")
        return dx, None, None


def sequence_nll(logits, target):
    return _SequenceNLLFn.apply(logits, target, torch.is_grad_enabled())


class _EmbedConvFn(torch.autograd.Function):
    """CausalConv1d(d=1) applied to one_hot(levels) (modules/wavenet.py:54,93 + modules/fns.py:6-15) as a gather of weight
    columns: the [B, classes, L] one-hot is never built."""

    @staticmethod
    @_on_device_of_first_tensor
    def forward(ctx, weight, bias, levels, grad_enabled=True):
        lib = _lib.load()
        _require_device(weight, "weight")
        if levels.dtype != torch.int64 or not levels.is_cuda or levels.dim() != 2:
            raise RuntimeError("wavenet_speech_amd: levels must be an int64 device tensor [B, L]")
        Co, classes, k = weight.shape
        B, L = levels.shape
        q = levels.contiguous()
        w = weight.detach().contiguous()
        b = bias.detach().contiguous() if bias is not None else None
        y = torch.empty(B, Co, L, dtype=torch.float32, device=weight.device)
        bad = torch.zeros(1, dtype=torch.int32, device=weight.device)
        _lib.check(lib.wn_embed_forward(_p(q), _p(w), _p(b), _p(y), B, L, classes, Co, k, _p(bad), _stream()), "wn_embed_forward")
        if os.environ.get("WN_NLL_CHECK", "1") != "0":
            _flags.WATCH.poll()
            _flags.WATCH.note(bad, lambda n, classes=classes: "wavenet_speech_amd: %d thread(s) saw a level outside [0, %d)" % (n, classes),
                              at_once=not (grad_enabled and any(ctx.needs_input_grad)))
        ctx.save_for_backward(q)
        ctx.dims, ctx.has_bias = (B, L, classes, Co, k), bias is not None
        return y

    @staticmethod
    @once_differentiable
    @_on_device_of_first_tensor
    def backward(ctx, d_y):
        lib = _lib.load()
        (q,) = ctx.saved_tensors
        B, L, classes, Co, k = ctx.dims
        dev = d_y.device
        dy = d_y.contiguous()
        dw = torch.empty(Co, classes, k, dtype=torch.float32, device=dev)
        db = torch.empty(Co, dtype=torch.float32, device=dev) if ctx.has_bias else None
        # the exact-fp32 weight-gradient GEMM (wgrad_kernel, fixed summation order) against a one-hot that exists only inside this
        # call: backward DOES materialise a one-hot (279 MB at 16 x 256 x 16000, from the lease pool); the forward pass and the
        # saved state never hold one.  (A gather-form kernel without it was 3.5x slower and was removed: csrc/wn_embed.hip.)
        layout = SeriesLayout(L, k - 1)
        shape = _lib.ConvShape(B, L, classes, Co, k, 1, 1, layout.ld, layout.halo)
        xin = Lease(B, classes, layout, dev)
        # a level outside [0, classes) was counted by the forward kernel (and raises, one call late in training calls): here it must
        # not become an out-of-bounds scatter (ROCm builds compile the index assert out) -- it contributes nothing, like in the kernels
        ok = (q >= 0) & (q < classes)
        window(xin.t, classes, layout).zero_().scatter_(1, q.clamp(0, classes - 1).unsqueeze(1), ok.unsqueeze(1).to(torch.float32))
        dyl = Lease(B, Co, layout, dev)
        load_series(dyl.t, dy, layout)
        ws_bytes = lib.wn_conv_wgrad_workspace_bytes(ctypes.byref(shape))
        ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=dev)
        _lib.check(lib.wn_conv_backward_weights(ctypes.byref(shape), _p(xin), _p(dyl), _p(dw), _p(db), _p(ws), ws_bytes,
                                                _stream()), "wn_conv_backward_weights")
        return dw, db, None, None


def embed_conv(levels, weight, bias):
    """entry_conv1d(one_hot(levels)) without the one-hot: levels [B, L] int64 -> [B, Co, L]"""
    return _EmbedConvFn.apply(weight, bias, levels, torch.is_grad_enabled())


# ------------------------------------------------------------------------------------------------------------------
# measurement hooks
# ------------------------------------------------------------------------------------------------------------------
def profile_enable(on=True):
    lib = _lib.load()
    _lib.check(lib.wn_prof_enable(1 if on else 0), "wn_prof_enable")


def profile_reset():
    _lib.check(_lib.load().wn_prof_reset(), "wn_prof_reset")


def profile_read():
    return _lib.profile_read()
