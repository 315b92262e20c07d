"""
Host side of the half-precision-MFMA modes of the residual stack ("f16x3", "f16", "bf16"): the same
torch.autograd.Function shape as functional._ResidualStackFn, driving the wn_h* entry points of the C ABI.

    f16x3  operands split into two fp16 planes, three MFMAs per product, fp32 accumulate: 22-bit operands, within 1e-4 of the fp32 path on conditioned models
           (same 1e-4 parity bar as the fp32 path) at 3/16 of the fp32 MFMA cost
    f16 / bf16   plain half storage + MFMA, fp32 accumulate (BASELINE configs[4] / configs[1]); their error is the
           storage format's (measured in tests/test_gpu_half.py), far from 1e-4 through 30 blocks

Activations live in the "half series" layout of include/wavenet_amd.h; torch supplies device memory (series.Lease with a
half dtype), the stream and autograd bookkeeping.  There is no CPU path.
"""
import ctypes
import os

import torch
from torch.autograd.function import once_differentiable

from . import _flags, _lib
from .functional import PARAMS_PER_BLOCK, _on_device_of_first_tensor, _p, _params_struct, _prep_params, _require_device, _stream
from .series import Lease

# The cotangent is scaled by a power of two so that max|d skips_sum| lands in [0.125, 0.25]: gradients may then grow by 2^18
# on their way back through the stack before fp16 overflows (the reference's random init grows ~sqrt(2) per block: 2^15 over
# 30 blocks).  The price is small because v_mfma_f32_32x32x16_f16 honours fp16 subnormals (tools/probes/mfma_f16_denorm.hip):
# below the normal range the lo plane just loses bits gradually -- an absolute floor of 3e-8, i.e. 2e-7 of a tensor whose
# largest element is 0.125.
GRAD_TARGET = 0.25


def _cp32(c):
    return (c + 31) // 32 * 32


class HalfLayout(object):
    """row geometry of the half series of one call (wn_hseries_layout)"""

    def __init__(self, length, max_abs_offset):
        self.length = int(length)
        self.ld, self.halo = _lib.hseries_layout(self.length, int(max_abs_offset))

    def key(self):
        return ("half", self.length, self.ld, self.halo)


class _Mode(object):
    def __init__(self, precision):
        self.name = precision
        self.code = _lib.PRECISIONS[precision]
        self.planes = 2 if precision == "f16x3" else 1
        self.dtype = torch.bfloat16 if precision == "bf16" else torch.float16


def _hlease(mode, batch, channels, layout, device):
    g = _cp32(channels) // 8
    return Lease(batch, channels, layout, device, dtype=mode.dtype, rows=mode.planes * g, pitch=layout.ld * 8)


def _shape(spec, batch, layout):
    return _lib.BlockShape(batch, layout.length, spec.ci, spec.co, spec.ms, spec.k, spec.d, int(spec.causal),
                           layout.ld, layout.halo)


_OVERFLOW_MSG = ("wavenet_speech_amd: fp16 overflow in the %s of the half-precision stack (a value beyond +-65504 after the "
                 "built-in 1/16 residual scaling); use precision='f32' or 'bf16' for this model")


def check_fp16_overflow():
    """wait for every outstanding device flag (see _flags.py) and raise if one is set"""
    _flags.check_device_flags()


_SCALE_ACC = {}      # per device: the absmax accumulator of wn_grad_scale


def _grad_scale(cotangent, mode):
    """(scale, 1 / scale) device scalars of a backward call: the dynamic power of two that puts max|cotangent| at GRAD_TARGET,
    computed on the device (no host sync).  bf16 has fp32's exponent range: its gradients need no scaling (None, None)."""
    if mode.dtype == torch.bfloat16:
        return None, None
    lib = _lib.load()
    dev = cotangent.device
    acc = _SCALE_ACC.get(dev)
    if acc is None:
        acc = _SCALE_ACC[dev] = torch.zeros(1, dtype=torch.int32, device=dev)   # (every call leaves it zero again)
    out = torch.empty(2, dtype=torch.float32, device=dev)
    c = cotangent.contiguous()
    if c.data_ptr() % 16:
        c = c.clone()                     # (a view into the middle of a buffer: the kernel reads 16-byte vectors)
    _lib.check(lib.wn_grad_scale(_p(c), c.numel(), ctypes.c_float(GRAD_TARGET), _p(out), _p(acc), _stream()), "wn_grad_scale")
    return out[0:1], out[1:2]


def _load(lib, mode, dense, lease, layout, scale, dyn, flag):
    B, C, L = dense.shape
    _lib.check(lib.wn_hseries_load(mode.code, _p(dense), _p(lease), B, C, L, layout.ld, layout.halo, ctypes.c_float(scale),
                                   _p(dyn), _p(flag), _stream()), "wn_hseries_load")


class StackPackTable(object):
    """Device-resident table of every weight-pack job of a stack (wn_hstack_pack_*): built once per (shapes, precision, pointers),
    then ONE launch per training step packs all blocks and the long-K skips_sum weights.  Parameters computed anew every step
    (anything that is not an nn.Parameter: the folded bottleneck x skip products) are declared dynamic -- the table holds their offsets inside their
    storages and each run supplies the storages' current addresses -- so steady-state training never rebuilds it."""
    MAX_DYNAMIC = 3

    @staticmethod
    def dynamic_storages(flat):
        """distinct storages of the tensors that are not nn.Parameters (recomputed every step), in first-seen order; None if too many"""
        seen = []
        for t in flat:
            if not isinstance(t, torch.nn.Parameter):
                st = t.untyped_storage()
                if all(st.data_ptr() != q.data_ptr() for q in seen):
                    seen.append(st)
        return seen if len(seen) <= StackPackTable.MAX_DYNAMIC else None

    @staticmethod
    def key_of(specs, mode, B, layout, prepped, storages, skipsum):
        dyn = [(st.data_ptr(), st.nbytes()) for st in storages]

        def rel(t):
            p = t.data_ptr()
            for i, (b, n) in enumerate(dyn):
                if b <= p < b + n:
                    return ("d", i, p - b)
            return p
        return (mode.name, B, layout.key(), bool(skipsum), tuple(n for _, n in dyn),
                tuple((s.ci, s.co, s.ms, s.k, s.d, s.causal) for s in specs), tuple(rel(t) for blk in prepped for t in blk))

    def __init__(self, lib, specs, mode, B, layout, prepped, storages, skipsum, device):
        n = len(specs)
        shapes = (_lib.BlockShape * n)(*[_shape(s, B, layout) for s in specs])
        params = (_lib.BlockParams * n)(*[_params_struct(blk) for blk in prepped])
        dyn = (_lib.MemRange * max(1, len(storages)))(*[_lib.MemRange(st.data_ptr(), st.nbytes()) for st in storages])
        nbytes = lib.wn_hstack_pack_table_bytes(n)
        host = ctypes.create_string_buffer(nbytes)
        offs = (ctypes.c_size_t * n)()
        ngroups = (n + _lib.MAX_STACK_GROUP - 1) // _lib.MAX_STACK_GROUP
        soffs = (ctypes.c_size_t * ngroups)()
        total, njobs, nblocks = ctypes.c_size_t(0), ctypes.c_int(0), ctypes.c_int(0)
        rc = lib.wn_hstack_pack_table_build(shapes, params, n, mode.code, 1 if skipsum else 0, dyn, len(storages), host, nbytes,
                                            offs, soffs, ctypes.byref(total), ctypes.byref(njobs), ctypes.byref(nblocks))
        if rc == -2:
            raise NotImplementedError("layout not covered by the pack-job table")   # WN_ERR_UNSUPPORTED: per-block packing instead
        _lib.check(rc, "wn_hstack_pack_table_build")
        self.table = torch.frombuffer(host, dtype=torch.uint8).to(device)       # the one host-to-device copy of the table's life
        self.nblk, self.njobs, self.launch_blocks, self.ndyn = n, njobs.value, nblocks.value, len(storages)
        self.block_offsets = list(offs)
        self.skipsum_offsets = list(soffs)
        self.total = total.value

    def run(self, lib, storages, device, flag=None):
        """pack everything into a fresh buffer; returns it (block l at data_ptr() + block_offsets[l])"""
        packed = torch.empty(self.total, dtype=torch.uint8, device=device)
        bases = (ctypes.c_void_p * max(1, self.ndyn))(*[st.data_ptr() for st in storages])
        _lib.check(lib.wn_hstack_pack_run(_p(self.table), self.nblk, self.njobs, self.launch_blocks, bases, self.ndyn, _p(packed),
                                          _p(flag), _stream()), "wn_hstack_pack_run")
        return packed


class _HalfStackFn(torch.autograd.Function):
    """skips_sum of a stack (modules/wavenet.py:98-100 with folded bottlenecks) on the half-precision MFMAs"""

    @staticmethod
    @_on_device_of_first_tensor
    def forward(ctx, x, specs, mode, grad_enabled, pack_cache, head, front, pool, *flat):
        """head: None, or (slope1, slope2) of an output block LeakyReLU(slope1), Conv1d 1x1, LeakyReLU(slope2), Conv1d 1x1
        (modules/wavenet.py:67-71, raw_ctcnet.py:89-93) whose parameters (w1, b1, w2, b2) are the last four tensors of `flat`:
        the block then runs inside this function, in the half series, and the function returns its output instead of skips_sum."""
        lib = _lib.load()
        _require_device(x, "input")
        _flags.WATCH.poll()
        n = len(specs)
        head_params = front_params = None
        if front is not None:      # (slope0, slope1): RawCTCNet.feature_layer = Conv1d(1 -> F, k, padding k - 1), LeakyReLU, Conv1d 1x1, LeakyReLU
            front_params = [t.detach().contiguous() for t in flat[-4:]]   # parameters w0, b0, w1, b1: the LAST four tensors of flat
            flat = flat[:-4]
        if head is not None:       # its parameters (w1, b1, w2, b2) come right before the front's
            head_params = [t.detach().contiguous() for t in flat[-4:]]
            flat = flat[:-4]
        assert len(flat) == n * PARAMS_PER_BLOCK
        B, C0, L = x.shape
        ctx.pool, ctx.in_length = int(pool), L
        if pool > 1:               # AvgPool1d(pool) of x (reference modules/classifier.py:53,102) fused into the load of the input series
            if front is not None:
                raise RuntimeError("wavenet_speech_amd: pooling and a feature layer in front of one stack are not combined")
            L = L // pool
            if L < 1:
                raise RuntimeError("wavenet_speech_amd: sequence shorter than the pooling window")
        if front is not None:
            fw0, fb0, fw1, fb1 = front_params
            if C0 != 1 or fw0.shape[1] != 1 or fw1.shape[2] != 1 or fw1.shape[1] != fw0.shape[0]:
                raise RuntimeError("wavenet_speech_amd: feature layer shapes %s, %s do not fit a one-channel signal" %
                                   (tuple(fw0.shape), tuple(fw1.shape)))
            L_in, L = L, L + fw0.shape[2] - 1     # padding k - 1 on both sides lengthens the sequence (raw_ctcnet.py:57-61)
            C0 = fw1.shape[0]
        if C0 != specs[0].ci:
            raise RuntimeError("wavenet_speech_amd: input has %d channels, first block expects %d" % (C0, specs[0].ci))
        for l in range(1, n):
            if specs[l].ci != specs[l - 1].co:
                raise RuntimeError("wavenet_speech_amd: block %d expects %d input channels but block %d produces %d"
                                   % (l, specs[l].ci, l - 1, specs[l - 1].co))
        dev = x.device
        layout = HalfLayout(L, max(s.reach() for s in specs))
        training = bool(grad_enabled) and any(ctx.needs_input_grad)
        flag = torch.zeros(1, dtype=torch.int32, device=dev) if mode.dtype == torch.float16 else None
        rs = float(lib.wn_hseries_residual_scale())
        cur = _hlease(mode, B, C0, layout, dev)
        ctx.front = None
        if front is None and pool > 1:
            _lib.check(lib.wn_hseries_load_pooled(mode.code, _p(x.detach().contiguous()), _p(cur), B, C0, ctx.in_length, int(pool),
                                                  layout.ld, layout.halo, ctypes.c_float(rs), None, _p(flag), _stream()),
                       "wn_hseries_load_pooled")
        elif front is None:
            _load(lib, mode, x.detach().contiguous(), cur, layout, rs, None, flag)
        else:
            # ---- feature layer in the series layout: the raw signal -> leaky(conv k) (elementwise kernel) -> leaky(conv 1x1) --------
            F0, k0 = fw0.shape[0], fw0.shape[2]
            xd = x.detach().contiguous()
            f1 = _hlease(mode, B, F0, layout, dev)
            _lib.check(lib.wn_hfeature_forward(mode.code, _p(xd), _p(fw0), _p(fb0), _p(f1), B, L_in, F0, k0, layout.ld, layout.halo,
                                               ctypes.c_float(rs), ctypes.c_float(front[0]), _p(flag), _stream()), "wn_hfeature_forward")
            fsh = _lib.ConvShape(B, L, F0, C0, 1, 1, 1, layout.ld, layout.halo)
            nbytes = lib.wn_hconv_packed_bytes(ctypes.byref(fsh), mode.code)
            if nbytes == 0:
                _lib.check(-1, "wn_hconv_packed_bytes")
            fpk = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            _lib.check(lib.wn_hconv_pack(ctypes.byref(fsh), mode.code, _p(fw1), _p(fb1), ctypes.c_float(rs), _p(fpk), _stream()),
                       "wn_hconv_pack")
            _lib.check(lib.wn_hconv_forward_series(ctypes.byref(fsh), mode.code, _p(fpk), _p(f1), _p(cur), ctypes.c_float(rs),
                                                   ctypes.c_float(front[1]), _p(flag), _stream()), "wn_hconv_forward_series")
            if training:
                ctx.front = (front, xd, f1, fsh, fpk, L_in, [tuple(t.shape) for t in front_params])
        ms = specs[0].ms
        S = torch.empty(B, ms, L, dtype=torch.float32, device=dev)
        saved, skip_w, skip_b = [], [], []
        zbuf = None
        for spec in specs:
            if spec.ms != ms:
                raise RuntimeError("wavenet_speech_amd: all blocks of a stack must share out_dim")
        prepped = [_prep_params(flat[l * PARAMS_PER_BLOCK:(l + 1) * PARAMS_PER_BLOCK], spec) for l, spec in enumerate(specs)]
        # every pack job of the stack (5 per block + in training the long-K skips_sum weights) from ONE launch of a device-resident
        # job table (StackPackTable); WN_PACK_TABLE=0 or an unsupported layout falls back to the per-block entry points
        table = packed_all = None
        frozen = pack_cache is not None and pack_cache.frozen and not grad_enabled
        if pack_cache is not None and not frozen and os.environ.get("WN_PACK_TABLE", "1") != "0":
            storages = StackPackTable.dynamic_storages(flat)
            if storages is not None:
                key = StackPackTable.key_of(specs, mode, B, layout, prepped, storages, training)
                table = pack_cache.tables.get(key)
                if table is None:
                    try:
                        table = StackPackTable(lib, specs, mode, B, layout, prepped, storages, training, dev)
                    except NotImplementedError:
                        table = False          # e.g. skip biases that are not equally spaced: per-block packing
                    if len(pack_cache.tables) >= 8:
                        pack_cache.tables.clear()
                    pack_cache.tables[key] = table
                if table:
                    packed_all = table.run(lib, storages, dev, flag)
                else:
                    table = None
        for l, spec in enumerate(specs):
            shape = _shape(spec, B, layout)
            params = prepped[l]
            if table is not None:
                packed = packed_all.data_ptr() + table.block_offsets[l]
            else:
                packed = pack_cache.get(l, layout, B) if (pack_cache is not None and pack_cache.frozen and not grad_enabled) else None
            if packed is None:
                nbytes = lib.wn_hblock_packed_bytes(ctypes.byref(shape), mode.code)
                if nbytes == 0:
                    _lib.check(-1, "wn_hblock_packed_bytes")
                packed = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                ps = _params_struct(params)
                _lib.check(lib.wn_hblock_pack_checked(ctypes.byref(shape), mode.code, ctypes.byref(ps), _p(packed), _p(flag), _stream()),
                           "wn_hblock_pack_checked")
                if pack_cache is not None and pack_cache.frozen and not grad_enabled:
                    pack_cache.put(l, layout, B, packed)
            r = _hlease(mode, B, spec.co, layout, dev) if l + 1 < n else None
            if training:
                sg, z = (_hlease(mode, B, spec.co, layout, dev) for _ in range(2))   # kept for backward; tanh = z / sg
            elif lib.wn_hblock_forward_is_fused(ctypes.byref(shape), mode.code):
                sg = z = None                  # inference through the fused kernel: z stays on the chip
            else:
                sg = None
                if zbuf is None or zbuf.channels != spec.co:
                    zbuf = _hlease(mode, B, spec.co, layout, dev)
                z = zbuf
            _lib.check(lib.wn_hblock_forward(ctypes.byref(shape), mode.code, _p(packed), _p(cur), _p(r),
                                             None if training else _p(S), 0 if l == 0 else 1, _p(sg), _p(z),
                                             _p(flag), _stream()), "wn_hblock_forward")
            if training:
                saved.append((cur, sg, z, packed, shape))
                skip_w.append(params[6])
                skip_b.append(params[7])
            cur = r
        series_head = head is not None and training and n <= _lib.MAX_STACK_GROUP
        ctx.skipsum_packed = None
        if training:
            G = _lib.MAX_STACK_GROUP
            bias_total = torch.stack(skip_b).sum(0).contiguous() if table is None else None
            for gi, g0 in enumerate(range(0, n, G)):
                idx = range(g0, min(g0 + G, n))
                m = len(idx)
                shape = _lib.SkipSumShape(B, L, ms, m, layout.ld, layout.halo)
                for i, l in enumerate(idx):
                    shape.channels[i] = specs[l].co
                zptrs = (ctypes.c_void_p * m)(*[saved[l][2].ptr for l in idx])
                if table is not None:
                    packed = packed_all.data_ptr() + table.skipsum_offsets[gi]
                else:
                    nbytes = lib.wn_hskipsum_packed_bytes(ctypes.byref(shape), mode.code)
                    if nbytes == 0:
                        _lib.check(-1, "wn_hskipsum_packed_bytes")
                    packed = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                    wptrs = (ctypes.c_void_p * m)(*[skip_w[l].data_ptr() for l in idx])
                    _lib.check(lib.wn_hskipsum_pack(ctypes.byref(shape), mode.code, wptrs, _p(bias_total) if g0 == 0 else None,
                                                    _p(packed), _stream()), "wn_hskipsum_pack")
                if series_head:
                    ctx.skipsum_packed = packed            # the head below writes leaky(S) as a series: no dense S
                else:
                    _lib.check(lib.wn_hskipsum_forward(ctypes.byref(shape), mode.code, _p(packed), zptrs, _p(S),
                                                       0 if g0 == 0 else 1, _stream()), "wn_hskipsum_forward")
        ctx.packed_all = packed_all       # the blocks' packed weights live here until backward has run
        ctx.head = None
        if head is not None:
            # ---- output block in the series layout: leaky(S) -> conv1 -> leaky -> conv2 (dense fp32 out) ------------------------
            w1, b1, w2, b2 = head_params
            c1, c2 = w1.shape[0], w2.shape[0]
            if w1.shape[1] != ms or w2.shape[1] != c1 or w1.shape[2] != 1 or w2.shape[2] != 1:
                raise RuntimeError("wavenet_speech_amd: output block shapes %s, %s do not follow a stack of out_dim %d"
                                   % (tuple(w1.shape), tuple(w2.shape), ms))
            sh1 = _lib.ConvShape(B, L, ms, c1, 1, 1, 1, layout.ld, layout.halo)
            sh2 = _lib.ConvShape(B, L, c1, c2, 1, 1, 1, layout.ld, layout.halo)
            pk = []
            for sh, w, b_ in ((sh1, w1, b1), (sh2, w2, b2)):
                nbytes = lib.wn_hconv_packed_bytes(ctypes.byref(sh), mode.code)
                if nbytes == 0:
                    _lib.check(-1, "wn_hconv_packed_bytes")
                p_ = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                _lib.check(lib.wn_hconv_pack(ctypes.byref(sh), mode.code, _p(w), _p(b_), ctypes.c_float(rs), _p(p_), _stream()),
                           "wn_hconv_pack")
                pk.append(p_)
            h0 = _hlease(mode, B, ms, layout, dev)
            if series_head:
                # the long-K skips_sum product writes leaky(S) / 16 straight into the series: no dense fp32 S at all
                skshape = _lib.SkipSumShape(B, L, ms, n, layout.ld, layout.halo)
                for i in range(n):
                    skshape.channels[i] = specs[i].co
                zptrs = (ctypes.c_void_p * n)(*[saved[l][2].ptr for l in range(n)])
                _lib.check(lib.wn_hskipsum_forward_series(ctypes.byref(skshape), mode.code, _p(ctx.skipsum_packed), zptrs, _p(h0),
                                                          ctypes.c_float(rs), ctypes.c_float(head[0]), _p(flag), _stream()),
                           "wn_hskipsum_forward_series")
            else:
                _load(lib, mode, torch.nn.functional.leaky_relu(S, head[0]), h0, layout, rs, None, flag)
            h1 = _hlease(mode, B, c1, layout, dev)
            _lib.check(lib.wn_hconv_forward_series(ctypes.byref(sh1), mode.code, _p(pk[0]), _p(h0), _p(h1), ctypes.c_float(rs),
                                                   ctypes.c_float(head[1]), _p(flag), _stream()), "wn_hconv_forward_series")
            y = torch.empty(B, c2, L, dtype=torch.float32, device=dev)
            _lib.check(lib.wn_hconv_forward(ctypes.byref(sh2), mode.code, _p(pk[1]), _p(h1), _p(y), _stream()), "wn_hconv_forward")
            if training:
                ctx.head = (head, sh1, sh2, pk, h0, h1, [tuple(t.shape) for t in head_params])
            S = y
        _flags.WATCH.note(flag, _OVERFLOW_MSG % "forward pass", at_once=not training)
        ctx.specs, ctx.saved, ctx.layout, ctx.batch, ctx.mode = specs, saved, layout, B, mode
        ctx.param_shapes = [tuple(t.shape) for t in flat]
        return S

    @staticmethod
    @once_differentiable
    @_on_device_of_first_tensor
    def backward(ctx, d_skips):
        lib = _lib.load()
        _flags.WATCH.poll()
        specs, layout, B, mode = ctx.specs, ctx.layout, ctx.batch, ctx.mode
        dev = d_skips.device
        d_skips = d_skips.contiguous()
        flag = torch.zeros(1, dtype=torch.int32, device=dev) if mode.dtype == torch.float16 else None
        dyn, dyn_inv = _grad_scale(d_skips, mode)
        head_grads = []
        if ctx.head is None:
            dS = _hlease(mode, B, specs[0].ms, layout, dev)
            _load(lib, mode, d_skips, dS, layout, 1.0, dyn, flag)
        else:
            # ---- output block, backwards, in the series: d_skips is the cotangent of its OUTPUT here ----------------------------
            (slope1, slope2), sh1, sh2, pk, h0, h1, hshapes = ctx.head
            rs = float(lib.wn_hseries_residual_scale())
            c1, c2 = sh1.out_channels, sh2.out_channels
            dY = _hlease(mode, B, c2, layout, dev)
            _load(lib, mode, d_skips, dY, layout, 1.0, dyn, flag)
            dh1 = _hlease(mode, B, c1, layout, dev)
            _lib.check(lib.wn_hconv_backward_data_series(ctypes.byref(sh2), mode.code, _p(pk[1]), _p(dY), _p(h1), ctypes.c_float(slope2),
                                                         _p(dh1), _p(flag), _stream()), "wn_hconv_backward_data_series")
            dS = _hlease(mode, B, specs[0].ms, layout, dev)
            _lib.check(lib.wn_hconv_backward_data_series(ctypes.byref(sh1), mode.code, _p(pk[0]), _p(dh1), _p(h0), ctypes.c_float(slope1),
                                                         _p(dS), _p(flag), _stream()), "wn_hconv_backward_data_series")
            for sh, xin, dy_, shp_w, shp_b in ((sh1, h0, dh1, hshapes[0], hshapes[1]), (sh2, h1, dY, hshapes[2], hshapes[3])):
                dw = torch.empty(shp_w, dtype=torch.float32, device=dev)
                db = torch.empty(shp_b, dtype=torch.float32, device=dev)
                ws_bytes = lib.wn_hconv_wgrad_workspace_bytes(ctypes.byref(sh), mode.code)
                ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=dev)
                _lib.check(lib.wn_hconv_backward_weights(ctypes.byref(sh), mode.code, _p(xin), _p(dy_), ctypes.c_float(rs), _p(dw), _p(db),
                                                         _p(dyn_inv), _p(ws), ws_bytes, _stream()), "wn_hconv_backward_weights")
                head_grads += [dw, db]
            ctx.head = None
        dr = None
        dx0 = None
        grads_flat = [None] * (len(specs) * PARAMS_PER_BLOCK)
        # weight gradients: small blocks (<= 128 channels) are two or three gradient tiles each -- their operands are kept and
        # several blocks go into ONE split-K launch + ONE reduction (wn_hblocks_backward_weights); WN_WGRAD_GROUP=1 = per block
        group_max = lib.wn_hblocks_wgrad_group_max(ctypes.byref(ctx.saved[0][4]), mode.code)
        env_group = os.environ.get("WN_WGRAD_GROUP")
        if env_group:
            group_max = max(1, min(group_max, int(env_group)))
        pending = []          # (l, shape, x, z, da, dg, dr, grads) of blocks whose weight gradients are not launched yet

        def flush():
            if not pending:
                return
            m = len(pending)
            shapes = (_lib.BlockShape * m)(*[e[1] for e in pending])
            arr = lambda i: (ctypes.c_void_p * m)(*[(e[i].ptr if e[i] is not None else None) for e in pending])
            dsk = (ctypes.c_void_p * m)(*[dS.ptr] * m)
            gs = (_lib.BlockParams * m)(*[_params_struct(e[7]) for e in pending])
            ws_bytes = lib.wn_hblocks_wgrad_workspace_bytes(shapes, m, mode.code)
            if ws_bytes == 0:
                _lib.check(-1, "wn_hblocks_wgrad_workspace_bytes")
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            _lib.check(lib.wn_hblocks_backward_weights(shapes, m, mode.code, arr(2), arr(3), arr(4), arr(5), arr(6), dsk, gs,
                                                       _p(dyn_inv), _p(ws), ws_bytes, _stream()), "wn_hblocks_backward_weights")
            del pending[:]

        # dz is pointwise in time and its dr is the dx of the block above: where two consecutive blocks both take the column-owner
        # kernels, dx of the upper and dz of the lower run as ONE launch (wn_hblock_backward_pair): the chain is then
        # dz(top), [dx(l) + dz(l - 1)] ..., dx(bottom) -- n + 1 launches instead of 2 n.
        gates = None          # (da, dg) of the block about to be processed, if the pair launch above it has produced them
        for l in range(len(specs) - 1, -1, -1):
            spec = specs[l]
            x, sg, z, packed, shape = ctx.saved[l]
            have_dz = gates is not None
            da, dg = gates if have_dz else (_hlease(mode, B, spec.co, layout, dev), _hlease(mode, B, spec.co, layout, dev))
            gates = None
            dx = dxd = None
            if l > 0 or ctx.front is not None:
                dx = _hlease(mode, B, spec.ci, layout, dev)
            elif ctx.needs_input_grad[0]:
                dxd = dx0 = torch.empty(B, spec.ci, layout.length, dtype=torch.float32, device=dev)
            paired = l > 0 and lib.wn_hblock_backward_pair_is_fused(ctypes.byref(shape), ctypes.byref(ctx.saved[l - 1][4]), mode.code) == 1
            if paired and not have_dz:
                # the top of a chain: its own dz first (dz only: no dx destination)
                _lib.check(lib.wn_hblock_backward_data(ctypes.byref(shape), mode.code, _p(packed), _p(dr), _p(dS), _p(z), _p(sg),
                                                       _p(da), _p(dg), None, None, _p(dyn_inv), _p(flag), _stream()),
                           "wn_hblock_backward_data")
                have_dz = True
            if paired:
                _xl, sgl, zl, packedl, shapel = ctx.saved[l - 1]
                dal, dgl = _hlease(mode, B, specs[l - 1].co, layout, dev), _hlease(mode, B, specs[l - 1].co, layout, dev)
                _lib.check(lib.wn_hblock_backward_pair(ctypes.byref(shape), _p(packed), ctypes.byref(shapel), _p(packedl), mode.code,
                                                       _p(dr), _p(da), _p(dg), _p(dS), _p(zl), _p(sgl), _p(dx), _p(dal), _p(dgl),
                                                       _p(flag), _stream()), "wn_hblock_backward_pair")
                gates = (dal, dgl)
            elif have_dz:
                # the bottom of a chain: the input gradient alone (series, masked by the feature layer's LeakyReLU, or dense)
                masked = l == 0 and ctx.front is not None
                _lib.check(lib.wn_hblock_backward_input(ctypes.byref(shape), mode.code, _p(packed), _p(dr), _p(da), _p(dg), _p(dx), _p(dxd),
                                                        _p(dyn_inv), _p(x) if masked else None,
                                                        ctypes.c_float(ctx.front[0][1] if masked else 1.0), _p(flag), _stream()),
                           "wn_hblock_backward_input")
            elif l == 0 and ctx.front is not None:
                # the stack's input is leaky(feature conv): its LeakyReLU backward rides in this block's dx epilogue (x = the stored activation)
                _lib.check(lib.wn_hblock_backward_data_masked(ctypes.byref(shape), mode.code, _p(packed), _p(dr), _p(dS), _p(z), _p(sg),
                                                              _p(da), _p(dg), _p(dx), _p(x), ctypes.c_float(ctx.front[0][1]), _p(flag),
                                                              _stream()), "wn_hblock_backward_data_masked")
            else:
                _lib.check(lib.wn_hblock_backward_data(ctypes.byref(shape), mode.code, _p(packed), _p(dr), _p(dS), _p(z), _p(sg),
                                                       _p(da), _p(dg), _p(dx), _p(dxd), _p(dyn_inv), _p(flag), _stream()),
                           "wn_hblock_backward_data")
            k = spec.k
            shapes = [(spec.co, spec.ci, k), (spec.co,), (spec.co, spec.ci, k), (spec.co,), (spec.co, spec.co), (spec.co,),
                      (spec.ms, spec.co), (spec.ms,), (spec.co, spec.ci), (spec.co,)]
            unused = (4, 5, 8, 9) if dr is None else ()
            grads = [None if i in unused else torch.empty(s, dtype=torch.float32, device=dev) for i, s in enumerate(shapes)]
            if group_max > 1:
                pending.append((l, shape, x, z, da, dg, dr, grads))
                if len(pending) >= group_max:
                    flush()
            else:
                ws_bytes = lib.wn_hblock_wgrad_workspace_bytes(ctypes.byref(shape), mode.code)
                ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=dev)
                gs = _params_struct(grads)
                _lib.check(lib.wn_hblock_backward_weights(ctypes.byref(shape), mode.code, _p(x), _p(z), _p(da), _p(dg), _p(dr),
                                                          _p(dS), ctypes.byref(gs), _p(dyn_inv), _p(ws), ws_bytes, _stream()),
                           "wn_hblock_backward_weights")
            grads_flat[l * PARAMS_PER_BLOCK:(l + 1) * PARAMS_PER_BLOCK] = grads
            dr = dx
            ctx.saved[l] = None
        flush()
        front_grads = []
        if ctx.front is not None:
            # ---- feature layer, backwards: dr is now the (masked) gradient of the stack's input, in the series ---------------------
            (slope0, slope1), xd, f1, fsh, fpk, L_in, fshapes = ctx.front
            rs = float(lib.wn_hseries_residual_scale())
            F0 = fsh.in_channels
            df1 = _hlease(mode, B, F0, layout, dev)
            _lib.check(lib.wn_hconv_backward_data_series(ctypes.byref(fsh), mode.code, _p(fpk), _p(dr), _p(f1), ctypes.c_float(slope0),
                                                         _p(df1), _p(flag), _stream()), "wn_hconv_backward_data_series")
            dw1 = torch.empty(fshapes[2], dtype=torch.float32, device=dev)
            db1 = torch.empty(fshapes[3], dtype=torch.float32, device=dev)
            ws_bytes = lib.wn_hconv_wgrad_workspace_bytes(ctypes.byref(fsh), mode.code)
            ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=dev)
            _lib.check(lib.wn_hconv_backward_weights(ctypes.byref(fsh), mode.code, _p(f1), _p(dr), ctypes.c_float(rs), _p(dw1), _p(db1),
                                                     _p(dyn_inv), _p(ws), ws_bytes, _stream()), "wn_hconv_backward_weights")
            k0 = fshapes[0][2]
            dw0 = torch.empty(fshapes[0], dtype=torch.float32, device=dev)
            db0 = torch.empty(fshapes[1], dtype=torch.float32, device=dev)
            ws_bytes = lib.wn_hfeature_wgrad_workspace_bytes(B, L_in, F0, k0)
            ws0 = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=dev)
            _lib.check(lib.wn_hfeature_backward_weights(mode.code, _p(xd), _p(df1), ctypes.c_float(1.0), _p(dw0), _p(db0), B, L_in, F0, k0,
                                                        layout.ld, layout.halo, _p(dyn_inv), _p(ws0), ws_bytes, _stream()),
                       "wn_hfeature_backward_weights")
            front_grads = [dw0, db0, dw1, db1]
            ctx.front = None
        _flags.WATCH.note(flag, _OVERFLOW_MSG % "backward pass", at_once=False)
        grads_flat = [None if g is None else g.view(shp) for g, shp in zip(grads_flat, ctx.param_shapes)]
        if dx0 is not None and ctx.pool > 1:
            from .functional import _unpool
            dx0 = _unpool(lib, dx0, ctx.in_length, ctx.pool)
        return (dx0, None, None, None, None, None, None, None) + tuple(grads_flat) + tuple(head_grads) + tuple(front_grads)


def residual_stack(x, specs, flat_params, precision, pack_cache=None, head=None, front=None, pool=1):
    """head: None, or ((slope1, slope2), [w1, b1, w2, b2]) of an output block LeakyReLU, Conv1d 1x1, LeakyReLU, Conv1d 1x1 that is
    to run inside the same function, in the half series: the result is then that block's output, not skips_sum.
    front: None, or ((slope0, slope1), [w0, b0, w1, b1]) of a feature layer Conv1d(1 -> F, k, padding k - 1), LeakyReLU, Conv1d 1x1,
    LeakyReLU in front of the stack: x is then the raw one-channel signal [B, 1, L]."""
    if precision not in ("f16x3", "f16", "bf16"):
        raise ValueError("unknown precision %r" % (precision,))
    flat = list(flat_params)
    hs = fs = None
    if head is not None:
        hs = (float(head[0][0]), float(head[0][1]))
        flat += list(head[1])
    if front is not None:
        fs = (float(front[0][0]), float(front[0][1]))
        flat += list(front[1])
    return _HalfStackFn.apply(x, tuple(specs), _Mode(precision), torch.is_grad_enabled(), pack_cache, hs, fs, int(pool), *flat)


class _HalfConvFn(torch.autograd.Function):
    """CausalConv1d / NonCausalConv1d / 1x1 Conv1d (modules/conv_ops.py:39-44, 73-79) on the half-precision kernels
    (wn_hconv_*): dense fp32 in and out like functional.dilated_conv, half series and MFMAs inside.  Used by the conv
    modules and the output stacks of a model that set_precision switched to a half mode."""

    @staticmethod
    @_on_device_of_first_tensor
    def forward(ctx, x, weight, bias, dilation, causal, mode, grad_enabled):
        lib = _lib.load()
        _require_device(x, "input")
        _require_device(weight, "weight")
        _flags.WATCH.poll()
        B, Ci, L = x.shape
        Co, Ci_w, k = weight.shape
        if Ci_w != Ci:
            raise RuntimeError("wavenet_speech_amd: input has %d channels, conv expects %d" % (Ci, Ci_w))
        dev = x.device
        reach = max(abs(o) for o in _lib.tap_offsets(k, dilation, causal))
        layout = HalfLayout(L, reach)
        shape = _lib.ConvShape(B, L, Ci, Co, k, int(dilation), int(bool(causal)), layout.ld, layout.halo)
        training = bool(grad_enabled) and any(ctx.needs_input_grad)
        flag = torch.zeros(1, dtype=torch.int32, device=dev) if mode.dtype == torch.float16 else None
        rs = float(lib.wn_hseries_residual_scale())           # inputs are stored as x / 16 like the residual stream
        nbytes = lib.wn_hconv_packed_bytes(ctypes.byref(shape), mode.code)
        if nbytes == 0:
            _lib.check(-1 if k <= _lib.MAX_TAPS else -2, "wn_hconv_packed_bytes")
        packed = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        w = weight.detach().contiguous()
        b = bias.detach().contiguous() if bias is not None else None
        _lib.check(lib.wn_hconv_pack(ctypes.byref(shape), mode.code, _p(w), _p(b), ctypes.c_float(rs), _p(packed), _stream()),
                   "wn_hconv_pack")
        xin = _hlease(mode, B, Ci, layout, dev)
        _load(lib, mode, x.detach().contiguous(), xin, layout, rs, None, flag)
        y = torch.empty(B, Co, L, dtype=torch.float32, device=dev)
        _lib.check(lib.wn_hconv_forward(ctypes.byref(shape), mode.code, _p(packed), _p(xin), _p(y), _stream()), "wn_hconv_forward")
        _flags.WATCH.note(flag, _OVERFLOW_MSG % "input of a conv", at_once=not training)
        if training:
            ctx.saved = (xin, packed, shape)
        ctx.layout, ctx.dims, ctx.has_bias, ctx.mode, ctx.rs = layout, (B, Ci, Co, k), bias is not None, mode, rs
        return y

    @staticmethod
    @once_differentiable
    @_on_device_of_first_tensor
    def backward(ctx, d_y):
        lib = _lib.load()
        _flags.WATCH.poll()
        xin, packed, shape = ctx.saved
        layout, mode = ctx.layout, ctx.mode
        B, Ci, Co, k = ctx.dims
        dev = d_y.device
        d_y = d_y.contiguous()
        flag = torch.zeros(1, dtype=torch.int32, device=dev) if mode.dtype == torch.float16 else None
        dyn, dyn_inv = _grad_scale(d_y, mode)
        dy = _hlease(mode, B, Co, layout, dev)
        _load(lib, mode, d_y, dy, layout, 1.0, dyn, flag)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(B, Ci, layout.length, dtype=torch.float32, device=dev)
            _lib.check(lib.wn_hconv_backward_data(ctypes.byref(shape), mode.code, _p(packed), _p(dy), _p(dx), _p(dyn_inv), _stream()),
                       "wn_hconv_backward_data")
        dw = torch.empty(Co, Ci, k, dtype=torch.float32, device=dev)
        db = torch.empty(Co, dtype=torch.float32, device=dev) if ctx.has_bias else None
        ws_bytes = lib.wn_hconv_wgrad_workspace_bytes(ctypes.byref(shape), mode.code)
        ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=dev)
        _lib.check(lib.wn_hconv_backward_weights(ctypes.byref(shape), mode.code, _p(xin), _p(dy), ctypes.c_float(ctx.rs), _p(dw),
                                                 _p(db), _p(dyn_inv), _p(ws), ws_bytes, _stream()), "wn_hconv_backward_weights")
        _flags.WATCH.note(flag, _OVERFLOW_MSG % "gradient of a conv", at_once=False)
        ctx.saved = None
        return dx, dw, db, None, None, None, None


def conv(x, weight, bias, dilation, causal, precision):
    """functional.dilated_conv in a half-precision mode ("f16x3" / "f16" / "bf16")"""
    if precision not in ("f16x3", "f16", "bf16"):
        raise ValueError("unknown precision %r" % (precision,))
    return _HalfConvFn.apply(x, weight, bias, int(dilation), bool(causal), _Mode(precision), torch.is_grad_enabled())

