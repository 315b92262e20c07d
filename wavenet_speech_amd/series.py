"""
Padded series layout (see include/wavenet_amd.h): device buffers float[B][round_up(C,8)][ld] whose
valid [C][L] window starts at column `halo`; everything else is zero and stays zero (the kernels
only ever write the valid window).  torch is used here purely as the device allocator.
"""
import collections
import os
import threading

import torch

from . import _lib


def round_up(x, m):
    return (x + m - 1) // m * m


class SeriesLayout(object):
    """Row geometry shared by every series of one call: L valid steps, `halo` zero columns either side."""

    def __init__(self, length, max_abs_offset):
        self.length = int(length)
        self.ld, self.halo = _lib.series_layout(self.length, int(max_abs_offset))

    def key(self):
        return (self.length, self.ld, self.halo)


class _Pool(object):
    """Free-list of zero-padded series buffers.  A buffer handed out by `lease` goes back to the pool when
    its Lease is garbage collected (i.e. when the autograd context that saved it dies), so several forward
    passes may be in flight without aliasing.  Buffers keep their zero padding for life, which is what
    lets the pool skip the memset a fresh allocation would need.

    Keys are exact shapes (batch, channels, length, halo, ld), so training on variable-length utterances
    would otherwise accumulate one set of buffers per distinct length: the idle (free) part of the pool is
    capped (WN_POOL_CAP_GB; default 45 % of the device's memory, so that the saved activations of one step of the
    largest configurations -- 36 GB at 256 ch x 30 blocks x 16 x 16k, 48 GB at 512 ch x 60 blocks x 2 x 48k -- are
    recycled rather than re-zeroed) and least-recently-used shapes are dropped first."""

    def __init__(self):
        self._free = collections.OrderedDict()   # key -> list of tensors, most recently used key last
        self.hold = None                         # while a HIP graph is being captured: every buffer handed out is also kept here
        self._lock = threading.Lock()
        self.free_bytes = 0
        env = os.environ.get("WN_POOL_CAP_GB")
        self.cap_bytes = int(float(env) * (1 << 30)) if env else None   # None: resolved on first use

    def take(self, key):
        with self._lock:
            lst = self._free.get(key)
            if lst:
                t = lst.pop()
                self.free_bytes -= t.numel() * t.element_size()
                self._free.move_to_end(key)
                return t
        return None

    def _cap(self, tensor):
        if self.cap_bytes is None:
            try:
                total = torch.cuda.get_device_properties(tensor.device).total_memory if tensor.is_cuda else 64 << 30
            except Exception:
                total = 64 << 30
            self.cap_bytes = int(0.45 * total)
        return self.cap_bytes

    def give(self, key, tensor):
        cap = self._cap(tensor)
        with self._lock:
            self._free.setdefault(key, []).append(tensor)
            self._free.move_to_end(key)
            self.free_bytes += tensor.numel() * tensor.element_size()
            while self.free_bytes > cap and self._free:
                old_key = next(iter(self._free))
                if old_key == key and len(self._free) == 1:
                    break
                for t in self._free.pop(old_key):
                    self.free_bytes -= t.numel() * t.element_size()

    def clear(self):
        with self._lock:
            self._free.clear()
            self.free_bytes = 0


POOL = _Pool()


class Lease(object):
    """A pooled series buffer.  `.t` is the [B][Cp][ld] tensor, `.ptr` its device address."""
    __slots__ = ("t", "ptr", "_key", "channels", "layout", "__weakref__")

    def __init__(self, batch, channels, layout, device, dtype=torch.float32, rows=None, pitch=None):
        """rows / pitch override the [round_up(C, 8)][ld] plane geometry (the half-precision layouts of
        functional_half use [planes * C/8][ld * 8] halves)."""
        cp = round_up(channels, 8) if rows is None else rows
        ld = layout.ld if pitch is None else pitch
        # The zero-padding invariant is tied to the exact valid window, so C, L and halo are part of the key -- and so is
        # the STREAM the lease is taken on: a buffer returns to the pool when its last Python reference dies, which can be
        # while its last kernel is still queued.  Handing it to a later lease on the same stream is safe (stream order);
        # handing it to another stream would not be, so buffers never migrate between streams.
        dev = torch.device(device)
        stream = torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0
        self._key = (str(dev), stream, batch, channels, layout.length, layout.halo, layout.ld, dtype, cp, ld)
        t = POOL.take(self._key)
        if t is None:
            try:
                t = torch.zeros(batch, cp, ld, dtype=dtype, device=dev)
            except torch.OutOfMemoryError:
                # the pool's idle buffers are invisible to torch's caching allocator: give them back and retry once
                POOL.clear()
                torch.cuda.empty_cache()
                t = torch.zeros(batch, cp, ld, dtype=dtype, device=dev)
        if POOL.hold is not None:
            POOL.hold.append(t)      # a captured graph addresses this buffer for as long as it is replayed (graphs.GraphedStep)
        self.t = t
        self.ptr = t.data_ptr()
        self.channels = channels
        self.layout = layout

    def view(self):
        """[B, C, L] strided view of the valid window."""
        lay = self.layout
        return self.t[:, :self.channels, lay.halo:lay.halo + lay.length]

    def __del__(self):
        try:
            POOL.give(self._key, self.t)
        except Exception:
            pass


def fresh_series(batch, channels, layout, device):
    """A zero-initialised series that is NOT pooled (for tensors handed back to the caller)."""
    return torch.zeros(batch, round_up(channels, 8), layout.ld, dtype=torch.float32, device=device)


def window(t, channels, layout):
    return t[:, :channels, layout.halo:layout.halo + layout.length]


def load_series(dst, x, layout):
    """Copy a dense [B, C, L] tensor into the valid window of a padded buffer."""
    window(dst, x.shape[1], layout).copy_(x)
    return dst
