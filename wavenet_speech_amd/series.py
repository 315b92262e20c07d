"""
Padded series layout (see include/wavenet_amd.h): device buffers float[B][round_up(C,8)][ld] whose
valid [C][L] window starts at column `halo`; everything else is zero and stays zero (the kernels
only ever write the valid window).  torch is used here purely as the device allocator.
"""
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
    lets the pool skip the memset a fresh allocation would need."""

    def __init__(self):
        self._free = {}
        self._lock = threading.Lock()
        self.allocated_bytes = 0

    def take(self, key):
        with self._lock:
            lst = self._free.get(key)
            if lst:
                return lst.pop()
        return None

    def give(self, key, tensor):
        with self._lock:
            self._free.setdefault(key, []).append(tensor)

    def clear(self):
        with self._lock:
            self._free.clear()
            self.allocated_bytes = 0


POOL = _Pool()


class Lease(object):
    """A pooled series buffer.  `.t` is the [B][Cp][ld] tensor, `.ptr` its device address."""
    __slots__ = ("t", "ptr", "_key", "channels", "layout", "__weakref__")

    def __init__(self, batch, channels, layout, device):
        cp = round_up(channels, 8)
        # the zero-padding invariant is tied to the exact valid window, so C, L and halo are part of the key
        self._key = (str(device), batch, channels, layout.length, layout.halo, layout.ld)
        t = POOL.take(self._key)
        if t is None:
            t = torch.zeros(batch, cp, layout.ld, dtype=torch.float32, device=device)
            POOL.allocated_bytes += t.numel() * 4
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
