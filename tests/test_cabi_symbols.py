"""CPU: libwavenet_amd.so loads without a GPU and exports every symbol include/wavenet_amd.h declares
(no compute calls here -- those need a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "wavenet_amd.h")


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(wn_[a-z0-9_]+)\s*\(", src)
    return sorted(set(n for n in names))


def test_header_declares_the_expected_surface():
    names = _declared_functions()
    for must in ("wn_block_pack", "wn_block_forward", "wn_block_backward_data", "wn_block_backward_weights",
                 "wn_conv_forward", "wn_series_layout", "wn_strerror", "wn_version"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from wavenet_speech_amd import _lib
    lib = _lib.load()
    for name in _declared_functions():
        assert hasattr(lib, name), "libwavenet_amd.so does not export %s" % name
    # and the ctypes table covers the header one-to-one
    assert sorted(_lib.SIGNATURES) == _declared_functions()


def test_host_only_entry_points():
    from wavenet_speech_amd import _lib
    lib = _lib.load()
    assert lib.wn_version() == 300
    assert lib.wn_strerror(0) == b"ok"
    assert b"workspace" in lib.wn_strerror(-5)
    assert lib.wn_round_up(13, 8) == 16
    assert lib.wn_autopad(2, 3) == 2 and lib.wn_autopad(2, 4) == 2 and lib.wn_autopad(3, 5) == 5
    assert _lib.tap_offsets(2, 512, True) == [-512, 0]
    assert _lib.tap_offsets(2, 3, False) == [-2, 1]
    assert _lib.tap_offsets(5, 3, True) == [-12, -9, -6, -3, 0]
    ld, halo = _lib.series_layout(16000, 512)
    assert (ld, halo) == (512 + 16000 + 512, 512)
    ld, halo = _lib.series_layout(130, 3)
    assert halo == 4 and ld == 4 + 256 + 4
    assert lib.wn_series_floats(2, 5, 264) == 2 * 8 * 264


def test_shape_errors_are_reported_not_crashed():
    from wavenet_speech_amd import _lib
    lib = _lib.load()
    ok = _lib.BlockShape(1, 100, 8, 8, 8, 2, 4, 1, 4 + 128 + 4, 4)
    assert lib.wn_block_packed_bytes(ctypes.byref(ok)) > 0
    assert lib.wn_block_wgrad_workspace_bytes(ctypes.byref(ok)) > 0
    small_halo = _lib.BlockShape(1, 100, 8, 8, 8, 2, 8, 1, 4 + 128 + 4, 4)     # taps reach 8 > halo 4
    assert lib.wn_block_packed_bytes(ctypes.byref(small_halo)) == 0
    assert lib.wn_block_forward(ctypes.byref(small_halo), None, None, None, None, 0, None, None, None) == -1
    too_wide = _lib.BlockShape(1, 100, 8, 8, 8, 9, 1, 1, 8 + 128 + 8, 8)       # kernel_width 9 > WN_MAX_TAPS
    assert lib.wn_block_forward(ctypes.byref(too_wide), None, None, None, None, 0, None, None, None) == -2
    assert lib.wn_block_forward(ctypes.byref(ok), None, None, None, None, 0, None, None, None) == -3  # NULL ptrs


def test_missing_library_is_loud(monkeypatch):
    from wavenet_speech_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libwavenet_amd.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()
