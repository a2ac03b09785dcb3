"""The C-ABI library loads (no GPU needed) and exports every symbol include/*.h declares; argument checks
that run before any launch return the documented error codes."""
import ctypes
import glob
import os
import re

import pytest

from scene_net_amd import _hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    names = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        src = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names.update(re.findall(r"\b(sn_[a-z_0-9]+)\s*\(", src))
    return names


def test_header_symbols_are_bound_and_exported():
    declared = _declared_symbols()
    assert {"sn_geneo_bank", "sn_conv_bank", "sn_voxel_bbox", "sn_voxel_desc", "sn_voxel_scatter",
            "sn_voxel_finalize", "sn_voxel_occupancy", "sn_voxel_prepare", "sn_last_error", "sn_version"} <= declared
    assert declared == set(_hip.SYMBOLS), "ctypes table and header disagree"
    lib = ctypes.CDLL(_hip.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name


def test_version_and_error_text():
    lib = _hip.load()
    assert lib.sn_version() >= 100
    assert isinstance(lib.sn_last_error(), bytes)
    assert lib.sn_device_count() >= 0


def test_argument_checks_need_no_gpu():
    lib = _hip.load()
    # null pointers / bad extents are rejected before anything is launched
    assert lib.sn_geneo_bank(None, None, 1, 9, 9, 9, None, None, None) == -1
    assert b"null" in lib.sn_last_error()
    assert lib.sn_conv_bank(None, 0, None, None, 1, 8, 8, 8, 1, 3, 3, 3, None, None, 0, None) == -1
    buf = ctypes.create_string_buffer(64)
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert lib.sn_conv_bank(p, 0, p, p, 1, 8, 8, 70, 4, 3, 3, 26, p, p, 0, None) == -2  # ky > 25
    assert b"ky=26" in lib.sn_last_error()
    assert lib.sn_conv_bank(p, 0, p, p, 1, 8, 8, 8, 4, 3, 3, 3, None, None, 0, None) == -1  # no output
    assert lib.sn_conv_bank(p, 0, p, p, 0, 8, 8, 8, 4, 3, 3, 3, p, p, 0, None) == -1  # B = 0
    assert lib.sn_voxel_scatter(p, None, p, 1, p, 4, 4, 4, p, p, None, 0, None, None) == -1  # towers w/o labels
    assert lib.sn_voxel_scatter(p, p, p, 1, p, 4, 4, 4, p, p, None, 3, None, None) == -2  # keep list missing
    # occupancy form: 512^3 bits do not fit the LDS bitmap even in 16 z-slabs -> caller must take the counting kernels
    assert lib.sn_voxel_occupancy(p, None, p, 1, p, 512, 512, 512, None, 0, p, p, None, 2, None, None, None, None,
                                  None) == -2
    assert b"LDS bitmap" in lib.sn_last_error()
    assert lib.sn_voxel_occupancy(p, None, p, 1, p, 64, 64, 64, None, 0, p, p, None, 1, None, None, None, None,
                                  None) == -1  # f64 output not offered
    assert lib.sn_voxel_prepare(p, p, 1, 64, 64, 0, 1, p, None, p, None) == -1
    assert lib.sn_voxel_finalize(p, None, 1, 4, 4, 4, p, None, p, None, None, None) == -1  # gt w/o towers
    assert lib.sn_grid_to_points(None, 0, 4, 4, 4, None, None, p, None) == -1
    assert lib.sn_grid_to_points(p, 0, 4, 0, 4, None, None, p, None) == -1   # empty grid
    assert b"empty" in lib.sn_last_error()
    assert lib.sn_grid_to_points(p, 0, 4, 4, 4, None, None, ctypes.c_void_p(p.value + 8), None) == -1  # out alignment


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_hip, "_lib", None)
    monkeypatch.setattr(_hip, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_hip.HipLibraryError, match="no CPU fallback"):
        _hip.load()
