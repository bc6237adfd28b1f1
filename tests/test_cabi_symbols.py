"""CPU-only: libmlpl_hip.so loads and exports every entry point include/mlpl_c.h declares (no compute calls)."""
import ctypes
import os
import re

import matchinglib_poselib_amd as mpa
from matchinglib_poselib_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    inc = os.path.join(ROOT, "include")
    for fn in sorted(os.listdir(inc)):
        if not fn.endswith(".h"):
            continue
        text = open(os.path.join(inc, fn)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(mlpl_[a-z0-9_]+)\s*\(", text))
    return names


def test_library_built_and_exports_all():
    assert os.path.exists(mpa.library_path()), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(mpa.library_path())
    decl = declared_symbols()
    assert len(decl) >= 20
    missing = [n for n in sorted(decl) if not hasattr(lib, n)]
    assert not missing, f"declared in include/*.h but not exported: {missing}"


def test_python_binding_covers_header():
    assert declared_symbols() == set(_lib._SIGNATURES), "ctypes table and mlpl_c.h disagree"
    mpa.load_library()


def test_no_device_is_loud():
    """Without a GPU the context must fail with MLPL_E_NO_DEVICE -- never fall back to the host."""
    lib = mpa.load_library()
    if lib.mlpl_device_count() > 0:
        return
    try:
        mpa.Context(0)
    except mpa.MlplError as e:
        assert e.code == _lib.MLPL_E_NO_DEVICE
    else:
        raise AssertionError("Context() succeeded without a device")


def test_dmatch_layout():
    assert mpa.DMATCH_DTYPE.itemsize == 16
    assert [mpa.DMATCH_DTYPE.fields[n][1] for n in ("queryIdx", "trainIdx", "imgIdx", "distance")] == [0, 4, 8, 12]
