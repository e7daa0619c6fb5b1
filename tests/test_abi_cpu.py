"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/tlsan.h declares; pure host entry points behave; the product refuses to run without
a GPU instead of falling back to anything."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
    from tlsan_amd import _lib as L
    if not os.path.exists(L.LIB_PATH):
        from tlsan_amd.build import build
        build()
    return L, L.load()


def test_header_symbols_exported():
    L, lib = _lib()
    hdr = open(os.path.join(ROOT, "include", "tlsan.h")).read()
    declared = set(re.findall(r"\b(tlsan_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(L.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.tlsan_abi_version() == L.ABI_VERSION == 14


def test_dense_layout_and_sizes():
    L, lib = _lib()
    for d, n in ((64, 4449), (128, 17601), (256, 70017)):
        dims = L.Dims(100, 200, 10, d, d // 2, d // 2, 8, 10)
        lay = L.DenseLayout()
        assert lib.tlsan_dense_layout_of(C.byref(dims), C.byref(lay)) == 0
        assert lay.n_dense == n
        assert lib.tlsan_state_bytes(C.byref(dims)) > 0
        w1 = lib.tlsan_workspace_bytes(C.byref(dims), 32, 4)
        w2 = lib.tlsan_workspace_bytes(C.byref(dims), 4096, 18)
        assert 0 < w1 < w2
    bad = L.Dims(100, 200, 10, 96, 48, 48, 8, 10)
    assert lib.tlsan_state_bytes(C.byref(bad)) == 0
    assert b"unsupported" in lib.tlsan_last_error()
    ok90 = L.Dims(100, 200, 10, 128, 64, 64, 8, 90)   # reference max_length: streamed long block
    assert lib.tlsan_workspace_bytes(C.byref(ok90), 32, 4) > 0
    bad = L.Dims(100, 200, 10, 128, 64, 64, 8, 97)
    assert lib.tlsan_workspace_bytes(C.byref(bad), 32, 4) == 0


def test_null_arguments_are_rejected_not_crashed():
    L, lib = _lib()
    dims = L.Dims(100, 200, 10, 128, 64, 64, 8, 10)
    assert lib.tlsan_forward(C.byref(dims), None, None, None, None, None, None, 0, None) == -1
    assert lib.tlsan_train_step(C.byref(dims), None, None, None, None, None, None, 0, None) == -1


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "tlsan_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f


def test_model_needs_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from tlsan_amd.model import Model
    from tests.helpers import make_config
    import numpy as np
    cfg = make_config()
    with pytest.raises(RuntimeError):
        Model(cfg, np.zeros(cfg["item_count"], np.int32))
