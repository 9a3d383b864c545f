"""CPU: the C-ABI library loads and exports every symbol include/eagle_hip.h declares; without a GPU it
refuses to open (no CPU fallback).  No compute calls here."""
import os
import re

import pytest

from conftest import ROOT
from eagleeverything_amd import _lib


def _declared():
    txt = open(os.path.join(ROOT, "include", "eagle_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(eagle_[a-zA-Z0-9_]+)\s*\(", txt)) - {"eagle_message_fn"})


def test_header_symbols_exported_and_bound():
    L = _lib.load()
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), "libeaglehip.so does not export " + n
        assert n in _lib.SIGNATURES, "no ctypes signature for " + n


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    L = _lib.load()
    assert not L.eagle_open(0)
    assert b"no CPU fallback" in L.eagle_open_error()
    from eagleeverything_amd import rcpp_api
    with pytest.raises(_lib.EagleError):
        rcpp_api.ReadBlock("/nonexistent", 0, 1, 1)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "eagleeverything_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".cpp", ".hip", ".h", ".c")):
                src = open(os.path.join(dp, fn)).read()
                assert "oracle" not in src.replace("oracle's", ""), "product file %s mentions the oracle" % fn
