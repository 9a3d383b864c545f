"""CPU: the C-ABI library loads and exports every symbol include/eagle_hip.h declares; without a GPU it
refuses to open (no CPU fallback).  No compute calls here."""
import os
import re
import subprocess

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


def _build_c_demo(tmp_path):
    import subprocess
    exe = str(tmp_path / "eagle_scan_demo")
    libdir = os.path.join(ROOT, "eagleeverything_amd")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "eagle_scan_demo.c"), "-L" + libdir, "-leaglehip",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-lm", "-o", exe])
    return exe


def test_c_example_builds_against_the_abi(tmp_path):
    """examples/eagle_scan_demo.c is plain C: it compiles against include/eagle_hip.h and links libeaglehip.so."""
    _lib.load()  # builds the library if needed
    assert os.path.exists(_build_c_demo(tmp_path))


@pytest.mark.gpu
def test_c_example_runs(tmp_path):
    import subprocess
    exe = _build_c_demo(tmp_path)
    out = subprocess.check_output([exe, os.path.join(ROOT, "tests", "golden", "geno_150x100.txt"), "0", "1", "2", str(tmp_path)],
                                  stderr=subprocess.DEVNULL, text=True)
    assert "n=150 L=100 trace(MMt)=9748 max(MMt)=89 " in out and "closed_form_mismatches=0" in out  # SURVEY section 4 known answers


def test_shims_typecheck():
    """The eight Rcpp-typed translation units of eagleeverything_amd/shim/ (what a maintainer drops into the R package's
    src/) against include/eagle_hip.h: every argument passed to the C ABI has the declared type.  No R in
    this container, so Rcpp / Eigen are tests/stubs/Rcpp.h (declarations only, labelled test infrastructure); the exported
    prototypes are the reference's (E/src/RcppExports.cpp:9-151)."""
    import glob
    shims = sorted(glob.glob(os.path.join(ROOT, "eagleeverything_amd", "shim", "*.cpp")))
    assert len(shims) == 8
    for f in shims:
        r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "tests", "stubs"),
                            "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "eagleeverything_amd", "shim"), f],
                           capture_output=True, text=True)
        assert r.returncode == 0, f + "\n" + r.stderr
