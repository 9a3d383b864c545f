"""CPU sanitizer coverage of the host side (-fsanitize=address,undefined on the CPU build; GPU ASan / XNACK do
not exist on the target pool).

  * tests/host/test_host.cpp exercises the HIP-free host pieces of the product (csrc/eagle_host.h: Rendezvous, split_markers,
    parse_selected_core, stream_chunk_rows_core, vara_tail_pieces, LineIndex + tokeniser) -- the very header libeaglehip.so
    compiles -- under ASan + UBSan, and the Rendezvous cases under TSan.  The failure-injection case is ADVICE r2's deadlock:
    a device that fails right after leaving round k while a slow peer has not yet woken up from round k.
  * the C oracle (test infrastructure) is built with `make -C oracle asan` and run on a golden case in a child process.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

SRC = os.path.join(ROOT, "tests", "host", "test_host.cpp")


def _build(tmp_path, name, flags):
    exe = str(tmp_path / name)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-pthread"] + flags + [SRC, "-o", exe])
    return exe


def test_host_pieces_under_asan_ubsan(tmp_path):
    exe = _build(tmp_path, "host_asan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=all"])
    r = subprocess.run([exe, "all"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "host checks passed" in r.stdout


def test_rendezvous_under_tsan(tmp_path):
    exe = _build(tmp_path, "host_tsan", ["-fsanitize=thread"])
    r = subprocess.run([exe, "rendezvous"], capture_output=True, text=True, timeout=300)  # a hang (the old bug) is a timeout here
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ThreadSanitizer" not in r.stderr


def test_oracle_under_asan_ubsan(tmp_path):
    """`make -C oracle asan` + one golden case through every oracle entry point the parity tests use, in a child process with the
    sanitizer runtime preloaded (python itself is not instrumented; leak checking off for the interpreter's sake)."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"])
    so = os.path.join(ROOT, "oracle", "libeagle_oracle_asan.so")
    asan_rt = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    ubsan_rt = subprocess.check_output(["gcc", "-print-file-name=libubsan.so"], text=True).strip()
    code = r'''
import os, sys, numpy as np
sys.path.insert(0, %r)
from oracle import oracle_c
oracle_c._SO = %r
oracle_c._lib = None
g = dict(np.load(os.path.join(%r, "synth_203x1531.npz")))
M8 = g["M8"]
n, L = M8.shape
from eagleeverything_amd import synth
d = %r
geno = synth.write_geno_pair(d, np.ascontiguousarray(M8.T))
blk = oracle_c.ReadBlock(geno["asciifileM"], 3, L, 7)
assert np.array_equal(blk, M8[3:10].astype(np.float64))
mmt = oracle_c.calculateMMt_rcpp(geno["asciifileM"], 8.0, 2, np.nan, (n, L))
assert np.array_equal(mmt, g["MMt"].astype(np.float64))
mmt_b = oracle_c.calculateMMt_rcpp(geno["asciifileM"], 0.0021, 2, np.nan, (n, L))   # the blocked branch
assert np.array_equal(mmt_b, mmt)
r = oracle_c.calculate_a_and_vara_rcpp(geno["asciifileMt"], np.nan, g["S"], g["V"], 8.0, (L, n), g["ahat"])
np.testing.assert_allclose(r["a"].ravel(), g["a"], rtol=1e-10)
np.testing.assert_allclose(r["vara"].ravel(), g["vara"], rtol=1e-9)
r2 = oracle_c.calculate_a_and_vara_rcpp(geno["asciifileMt"], np.array([5.0, 7.0]), g["S"], g["V"], 0.003, (L, n), g["ahat"])  # blocked + masked
assert r2["a"].ravel()[5] == 0.0 and r2["vara"].ravel()[7] == 0.0
print("oracle asan ok")
''' % (ROOT, so, GOLDEN, str(tmp_path))
    env = dict(os.environ, LD_PRELOAD=asan_rt + ":" + ubsan_rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=23",
               UBSAN_OPTIONS="halt_on_error=1:exitcode=24", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "oracle asan ok" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
