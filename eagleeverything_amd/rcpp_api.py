"""Host-side mirror of the reference's exported C++ functions, over the C ABI of libeaglehip.so.

Same names, argument order and error behaviour as the functions registered in
E/src/RcppExports.cpp:154-170 (E/ = MyPackage/Eagle/ of the reference), so a parity test reads like a
call into the reference:

    ReadBlock(asciifname, start_row, numcols, numrows_in_block)                          RcppExports.cpp:9
    calculateMMt_rcpp(f_name_ascii, max_memory_in_Gbytes, num_cores, selected_loci, dims, quiet, message)   :37
    calculate_a_and_vara_rcpp(f_name_ascii, selected_loci, inv_MMt_sqrt, dim_reduced_vara,
                              max_memory_in_Gbytes, dims, a, quiet, message)             :54
    calculate_reduced_a_rcpp(f_name_ascii, varG, P, y, max_memory_in_Gbytes, dims, selected_loci, quiet, message)  :73
    extract_geno_rcpp(f_name_ascii, max_memory_in_Gbytes, selected_locus, dims)                                  :129

R matrices are column-major; numpy arrays are converted to Fortran order on the way in and come back so.
NA is numpy.nan.  Every call runs on the GPU; nothing here computes.
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import EagleError, c_dp, c_lp

_ctx = {}
_callbacks = {}


def context(device=0):
    """The eagle_ctx of `device` (opened on first use; fails loudly without a gfx950 device).  `device` may be a tuple of
    device numbers: ONE context that shards every call's markers over those GPUs (eagle_open_devices; the reference's unused
    AM(..., ngpu) hook, E/R/AM.R:185-196) -- pass the same tuple as `device=` to the functions below."""
    if device not in _ctx:
        L = _lib.load()
        if isinstance(device, tuple):
            arr = (C.c_int * len(device))(*[int(d) for d in device])
            h = L.eagle_open_devices(arr, len(device))
        else:
            h = L.eagle_open(int(device))
        if not h:
            raise EagleError(-6, L.eagle_open_error().decode())
        _ctx[device] = h
    return _ctx[device]


def close_all():
    L = _lib.load()
    for h in _ctx.values():
        L.eagle_close(h)
    _ctx.clear()
    _callbacks.clear()


def _check(ctx, rc, soft_ok=False):
    if rc < 0 or (rc > 0 and not soft_ok):
        raise EagleError(rc, _lib.load().eagle_last_error(ctx).decode())
    return rc


def _dp(a):
    return a.ctypes.data_as(c_dp)


def _f64F(a):
    return np.require(np.asarray(a, dtype=np.float64), requirements=["F", "ALIGNED"])


def _sel(selected_loci):
    s = np.atleast_1d(np.asarray(selected_loci, dtype=np.float64)).copy()
    return s, _dp(s), s.size


def _dims(dims):
    d = (C.c_long * 2)(int(dims[0]), int(dims[1]))
    return d


def _set_message(ctx, message):
    L = _lib.load()
    if message is None:
        L.eagle_set_message_callback(ctx, C.cast(None, _lib.MESSAGE_FN), None)
        _callbacks.pop(ctx, None)
    else:
        cb = _lib.MESSAGE_FN(lambda text, user: message(text.decode()))
        _callbacks[ctx] = cb  # keep alive
        L.eagle_set_message_callback(ctx, cb, None)


def device_info(device=0):
    L = _lib.load()
    ctx = context(device)
    arch = C.create_string_buffer(64)
    cu = C.c_int()
    hbm = C.c_int64()
    _check(ctx, L.eagle_device_info(ctx, arch, 64, C.byref(cu), C.byref(hbm)))
    return {"arch": arch.value.decode(), "cu_count": cu.value, "hbm_bytes": hbm.value}


def set_scan_mode(mode, device=0):
    """1 (default) = int8-slice vara kernel, 0 = fp64 MFMA vara kernel."""
    ctx = context(device)
    _check(ctx, _lib.load().eagle_set_scan_mode(ctx, int(mode)))


def set_scan_slices(nslices, device=0):
    """Base-256 digits of W used by the int8-slice kernel (1..8, default 7)."""
    ctx = context(device)
    _check(ctx, _lib.load().eagle_set_scan_slices(ctx, int(nslices)))


def set_scan_rounding(stochastic, device=0):
    """0 (default) = digits of W rounded to nearest (guaranteed bound); 1 = stochastic rounding (probabilistic certificate,
    failure probability < 1e-30 per marker, one digit fewer)."""
    ctx = context(device)
    _check(ctx, _lib.load().eagle_set_scan_rounding(ctx, int(stochastic)))


def set_scan_budget(relative_budget, device=0):
    """Relative digit budget of the int8 scan.  A context whose budget was never set tries 1e-7 first and falls back to 5e-7 (half of the
    path's 1e-6 tolerance); a value given here becomes THE budget, 0 restores that default policy.  The certificate sends every marker
    whose own bound exceeds 1.8 x the budget it enforces (last_scan_enforced) to the fp64 kernel."""
    ctx = context(device)
    _check(ctx, _lib.load().eagle_set_scan_budget(ctx, float(relative_budget)))


def last_scan_budget(device=0):
    """(budget in force for the last digit-slice scan, level of the spectral bound that took the digit off, error bound of an int8-made W)."""
    ctx = context(device)
    b, lv, we = C.c_double(), C.c_int(), C.c_double()
    _check(ctx, _lib.load().eagle_last_scan_budget(ctx, C.byref(b), C.byref(lv), C.byref(we)))
    return b.value, lv.value, we.value


def last_scan_enforced(device=0):
    """(budget the certificate of the last digit-slice scan enforced per marker, markers over the tight threshold): the budget in force, or
    the default behind a tight one when more than 512 markers of the whole scan missed the tight threshold (eagle_last_scan_enforced)."""
    ctx = context(device)
    b, nt = C.c_double(), C.c_long()
    _check(ctx, _lib.load().eagle_last_scan_enforced(ctx, C.byref(b), C.byref(nt)))
    return b.value, nt.value


def prepare_scan(n, L, device=0):
    """Start the allocation of the scan's device arena on a background thread (eagle_prepare_scan; calculateMMt_rcpp does it by itself)."""
    ctx = context(device)
    _check(ctx, _lib.load().eagle_prepare_scan(ctx, int(n), int(L)))


def set_w_mode(mode, device=0):
    """Which engine forms W = S (V S) for a digit-slice scan: 1 (default) = int8 digit slices from 4,096 padded individuals up, 0 = always
    the fp64 GEMM, 2 = int8 at any size (eagle_set_w_mode, csrc/eagle_w8.hip)."""
    ctx = context(device)
    _check(ctx, _lib.load().eagle_set_w_mode(ctx, int(mode)))


class _WInfo(C.Structure):
    _fields_ = [("int8", C.c_int), ("declined", C.c_int), ("k1", C.c_int), ("T1", C.c_int), ("pairs1", C.c_int), ("k2", C.c_int), ("T2", C.c_int),
                ("pairs2", C.c_int), ("eta", C.c_double), ("eta_x", C.c_double), ("target", C.c_double), ("mean_diag", C.c_double),
                ("asym_term", C.c_double), ("pipelined", C.c_int), ("pad", C.c_int)]


def last_w_info(device=0):
    """eagle_last_w_info as a dict: which engine formed the W of the last scan on this device, its configuration and error bound."""
    ctx = context(device)
    i = _WInfo()
    _check(ctx, _lib.load().eagle_last_w_info(ctx, C.byref(i)))
    return {k: getattr(i, k) for k, _ in _WInfo._fields_}


def drop_cache(device=0):
    _lib.load().eagle_drop_cache(context(device))


def ReadBlock(asciifname, start_row, numcols, numrows_in_block, device=0):
    L = _lib.load()
    ctx = context(device)
    out = np.zeros((int(numrows_in_block), int(numcols)), dtype=np.float64, order="F")
    _check(ctx, L.eagle_read_block(ctx, os.fsencode(asciifname), int(start_row), int(numcols), int(numrows_in_block),
                                   _dp(out)))
    return out


def calculateMMt_rcpp(f_name_ascii, max_memory_in_Gbytes, num_cores, selected_loci, dims, quiet=True, message=None,
                      device=0):
    L = _lib.load()
    ctx = context(device)
    _set_message(ctx, message)
    n = int(dims[0])
    s, sp, ns = _sel(selected_loci)
    out = np.zeros((n, n), dtype=np.float64, order="F")
    _check(ctx, L.eagle_calculateMMt(ctx, os.fsencode(f_name_ascii), float(max_memory_in_Gbytes), int(num_cores), sp, ns,
                                     _dims(dims), int(bool(quiet)), _dp(out)))
    return out


def calculate_a_and_vara_rcpp(f_name_ascii, selected_loci, inv_MMt_sqrt, dim_reduced_vara, max_memory_in_Gbytes, dims,
                              a, quiet=True, message=None, device=0):
    L = _lib.load()
    ctx = context(device)
    _set_message(ctx, message)
    Lm, n = int(dims[0]), int(dims[1])
    S = _f64F(inv_MMt_sqrt)
    V = _f64F(dim_reduced_vara)
    ah = _f64F(np.ravel(a))
    if S.shape != (n, n) or V.shape != (n, n) or ah.size != n:
        raise ValueError("inv_MMt_sqrt / dim_reduced_vara must be n x n and a of length n")
    s, sp, ns = _sel(selected_loci)
    a_out = np.zeros(Lm)
    v_out = np.zeros(Lm)
    rc = _check(ctx, L.eagle_calculate_a_and_vara(ctx, os.fsencode(f_name_ascii), sp, ns, _dp(S), _dp(V),
                                                  float(max_memory_in_Gbytes), _dims(dims), _dp(ah), int(bool(quiet)),
                                                  _dp(a_out), _dp(v_out)), soft_ok=True)
    if rc == 1:  # List(a = 0, vara = 0), calculate_a_and_vara_rcpp.cpp:141-142
        return {"a": np.zeros(1), "vara": np.zeros(1)}
    return {"a": a_out.reshape(Lm, 1), "vara": v_out.reshape(Lm, 1)}


def scan_with_W(f_name_ascii, selected_loci, W, v, max_memory_in_Gbytes, dims, quiet=True, message=None, device=0):
    """eagle_scan_with_W: the scan of calculate_a_and_vara_rcpp with W = S V S and v = S a_hat ready-made (inside AM():
    W = varG^2 P, v = varG P y; no n^3 product).  Not a symbol of the reference; an R-side shortcut."""
    L = _lib.load()
    ctx = context(device)
    _set_message(ctx, message)
    Lm, n = int(dims[0]), int(dims[1])
    Wm = _f64F(W)
    vv = _f64F(np.ravel(v))
    if Wm.shape != (n, n) or vv.size != n:
        raise ValueError("W must be n x n and v of length n")
    s, sp, ns = _sel(selected_loci)
    a_out = np.zeros(Lm)
    v_out = np.zeros(Lm)
    _check(ctx, L.eagle_scan_with_W(ctx, os.fsencode(f_name_ascii), sp, ns, _dp(Wm), _dp(vv), float(max_memory_in_Gbytes), _dims(dims),
                                    int(bool(quiet)), _dp(a_out), _dp(v_out)))
    return {"a": a_out.reshape(Lm, 1), "vara": v_out.reshape(Lm, 1)}


def spectral_prepare(f_name_ascii, dims, U, max_memory_in_Gbytes=8.0, device=0):
    """eagle_spectral_prepare: Z = Mt U once per AM() run (dims = (L, n) of Mt.ascii, U = eigenvectors of the normalised MM^T)."""
    L = _lib.load()
    ctx = context(device)
    Um = _f64F(U)
    n = int(dims[1])
    if Um.shape != (n, n):
        raise ValueError("U must be n x n")
    _check(ctx, L.eagle_spectral_prepare(ctx, os.fsencode(f_name_ascii), _dims(dims), _dp(Um), float(max_memory_in_Gbytes)))


def spectral_scan(lam, UtX, Uty, varE, varG, n_markers, selected_loci=np.nan, device=0):
    """eagle_spectral_scan: a and vara of every marker from one pass over Z (see include/eagle_hip.h section 1d)."""
    L = _lib.load()
    ctx = context(device)
    lam = _f64F(np.ravel(lam))
    UtX = _f64F(np.atleast_2d(UtX).reshape(lam.size, -1))
    Uty = _f64F(np.ravel(Uty))
    p = UtX.shape[1]
    s, sp, ns = _sel(selected_loci)
    a_out = np.zeros(int(n_markers))
    v_out = np.zeros(int(n_markers))
    _check(ctx, L.eagle_spectral_scan(ctx, _dp(lam), _dp(UtX), _dp(Uty), p, float(varE), float(varG), sp, ns, _dp(a_out), _dp(v_out)))
    return {"a": a_out.reshape(-1, 1), "vara": v_out.reshape(-1, 1)}


def calculate_reduced_a_rcpp(f_name_ascii, varG, P, y, max_memory_in_Gbytes, dims, selected_loci, quiet=True,
                             message=None, device=0):
    L = _lib.load()
    ctx = context(device)
    _set_message(ctx, message)
    n, Lm = int(dims[0]), int(dims[1])
    Pm = _f64F(P)
    yv = _f64F(np.ravel(y))
    s, sp, ns = _sel(selected_loci)
    out = np.zeros(Lm)
    rc = _check(ctx, L.eagle_calculate_reduced_a(ctx, os.fsencode(f_name_ascii), float(varG), _dp(Pm), _dp(yv),
                                                 float(max_memory_in_Gbytes), _dims(dims), sp, ns, int(bool(quiet)),
                                                 _dp(out)), soft_ok=True)
    if rc == 1:  # 1 x 1 zero matrix, calculate_reduced_a_rcpp.cpp:94-103
        return np.zeros((1, 1))
    return out.reshape(Lm, 1)


def extract_geno_rcpp(f_name_ascii, max_memory_in_Gbytes, selected_locus, dims, device=0):
    """E/src/extract_geno_rcpp.cpp:16-89: column selected_locus (0-based) of M.ascii as int32 -1/0/1."""
    L = _lib.load()
    ctx = context(device)
    n = int(dims[0])
    out = np.zeros(n, dtype=np.int32)
    _check(ctx, L.eagle_extract_geno(ctx, os.fsencode(f_name_ascii), float(max_memory_in_Gbytes), int(selected_locus),
                                     _dims(dims), out.ctypes.data_as(C.POINTER(C.c_int))))
    return out


def last_scan_argmax(device=0):
    """find_qtl.R:71-83 evaluated on the device on the last scan: (1-based index, tsq max, near ties)."""
    L = _lib.load()
    ctx = context(device)
    idx = C.c_long()
    mx = C.c_double()
    ties = C.c_long()
    _check(ctx, L.eagle_last_scan_argmax(ctx, C.byref(idx), C.byref(mx), C.byref(ties)))
    return idx.value, mx.value, ties.value


class _StreamStats(C.Structure):
    _fields_ = [("chunks", C.c_long), ("file_bytes", C.c_long), ("pread_s", C.c_double), ("load_s", C.c_double),
                ("wait_s", C.c_double), ("kernel_s", C.c_double), ("wall_s", C.c_double), ("load_first_s", C.c_double),
                ("starved_s", C.c_double)]


def last_stream_stats(device=0):
    """Out-of-core bookkeeping of the last call that streamed its file in marker chunks (include/eagle_hip.h,
    eagle_last_stream_stats), plus the derived storage rate and the fraction of the load time hidden under kernels."""
    L = _lib.load()
    ctx = context(device)
    st = _StreamStats()
    _check(ctx, L.eagle_last_stream_stats(ctx, C.byref(st)))
    d = {k: getattr(st, k) for k, _ in _StreamStats._fields_}
    d["read_GBps"] = d["file_bytes"] / 1e9 / d["pread_s"] if d["pread_s"] > 0 else 0.0
    later = d["load_s"] - d["load_first_s"]
    d["load_hidden_frac"] = max(0.0, 1.0 - d["starved_s"] / later) if later > 0 else 1.0
    return d


def last_scan_certificate(device=0):
    """(markers re-evaluated in fp64, of which flagged by their own error bound, whether a block fell back to fp64 entirely) of the
    last digit-slice calculate_a_and_vara_rcpp call (include/eagle_hip.h, eagle_last_scan_certificate)."""
    L = _lib.load()
    ctx = context(device)
    nre, nfl, fell = C.c_long(), C.c_long(), C.c_int()
    _check(ctx, L.eagle_last_scan_certificate(ctx, C.byref(nre), C.byref(nfl), C.byref(fell)))
    return nre.value, nfl.value, bool(fell.value)


def last_scan_digits(device=0):
    """(digit slices the last digit-slice scan used, slices cut from W, spectral bound or 0.0): eagle_last_scan_digits."""
    L = _lib.load()
    ctx = context(device)
    used, cut, H = C.c_int(), C.c_int(), C.c_double()
    _check(ctx, L.eagle_last_scan_digits(ctx, C.byref(used), C.byref(cut), C.byref(H)))
    return used.value, cut.value, H.value


class _ScanTiming(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("call_wall_s", "device_wall_s", "host_setup_s", "upload_ms", "w_ms", "load_wait_ms", "prepare_ms",
                                          "vara_ms", "certify_ms", "d2h_ms")] + [("blocks", C.c_long), ("markers", C.c_long)]


def last_scan_timing(device=0, device_index=0):
    """Phase clock of the last calculate_a_and_vara_rcpp / scan_with_W call on one device of the context (include/eagle_hip.h,
    eagle_last_scan_timing)."""
    L = _lib.load()
    ctx = context(device)
    st = _ScanTiming()
    _check(ctx, L.eagle_last_scan_timing(ctx, int(device_index), C.byref(st)))
    return {k: getattr(st, k) for k, _ in _ScanTiming._fields_}


def scan_operand_cache_stats(device=0):
    """(hits, misses) of the device copy of S = inv_MMt_sqrt kept between calculate_a_and_vara_rcpp calls (include/eagle_hip.h)."""
    L = _lib.load()
    ctx = context(device)
    h, m = C.c_long(), C.c_long()
    _check(ctx, L.eagle_scan_operand_cache_stats(ctx, C.byref(h), C.byref(m)))
    return h.value, m.value


def last_mmt_normalised(n, device=0):
    """calcMMt.R:13 applied on the device to the last calculateMMt result."""
    L = _lib.load()
    ctx = context(device)
    out = np.zeros((n, n), dtype=np.float64, order="F")
    mx = C.c_double()
    _check(ctx, L.eagle_last_mmt_normalised(ctx, _dp(out), C.byref(mx)))
    return out, mx.value


# ---- marker-file ingestion (E/src/RcppExports.cpp:92-151): same argument order as the reference's exports ----
def getRowColumn(fname, device=0):
    L = _lib.load()
    ctx = context(device)
    d = (C.c_long * 2)()
    _check(ctx, L.eagle_get_row_column(ctx, os.fsencode(fname), d))
    return [int(d[0]), int(d[1])]


def createM_ASCII_rcpp(f_name, f_name_ascii, type, AA, AB, BB, max_memory_in_Gbytes, dims, quiet=True, message=None,
                       missing="NA", device=0):
    """-> bool it_worked (createM_ASCII_rcpp.cpp:18-106).  last_error() describes a False."""
    L = _lib.load()
    ctx = context(device)
    _set_message(ctx, message)
    enc = lambda v: str(v).encode()  # R passes AA=0, AB=1, BB=2 through as.character
    rc = _check(ctx, L.eagle_create_M_ascii(ctx, os.fsencode(f_name), os.fsencode(f_name_ascii), enc(type), enc(AA), enc(AB),
                                            enc(BB), float(max_memory_in_Gbytes), _dims(dims), int(bool(quiet)), enc(missing)),
                soft_ok=True)
    return rc == 0


def createMt_ASCII_rcpp(f_name, f_name_ascii, type, max_memory_in_Gbytes, dims, quiet=True, message=None, device=0):
    L = _lib.load()
    ctx = context(device)
    _set_message(ctx, message)
    _check(ctx, L.eagle_create_Mt_ascii(ctx, os.fsencode(f_name), os.fsencode(f_name_ascii), str(type).encode(),
                                        float(max_memory_in_Gbytes), _dims(dims), int(bool(quiet))))


# ---- the dense model algebra on the device, through the C ABI (opt-in; include/eagle_hip.h section 1c) ----
def _square_any_order(A):
    """(buffer, transposed): a float64 n x n array usable as a column-major matrix without a copy when it is contiguous in
    either order -- a C-ordered buffer read column-major is the transpose, which the callers below undo for free
    (symmetric input, or inv(A^T) = inv(A)^T).  A 200 MB layout change on one host core costs more than the device call.
    Triangles: eagle_sym_eig reads the LOWER and eagle_chol2inv the UPPER triangle of the column-major matrix (as R's eigen() and
    chol() do); a C-ordered array goes over as its transpose, so the OTHER triangle is read -- identical for an exactly symmetric
    matrix, different at rounding level for one that is symmetric only to rounding (pass np.asfortranarray(A) to pin R's triangle)."""
    A = np.asarray(A, dtype=np.float64)
    if A.ndim != 2 or A.shape[0] != A.shape[1]:
        raise ValueError("square matrix expected")
    if A.flags.f_contiguous and A.flags.aligned:
        return A, False
    if A.flags.c_contiguous and A.flags.aligned:
        return A, True
    return _f64F(A), False


def sym_eig(A, only_values=False, device=0):
    """eigen(A, symmetric=TRUE): (values in decreasing order, vectors in columns) like R."""
    L = _lib.load()
    ctx = context(device)
    A, _ = _square_any_order(A)   # symmetric: the transpose is the same matrix
    n = A.shape[0]
    w = np.empty(n)
    U = None if only_values else np.empty((n, n), order="F")
    _check(ctx, L.eagle_sym_eig(ctx, _dp(A), n, _dp(w), None if only_values else _dp(U)))
    return w, U


def chol2inv(A, device=0):
    """chol2inv(chol(A)); raises EagleError(1, R's chol() message) when A is not positive definite."""
    L = _lib.load()
    ctx = context(device)
    A, tr = _square_any_order(A)  # symmetric in, symmetric out: returned in the caller's order
    n = A.shape[0]
    out = np.empty((n, n), order="C" if tr else "F")
    _check(ctx, L.eagle_chol2inv(ctx, _dp(A), n, _dp(out)))
    return out


def inverse(A, device=0):
    """solve(A)."""
    L = _lib.load()
    ctx = context(device)
    A, tr = _square_any_order(A)  # a C-ordered A is handed over as A^T; inv(A^T) read back row-major is inv(A)
    n = A.shape[0]
    out = np.empty((n, n), order="C" if tr else "F")
    _check(ctx, L.eagle_inverse(ctx, _dp(A), n, _dp(out)))
    return out


def matmul(A, B, device=0):
    """A %*% B on the library's fp64 MFMA GEMM."""
    L = _lib.load()
    ctx = context(device)
    A, B = np.atleast_2d(np.asarray(A, dtype=np.float64)), np.atleast_2d(np.asarray(B, dtype=np.float64))
    m, k = A.shape
    k2, n = B.shape
    if k != k2:
        raise ValueError("non-conformable arguments")
    if A.flags.c_contiguous and B.flags.c_contiguous and A.flags.aligned and B.flags.aligned and not (A.flags.f_contiguous and B.flags.f_contiguous):
        # row-major operands: their buffers read column-major are A^T (k x m) and B^T (n x k); B^T A^T = (A B)^T, whose
        # column-major image is A B row-major -- no layout change on the host
        out = np.empty((m, n), order="C")
        _check(ctx, L.eagle_matmul(ctx, _dp(B), _dp(A), n, k, m, _dp(out)))
        return out
    A, B = _f64F(A), _f64F(B)
    out = np.empty((m, n), order="F")
    _check(ctx, L.eagle_matmul(ctx, _dp(A), _dp(B), m, k, n, _dp(out)))
    return out


def mmt_sqrt_and_sqrtinv(MMt, device=0):
    """E/R/calculateMMt_sqrt_and_sqrtinv.R:15-47 -> (sqrt, invsqrt, trace of their product), or None where the R function
    returns NULL (MMt not positive definite)."""
    L = _lib.load()
    ctx = context(device)
    M = _f64F(MMt)
    n = M.shape[0]
    sq, inv = np.zeros((n, n), order="F"), np.zeros((n, n), order="F")
    tr = C.c_double()
    rc = _check(ctx, L.eagle_mmt_sqrt_and_sqrtinv(ctx, _dp(M), n, _dp(sq), _dp(inv), C.byref(tr)), soft_ok=True)
    if rc == 1:
        return None
    return sq, inv, tr.value


def last_error(device=0):
    return _lib.load().eagle_last_error(context(device)).decode()
