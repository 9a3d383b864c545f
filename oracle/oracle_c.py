"""ctypes binding of oracle/libeagle_oracle.so (the plain-C restatement).  TEST INFRASTRUCTURE ONLY.

Mirrors the argument order of the reference's exported C++ functions
(E/src/RcppExports.cpp:9,37,54,73) so parity tests read like calls into the reference.
parity unpinned: see the header of eagle_oracle.c.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libeagle_oracle.so")
_lib = None

c_dp = C.POINTER(C.c_double)
c_lp = C.POINTER(C.c_long)


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("eagle_oracle.c", "eagle_oracle_ingest.c")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libeagle_oracle.so"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.eo_last_error.restype = C.c_char_p
        L.eo_read_block.argtypes = [C.c_char_p, C.c_long, C.c_long, C.c_long, c_dp]
        L.eo_calculateMMt.argtypes = [C.c_char_p, C.c_double, C.c_int, c_dp, C.c_long, C.c_long, C.c_long, c_dp, c_lp]
        L.eo_mmt_from_i8.argtypes = [C.c_void_p, C.c_long, C.c_long, C.c_long, c_dp]
        L.eo_normalise_MMt.argtypes = [c_dp, C.c_long]
        L.eo_normalise_MMt.restype = None
        L.eo_calculate_a_and_vara.argtypes = [C.c_char_p, c_dp, C.c_long, c_dp, c_dp, C.c_double, C.c_long, C.c_long,
                                              c_dp, c_dp, c_dp, c_lp]
        L.eo_scan_from_i8.argtypes = [C.c_void_p, C.c_long, C.c_long, C.c_long, c_dp, c_dp, c_dp, c_dp, c_dp]
        L.eo_scan_from_i8_with_W.argtypes = [C.c_void_p, C.c_long, C.c_long, C.c_long, c_dp, c_dp, c_dp, c_dp]
        L.eo_scan_operands_pub.argtypes = [c_dp, c_dp, c_dp, C.c_long, c_dp, c_dp]
        L.eo_calculate_reduced_a.argtypes = [C.c_char_p, C.c_double, c_dp, c_dp, C.c_double, C.c_long, C.c_long, c_dp,
                                             C.c_long, c_dp]
        L.eo_tsq_argmax.argtypes = [c_dp, c_dp, C.c_long, c_dp, c_dp]
        L.eo_tsq_argmax.restype = C.c_long
        L.eo_num_threads.restype = C.c_int
        L.eo_extract_geno.argtypes = [C.c_char_p, C.c_long, C.c_long, C.c_long, C.POINTER(C.c_int)]
        L.eo_set_num_threads.argtypes = [C.c_int]
        L.eo_set_num_threads.restype = None
        L.eo_getRowColumn.argtypes = [C.c_char_p, c_lp]
        L.eo_create_ascii_text.argtypes = [C.c_char_p, C.c_char_p, C.c_long, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p,
                                           c_lp, c_lp, C.c_char_p, C.c_long]
        L.eo_create_ascii_plink.argtypes = [C.c_char_p, C.c_char_p, C.c_long, c_lp, c_lp, c_lp, C.POINTER(C.c_int)]
        L.eo_createMt_ascii.argtypes = [C.c_char_p, C.c_char_p, C.c_long, C.c_long]
        _lib = L
    return _lib


class OracleError(RuntimeError):
    pass


def _dp(a):
    return a.ctypes.data_as(c_dp)


def _f64(a, order="F"):
    return np.require(np.asarray(a, dtype=np.float64), requirements=["ALIGNED", "WRITEABLE", "F" if order == "F" else "C"])


def _sel(selected_loci):
    s = np.atleast_1d(np.asarray(selected_loci, dtype=np.float64)).copy()
    return s, _dp(s), s.size


def _check(rc, soft_ok=False):
    if rc < 0 or (rc > 0 and not soft_ok):
        raise OracleError(lib().eo_last_error().decode())
    return rc


def ReadBlock(asciifname, start_row, numcols, numrows_in_block):
    """E/src/ReadBlock.cpp:16-68 -> (numrows x numcols) float64, values in {-1,0,1}."""
    out = np.zeros((numrows_in_block, numcols), dtype=np.float64, order="F")
    _check(lib().eo_read_block(os.fsencode(asciifname), start_row, numcols, numrows_in_block, _dp(out)))
    return out


def calculateMMt_rcpp(f_name_ascii, max_memory_in_Gbytes, num_cores, selected_loci, dims, quiet=True, message=None,
                      return_branch=False):
    """E/src/calculateMMt_rcpp.cpp:19-185; dims=(n, L)."""
    n, L = int(dims[0]), int(dims[1])
    s, sp, ns = _sel(selected_loci)
    out = np.zeros((n, n), dtype=np.float64, order="F")
    br = C.c_long(-1)
    _check(lib().eo_calculateMMt(os.fsencode(f_name_ascii), float(max_memory_in_Gbytes), int(num_cores), sp, ns, n, L,
                                 _dp(out), C.byref(br)))
    return (out, br.value) if return_branch else out


def normalise_MMt(MMt):
    """E/R/calcMMt.R:13"""
    out = _f64(MMt).copy(order="F")
    lib().eo_normalise_MMt(_dp(out), out.shape[0])
    return out


def calculate_a_and_vara_rcpp(f_name_ascii, selected_loci, inv_MMt_sqrt, dim_reduced_vara, max_memory_in_Gbytes, dims,
                              a, quiet=True, message=None, return_branch=False):
    """E/src/calculate_a_and_vara_rcpp.cpp:22-241; dims=(L, n) of Mt.  Returns dict(a=, vara=)."""
    L, n = int(dims[0]), int(dims[1])
    S = _f64(inv_MMt_sqrt)
    V = _f64(dim_reduced_vara)
    ah = _f64(np.ravel(a))
    s, sp, ns = _sel(selected_loci)
    a_out = np.zeros(L)
    v_out = np.zeros(L)
    br = C.c_long(-1)
    rc = _check(lib().eo_calculate_a_and_vara(os.fsencode(f_name_ascii), sp, ns, _dp(S), _dp(V),
                                              float(max_memory_in_Gbytes), L, n, _dp(ah), _dp(a_out), _dp(v_out),
                                              C.byref(br)), soft_ok=True)
    if rc == 1:  # sentinel List(a=0, vara=0), calculate_a_and_vara_rcpp.cpp:141-142
        res = {"a": np.zeros(1), "vara": np.zeros(1)}
    else:
        res = {"a": a_out.reshape(L, 1), "vara": v_out.reshape(L, 1)}
    return (res, br.value) if return_branch else res


def calculate_reduced_a_rcpp(f_name_ascii, varG, P, y, max_memory_in_Gbytes, dims, selected_loci, quiet=True,
                             message=None):
    """E/src/calculate_reduced_a_rcpp.cpp:20-171; dims=(n, L) of M, file is Mt.ascii."""
    n, L = int(dims[0]), int(dims[1])
    Pm = _f64(P)
    yv = _f64(np.ravel(y))
    s, sp, ns = _sel(selected_loci)
    out = np.zeros(L)
    rc = _check(lib().eo_calculate_reduced_a(os.fsencode(f_name_ascii), float(varG), _dp(Pm), _dp(yv),
                                             float(max_memory_in_Gbytes), n, L, sp, ns, _dp(out)), soft_ok=True)
    if rc == 1:
        return np.zeros((1, 1))
    return out.reshape(L, 1)


def extract_geno_rcpp(f_name_ascii, max_memory_in_Gbytes, selected_locus, dims):
    """E/src/extract_geno_rcpp.cpp:16-89."""
    n, L = int(dims[0]), int(dims[1])
    out = np.zeros(n, dtype=np.int32)
    _check(lib().eo_extract_geno(os.fsencode(f_name_ascii), int(selected_locus), n, L, out.ctypes.data_as(C.POINTER(C.c_int))))
    return out


def tsq_argmax(a, vara):
    """E/R/find_qtl.R:71-83 -> (tsq, 1-based index, max)."""
    a = _f64(np.ravel(a))
    v = _f64(np.ravel(vara))
    tsq = np.zeros(a.size)
    mx = C.c_double()
    idx = lib().eo_tsq_argmax(_dp(a), _dp(v), a.size, _dp(tsq), C.byref(mx))
    return tsq, int(idx), mx.value


def mmt_from_i8(M8):
    """In-memory branch (calculateMMt_rcpp.cpp:84-95) on an int8 (n x L) matrix -- CPU-baseline leg."""
    M8 = np.ascontiguousarray(M8, dtype=np.int8)
    n, L = M8.shape
    out = np.zeros((n, n), dtype=np.float64, order="F")
    _check(lib().eo_mmt_from_i8(M8.ctypes.data, n, L, L, _dp(out)))
    return out


def scan_from_i8(Mt8, S, V, ahat):
    """In-memory branch (calculate_a_and_vara_rcpp.cpp:74-112) on an int8 (L x n) matrix."""
    Mt8 = np.ascontiguousarray(Mt8, dtype=np.int8)
    L, n = Mt8.shape
    S = _f64(S); V = _f64(V); ah = _f64(np.ravel(ahat))
    a = np.zeros(L); v = np.zeros(L)
    _check(lib().eo_scan_from_i8(Mt8.ctypes.data, L, n, n, _dp(S), _dp(V), _dp(ah), _dp(a), _dp(v)))
    return a, v


def scan_operands(S, V, ahat):
    """v = S*ahat, W = S*(V*S) (row-major) -- calculate_a_and_vara_rcpp.cpp:90,97-98."""
    S = _f64(S); V = _f64(V); ah = _f64(np.ravel(ahat))
    n = S.shape[0]
    v = np.zeros(n); W = np.zeros((n, n), order="C")
    _check(lib().eo_scan_operands_pub(_dp(S), _dp(V), _dp(ah), n, _dp(v), _dp(W)))
    return v, W


def scan_from_i8_with_W(Mt8, v, W_rm):
    Mt8 = np.ascontiguousarray(Mt8, dtype=np.int8)
    L, n = Mt8.shape
    v = _f64(v); W = np.require(np.asarray(W_rm, dtype=np.float64), requirements=["C", "ALIGNED"])
    a = np.zeros(L); va = np.zeros(L)
    _check(lib().eo_scan_from_i8_with_W(Mt8.ctypes.data, L, n, n, _dp(v), _dp(W), _dp(a), _dp(va)))
    return a, va


def num_threads():
    return int(lib().eo_num_threads())


def set_num_threads(t):
    lib().eo_set_num_threads(int(t))


# ---- marker-file ingestion (eagle_oracle_ingest.c); argument order of E/src/RcppExports.cpp:92,113,143 ----
def getRowColumn(fname):
    d = (C.c_long * 2)()
    rc = lib().eo_getRowColumn(os.fsencode(fname), d)
    if rc:
        raise OracleError("ERROR: Could not open  %s" % fname)
    return [int(d[0]), int(d[1])]


def createM_ASCII_rcpp(f_name, f_name_ascii, type, AA, AB, BB, max_memory_in_Gbytes, dims, quiet=True, message=None,
                       missing="NA"):
    """Returns (it_worked, info): info = dict(kind, row, token/columns/locus, missing_seen)."""
    er, ec, el = C.c_long(0), C.c_long(0), C.c_long(0)
    info = {}
    if type == "PLINK":
        ms = C.c_int(0)
        rc = lib().eo_create_ascii_plink(os.fsencode(f_name), os.fsencode(f_name_ascii), int(dims[1]), C.byref(er), C.byref(el),
                                         C.byref(ec), C.byref(ms))
        info["missing_seen"] = bool(ms.value)
    else:
        tok = C.create_string_buffer(64)
        rc = lib().eo_create_ascii_text(os.fsencode(f_name), os.fsencode(f_name_ascii), int(dims[1]), str(AA).encode(),
                                        str(AB).encode(), str(BB).encode(), str(missing).encode(), C.byref(er), C.byref(ec), tok, 64)
        info["token"] = tok.value.decode()
    if rc < 0:
        return False, {"kind": "open"}
    info.update(kind={0: "ok", 1: "token", 2: "columns", 3: "alleles"}[rc], row=er.value, columns=ec.value, locus=el.value)
    return rc == 0, info


def createMt_ASCII_rcpp(f_name, f_name_ascii, type, max_memory_in_Gbytes, dims, quiet=True, message=None):
    rc = lib().eo_createMt_ascii(os.fsencode(f_name), os.fsencode(f_name_ascii), int(dims[0]), int(dims[1]))
    if rc:
        raise OracleError("createMt_ASCII_rcpp failed (%d)" % rc)
