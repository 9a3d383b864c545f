"""Independent numpy restatement of the hot path (OpenBLAS fp64 / exact int64).  TEST INFRASTRUCTURE ONLY.

Used in the build container to cross-check the C oracle and to generate tests/golden fixtures.
parity unpinned: the reference cannot be run (no R / Rcpp / Eigen); see oracle/eagle_oracle.c.
Reference lines: E/src/ReadBlock.cpp:47-58, calculateMMt_rcpp.cpp:84-95, calculate_a_and_vara_rcpp.cpp:76-112,
calculate_reduced_a_rcpp.cpp:65-85, E/R/calcMMt.R:13, E/R/find_qtl.R:71-83.
"""
import numpy as np


def read_ascii(path, start_row, numcols, numrows):
    """ReadBlock.cpp:47-58 -> int8 (numrows x numcols) in {-1,0,1}."""
    rows = []
    with open(path, "rb") as f:
        for rr, line in enumerate(f):
            if rr >= start_row + numrows:
                break
            if rr >= start_row:
                b = np.frombuffer(line.rstrip(b"\r\n"), dtype=np.uint8)[:numcols]
                if b.size < numcols:
                    raise ValueError("short line")
                rows.append(b.astype(np.int16) - ord("0") - 1)
    if len(rows) != numrows:
        raise ValueError("short file")
    return np.stack(rows).astype(np.int8) if rows else np.zeros((0, numcols), np.int8)


def _mask_active(sel):
    sel = np.atleast_1d(np.asarray(sel, dtype=np.float64))
    return sel.size > 0 and not np.isnan(sel[0])


def mmt_int64(M8, selected_loci=np.nan):
    """calculateMMt_rcpp.cpp:88-95 in exact integer arithmetic. M8: (n x L) int8."""
    G = np.asarray(M8, dtype=np.int64).copy()
    if _mask_active(selected_loci):
        G[:, np.asarray(selected_loci, dtype=np.int64)] = 0
    # int64 matmul is slow in numpy; float64 is exact while |sum| < 2^53
    Gf = G.astype(np.float64)
    out = Gf @ Gf.T
    return np.rint(out).astype(np.int64)


def normalise(MMt):
    MMt = np.asarray(MMt, dtype=np.float64)
    return MMt / MMt.max() + np.diag(np.full(MMt.shape[0], 0.95))


def a_and_vara(Mt8, S, V, ahat, selected_loci=np.nan):
    """calculate_a_and_vara_rcpp.cpp:76-112. Mt8: (L x n) int8; S, V (n x n); ahat (n,)."""
    Mt = np.asarray(Mt8, dtype=np.float64).copy()
    if _mask_active(selected_loci):
        Mt[np.asarray(selected_loci, dtype=np.int64), :] = 0.0
    v = S @ np.ravel(ahat)
    a = Mt @ v
    W = S @ (V @ S)
    T = Mt @ W
    vara = np.einsum("ij,ij->i", T, Mt)
    return a, vara


def reduced_a(Mt8, varG, P, y, selected_loci=np.nan):
    Mt = np.asarray(Mt8, dtype=np.float64).copy()
    if _mask_active(selected_loci):
        Mt[np.asarray(selected_loci, dtype=np.int64), :] = 0.0
    return varG * (Mt @ (P @ np.ravel(y)))


def tsq_argmax(a, vara):
    with np.errstate(divide="ignore", invalid="ignore"):
        tsq = np.ravel(a) ** 2 / np.ravel(vara)
    if np.all(np.isnan(tsq)):
        return tsq, 0, np.nan
    mx = np.nanmax(tsq)
    return tsq, int(np.flatnonzero(tsq == mx)[0]) + 1, mx
