"""Dense n x n model algebra that feeds the marker scan.

north_star keeps calculateH / calculateP / the eigendecomposition of MM^T on host LAPACK; in the R
package these are base-R calls.  This numpy restatement exists so that the scan can be driven with
realistic operands (S = MMt^-1/2, V = Var(a_hat), a_hat) on a box without R.  It is *not* part of
the GPU hot path and is not a fallback for it.  Host LAPACK is the default; set_algebra("device") moves the
O(n^3) primitives (eigh, chol2inv, inv, n x n products) to the GPU through the C ABI (eagle_sym_eig & co., include/eagle_hip.h
section 1c) -- SURVEY 8 f-4, the Amdahl term of a full AM() run once the scan takes 30 ms -- without changing a formula.

Reference lines (E/ = MyPackage/Eagle/):
  calculateH ........................ E/R/calculateH.R:36
  calculateP ........................ E/R/calculateP.R:27-28
  calculateMMt_sqrt_and_sqrtinv ..... E/R/calculateMMt_sqrt_and_sqrtinv.R:15-47
  calculate_reduced_a ............... E/R/calculate_reduced_a.R:31
  calculate_reduced_vara ............ E/R/calculate_reduced_vara.R:21-35
No recorded outputs exist in the reference for any of these: parity unpinned.
"""
import numpy as np
import scipy.linalg as sla


class _HostLA:
    """The O(n^3) primitives of this module on host LAPACK / BLAS (the default, as north_star asks)."""
    name = "host"

    @staticmethod
    def eigh(A):
        return np.linalg.eigh(A)

    @staticmethod
    def eigh_desc(A):
        """R's eigen(): values decreasing (views of LAPACK's ascending result)."""
        w, U = np.linalg.eigh(A)
        return w[::-1], U[:, ::-1]

    @staticmethod
    def chol2inv(A):
        c, low = sla.cho_factor(A, lower=False, check_finite=False)
        return sla.cho_solve((c, low), np.eye(A.shape[0]), check_finite=False)

    @staticmethod
    def inv(A):
        return np.linalg.inv(A)

    @staticmethod
    def mm(A, B):
        return A @ B


class _DeviceLA:
    """the same primitives on the GPU through the C ABI of libeaglehip.so (include/eagle_hip.h section 1c:
    eagle_sym_eig / eagle_chol2inv / eagle_inverse on rocSOLVER, eagle_matmul on the library's own fp64 MFMA GEMM) -- what an
    R caller would bind too.  Opt-in (set_algebra("device")); no torch in this path.  numpy in, numpy out."""
    name = "device"

    def __init__(self, device=0):
        from . import rcpp_api
        self.api = rcpp_api
        self.device = device

    def eigh(self, A):
        w, U = self.api.sym_eig(A, device=self.device)   # R's order (decreasing); numpy's convention is ascending
        return w[::-1].copy(), np.ascontiguousarray(U[:, ::-1])

    def eigh_desc(self, A):
        return self.api.sym_eig(A, device=self.device)   # already R's order: no 8 n^2-byte reversal copies

    def chol2inv(self, A):
        return np.ascontiguousarray(self.api.chol2inv(A, device=self.device))

    def inv(self, A):
        return np.ascontiguousarray(self.api.inverse(A, device=self.device))

    def mm(self, A, B):
        return np.ascontiguousarray(self.api.matmul(A, B, device=self.device))


_la = _HostLA()


def set_algebra(kind="host", device=0):
    """"host" (default) or "device": where eigh / chol2inv / inv / n x n products of the model algebra run."""
    global _la
    _la = _DeviceLA(device) if kind == "device" else _HostLA()
    return _la.name


def algebra():
    return _la


def calculateH(MMt, varE, varG):
    if varE < 0 or varG < 0:
        raise ValueError("variance components cannot be negative")  # calculateH.R:19-30
    n = MMt.shape[0]
    H = varG * MMt                 # varE * diag(n) + varG * MMt with one pass and no n x n identity
    H.flat[:: n + 1] += varE
    return H


def _chol2inv(A):
    return _la.chol2inv(A)


def calculateP(H, X):
    if H.shape[0] != X.shape[0]:
        raise ValueError("The number of rows in H and X are not the same.")  # calculateP.R:22-25
    Hinv = _chol2inv(H)
    HX = Hinv @ X
    T = HX @ np.linalg.solve(X.T @ HX, HX.T)
    return np.subtract(Hinv, T, out=T)


_mmt_sqrt_memo = {"key": None, "value": None}


def _content_key(A):
    """Identity of a matrix by content (a 128-bit hash of its bytes: 60 ms for 800 MB, against seconds of eigen-decomposition)."""
    import xxhash
    order = "C"
    if not A.flags.c_contiguous:
        if A.flags.f_contiguous:
            A, order = A.T, "F"       # the same bytes, read in place
        else:
            A = np.ascontiguousarray(A)
    return (A.shape, order, A.dtype.str, xxhash.xxh3_128_hexdigest(memoryview(A).cast("B")))


def calculateMMt_sqrt_and_sqrtinv(MMt, checkres=True):
    """eigen(MMt, symmetric=TRUE); sqrt = U diag(sqrt(l)) U^T ; invsqrt = chol2inv(chol(sqrt)).

    MMt is the same matrix in every iteration of an AM() run, yet find_qtl.R:19 calls this every time (1.6 s per iteration on the
    device, 12 s on host LAPACK at n = 10,000): the last result is kept and returned for a matrix with the same CONTENT.  The kept
    matrices are column-major, the layout the C ABI takes them in, so the wrapper does not copy 800 MB per call either."""
    key = (_la.name, _content_key(MMt))
    if _mmt_sqrt_memo["key"] == key:
        return _mmt_sqrt_memo["value"]
    r = _calculateMMt_sqrt_and_sqrtinv(MMt, checkres)
    r = {k: np.asfortranarray(v) for k, v in r.items()}
    _mmt_sqrt_memo["key"], _mmt_sqrt_memo["value"] = key, r
    return r


def _calculateMMt_sqrt_and_sqrtinv(MMt, checkres=True):
    """eigen(MMt, symmetric=TRUE); sqrt = U diag(sqrt(l)) U^T ; invsqrt = chol2inv(chol(sqrt))."""
    if _la.name == "device":  # the whole function in one C-ABI call, the matrices staying in HBM between its steps
        r = _la.api.mmt_sqrt_and_sqrtinv(MMt, device=_la.device)
        if r is None:
            raise ValueError("M %*% t(M) is not positive definite")  # :15-23
        sq, inv, tr = r
        if checkres and int(np.trunc(tr)) != MMt.shape[0]:  # :35-46
            import warnings
            warnings.warn("sqrt(MMt) %*% invsqrt(MMt) trace = %r, expected %d" % (tr, MMt.shape[0]))
        return {"sqrt_MMt": np.ascontiguousarray(sq), "inverse_sqrt_MMt": np.ascontiguousarray(inv)}
    evals, U = _la.eigh(MMt)
    if evals.min() <= 0:
        raise ValueError("M %*% t(M) is not positive definite")  # :15-23
    sq = _la.mm(U * np.sqrt(evals), U.T)
    sq = 0.5 * (sq + sq.T)
    inv = _chol2inv(sq)
    if checkres:  # :35-46
        tr = float(np.sum(sq * inv.T))  # trace(sq @ inv) without the n^3 product
        if int(np.trunc(tr)) != MMt.shape[0]:
            import warnings
            warnings.warn("sqrt(MMt) %*% invsqrt(MMt) trace = %r, expected %d" % (tr, MMt.shape[0]))
    return {"sqrt_MMt": sq, "inverse_sqrt_MMt": inv}


def calculate_reduced_a(varG, P, MMtsqrt, y):
    if P.shape[0] != np.size(y):
        raise ValueError("dimension mismatch between P and y")
    return varG * (MMtsqrt @ (P @ np.ravel(y)))  # two matrix-vector products


def calculate_reduced_vara(X, varE, varG, invMMt, MMtsqrt):
    """vars = varG*I - (D1 + D1 C (A - B D1 C)^-1 B D1)   (calculate_reduced_vara.R:21-35)."""
    n = invMMt.shape[0]
    Ze = MMtsqrt
    r1 = 1.0 / varE
    g1 = 1.0 / varG
    A = r1 * (X.T @ X)
    B = r1 * (X.T @ Ze)
    Cm = r1 * (Ze.T @ X)
    D = _la.mm(Ze.T, Ze)           # r1 * Ze'Ze + g1 * diag(n), in place: the same operations per element, two passes fewer
    D *= r1
    D.flat[:: n + 1] += g1
    D1 = _la.inv(D)
    D1C = D1 @ Cm
    BD1 = B @ D1
    mid = np.linalg.solve(A - B @ D1C, BD1)
    T = D1C @ mid
    np.add(D1, T, out=T)
    np.subtract(0.0, T, out=T)     # varG * diag(n) - (D1 + D1C mid): off the diagonal 0 - t, on it varG - t
    T.flat[:: n + 1] += varG
    return T


def scan_operands(MMt_norm, X, y, varE, varG):
    """Everything .find_qtl (E/R/find_qtl.R:5-49) builds before calling calculate_a_and_vara."""
    H = calculateH(MMt_norm, varE, varG)
    P = calculateP(H, X)
    sq = calculateMMt_sqrt_and_sqrtinv(MMt_norm, checkres=False)
    hat_a = calculate_reduced_a(varG, P, sq["sqrt_MMt"], y)
    invMMt = _chol2inv(MMt_norm)  # AM.R:422
    var_hat_a = calculate_reduced_vara(X, varE, varG, invMMt, sq["sqrt_MMt"])
    return {"S": sq["inverse_sqrt_MMt"], "Shalf": sq["sqrt_MMt"], "V": var_hat_a, "ahat": hat_a, "P": P}
