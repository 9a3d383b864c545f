"""Non-R driver of the AM() forward-selection loop -- host side, numpy/LAPACK.

This is the reference's model loop restated so that "the same SNPs are selected" can be demonstrated end to end on
a box without R: constructX -> calcMMt (GPU) -> emma.REMLE -> extBIC via emma.MLE -> find_qtl (GPU scan + arg-max).
Everything n x n stays on the host by design (north_star keeps emma_REMLE / calculateP / calculateH on host LAPACK);
the two marker-dimension operations go through a `backend` whose default is the HIP library (r_api).

Reference lines (E/ = MyPackage/Eagle/):
  AM loop ........................ E/R/AM.R:400-475 (stop rule :448, maxit :463)
  .calcVC ........................ E/R/calcVC.R:1-8           emma.REMLE  E/R/emma_REMLE.R:27-131
  .calc_extBIC ................... E/R/calc_extBIC.R:1-12     emma.MLE    E/R/emma_MLE.R:2-117
  emma.eigen.R.wo.Z / L.wo.Z ..... E/R/emma_eigen_R_wo_Z.R:2-21, E/R/emma_eigen_L_wo_Z.R:1-12
  constructX / extract_geno ...... E/R/constructX.R:1-24, E/R/extract_geno.R:1-19

Only the no-Z model is restated: with a Z matrix the reference's own .find_qtl fails (SURVEY.md section 8a, config 5 note).
parity unpinned: the reference records no outputs, and R's uniroot (Brent, tol = eps^0.25) is restated from the
published zeroin algorithm, not from R's source.
"""
import math

import numpy as np
from scipy.special import gammaln

from . import host_model

_EPS25 = np.finfo(np.float64).eps ** 0.25  # uniroot's default tol


def _zeroin(f, a, b, tol=_EPS25, maxit=1000):
    """Brent's zeroin (Forsythe, Malcolm & Moler), the algorithm behind R's uniroot."""
    fa, fb = f(a), f(b)
    c, fc = a, fa
    eps = np.finfo(np.float64).eps
    if fa == 0.0:
        return a
    if fb == 0.0:
        return b
    for _ in range(maxit + 1):
        prev_step = b - a
        if abs(fc) < abs(fb):
            a, b, c = b, c, b
            fa, fb, fc = fb, fc, fb
        tol_act = 2 * eps * abs(b) + tol / 2
        new_step = (c - b) / 2
        if abs(new_step) <= tol_act or fb == 0.0:
            return b
        if abs(prev_step) >= tol_act and abs(fa) > abs(fb):
            cb = c - b
            if a == c:
                t1 = fb / fa
                p = cb * t1
                q = 1.0 - t1
            else:
                q = fa / fc
                t1 = fb / fc
                t2 = fb / fa
                p = t2 * (cb * q * (q - t1) - (b - a) * (t1 - 1.0))
                q = (q - 1.0) * (t1 - 1.0) * (t2 - 1.0)
            if p > 0:
                q = -q
            else:
                p = -p
            if p < (0.75 * cb * q - abs(tol_act * q) / 2) and p < abs(prev_step * q / 2):
                new_step = p / q
        if abs(new_step) < tol_act:
            new_step = tol_act if new_step > 0 else -tol_act
        a, fa = b, fb
        b += new_step
        fb = f(b)
        if (fb > 0 and fc > 0) or (fb < 0 and fc < 0):
            c, fc = a, fa
    return b


def emma_eigen_L_wo_Z(K):
    ev, U = host_model.algebra().eigh_desc(K)  # R's eigen(): decreasing order
    return {"values": np.ascontiguousarray(ev), "vectors": U if U.flags.f_contiguous or U.flags.c_contiguous else U.copy()}


def emma_eigen_R_wo_Z(K, X):
    n, q = X.shape
    la = host_model.algebra()
    S = X @ np.linalg.solve(X.T @ X, X.T)      # S = diag(n) - X (X'X)^-1 X', without the n x n identity
    np.negative(S, out=S)
    S.flat[:: n + 1] += 1.0
    K1 = K.copy()                              # K + diag(n)
    K1.flat[:: n + 1] += 1.0
    ev, U = la.eigh_desc(la.mm(la.mm(S, K1), S))
    return {"values": ev[: n - q] - 1.0, "vectors": U[:, : n - q].copy()}


def _grid(ngrids, llim, ulim):
    logdelta = np.arange(ngrids + 1) / ngrids * (ulim - llim) + llim
    return logdelta, np.exp(logdelta)


def _reml_ll(logdelta, lam, etas):
    nq = etas.size
    d = math.exp(logdelta)
    return 0.5 * (nq * (math.log(nq / (2 * math.pi)) - 1 - math.log(np.sum(etas * etas / (lam + d)))) - np.sum(np.log(lam + d)))


def _reml_dll(logdelta, lam, etas):
    nq = etas.size
    d = math.exp(logdelta)
    ld = lam + d
    e2 = etas * etas
    return 0.5 * (nq * np.sum(e2 / (ld * ld)) / np.sum(e2 / ld) - np.sum(1.0 / ld))


def _ml_ll(logdelta, lam, etas, xi):
    n = xi.size
    d = math.exp(logdelta)
    return 0.5 * (n * (math.log(n / (2 * math.pi)) - 1 - math.log(np.sum(etas * etas / (lam + d)))) - np.sum(np.log(xi + d)))


def _ml_dll(logdelta, lam, etas, xi):
    n = xi.size
    d = math.exp(logdelta)
    ld = lam + d
    e2 = etas * etas
    return 0.5 * (n * np.sum(e2 / (ld * ld)) / np.sum(e2 / ld) - np.sum(1.0 / (xi + d)))


def _optimise(dLL, logdelta, llim, ulim, esp, ll_fn, dll_fn):
    """The bracket rule shared by emma.REMLE (:60-76) and emma.MLE (:43-60)."""
    m = logdelta.size
    opt_ld, opt_ll = [], []
    if dLL[0] < esp:
        opt_ld.append(llim); opt_ll.append(ll_fn(llim))
    if dLL[m - 2] > 0 - esp:
        opt_ld.append(ulim); opt_ll.append(ll_fn(ulim))
    for i in range(m - 1):
        if dLL[i] * dLL[i + 1] < 0 - esp * esp and dLL[i] > 0 and dLL[i + 1] < 0:
            r = _zeroin(dll_fn, logdelta[i], logdelta[i + 1])
            opt_ld.append(r); opt_ll.append(ll_fn(r))
    k = int(np.argmax(opt_ll))
    return math.exp(opt_ld[k]), opt_ll[k]


def emma_REMLE(y, X, K, ngrids=100, llim=-10, ulim=10, esp=1e-10, eig_R=None):
    n, q = y.size, X.shape[1]
    if np.linalg.det(X.T @ X) == 0:
        return {"REML": 0, "delta": 0, "ve": 0, "vg": 0}
    if eig_R is None:
        eig_R = emma_eigen_R_wo_Z(K, X)
    lam = eig_R["values"]
    etas = eig_R["vectors"].T @ y
    logdelta, delta = _grid(ngrids, llim, ulim)
    Lam = lam[:, None] + delta[None, :]
    E2 = (etas * etas)[:, None]
    dLL = 0.5 * delta * ((n - q) * np.sum(E2 / (Lam * Lam), axis=0) / np.sum(E2 / Lam, axis=0) - np.sum(1.0 / Lam, axis=0))
    maxdelta, maxLL = _optimise(dLL, logdelta, llim, ulim, esp, lambda ld: _reml_ll(ld, lam, etas), lambda ld: _reml_dll(ld, lam, etas))
    maxva = np.sum(etas * etas / (lam + maxdelta)) / (n - q)
    return {"REML": maxLL, "delta": maxdelta, "ve": maxva * maxdelta, "vg": maxva}


def emma_MLE(y, X, K, ngrids=100, llim=-10, ulim=10, esp=1e-10, eig_L=None, eig_R=None):
    n = y.size
    if np.linalg.det(X.T @ X) == 0:
        return {"ML": 0, "delta": 0, "ve": 0, "vg": 0}
    if eig_L is None:
        eig_L = emma_eigen_L_wo_Z(K)
    if eig_R is None:
        eig_R = emma_eigen_R_wo_Z(K, X)
    lam, xi = eig_R["values"], eig_L["values"]
    etas = eig_R["vectors"].T @ y
    logdelta, delta = _grid(ngrids, llim, ulim)
    Lam = lam[:, None] + delta[None, :]
    Xis = xi[:, None] + delta[None, :]
    E2 = (etas * etas)[:, None]
    dLL = 0.5 * delta * (n * np.sum(E2 / (Lam * Lam), axis=0) / np.sum(E2 / Lam, axis=0) - np.sum(1.0 / Xis, axis=0))
    maxdelta, maxLL = _optimise(dLL, logdelta, llim, ulim, esp, lambda ld: _ml_ll(ld, lam, etas, xi),
                                lambda ld: _ml_dll(ld, lam, etas, xi))
    maxva = np.sum(etas * etas / (lam + maxdelta)) / n
    return {"ML": maxLL, "delta": maxdelta, "ve": maxva * maxdelta, "vg": maxva}


def calcVC(trait, currentX, MMt, eig_R=None):
    r = emma_REMLE(trait, currentX, MMt, eig_R=eig_R)
    return {"vg": r["vg"], "ve": r["ve"]}


def _lchoose(n, k):
    return gammaln(n + 1) - gammaln(k + 1) - gammaln(n - k + 1)


def calc_extBIC(trait, currentX, MMt, nmarkers, eig_L=None, eig_R=None):
    res = emma_MLE(trait, currentX, MMt, llim=-100, ulim=100, eig_L=eig_L, eig_R=eig_R)
    BIC = -2 * res["ML"] + (currentX.shape[1] + 1) * math.log(trait.size)
    return BIC + 2 * _lchoose(nmarkers, currentX.shape[1] - 1)


class HipBackend:
    """The two marker-dimension calls of the loop on the GPU (r_api mirrors the R wrappers' marshalling)."""

    def __init__(self, device=0):
        from . import r_api, rcpp_api
        self.r_api, self.rcpp_api, self.device = r_api, rcpp_api, device

    def calcMMt(self, geno, availmemGb, ncpu, selected_loci, quiet):
        return self.r_api.calcMMt(geno, availmemGb, ncpu, selected_loci, quiet, device=self.device)

    def find_qtl(self, **kw):
        return self.r_api.find_qtl(device=self.device, **kw)

    def extract_geno(self, geno, colnum):
        """extract_geno.R:8-12: column colnum (1-based) of M.ascii, served from the HBM-resident copy calcMMt left."""
        return self.r_api.extract_geno(geno["asciifileM"], colnum, dim_of_ascii_M=geno["dim_of_ascii_M"],
                                       device=self.device).astype(np.int64)


class SpectralBackend(HipBackend):
    """The same loop with the scan in the eigenbasis of MM^T (include/eagle_hip.h section 1d; OPT-IN, needs the R-side change
    INTEGRATION.md describes): K = MM^T/max + 0.95 I is fixed for the whole run, so Z = Mt U is made once after calcMMt and
    every find_qtl is one HBM-bound pass over Z instead of an n x n quadratic form per marker.  Selects the markers the
    reference-shaped path selects (tests/test_am_driver.py)."""

    def __init__(self, device=0):
        super().__init__(device)
        self.lam = self.U = None
        self.L = None

    def calcMMt(self, geno, availmemGb, ncpu, selected_loci, quiet):
        MMt = super().calcMMt(geno, availmemGb, ncpu, selected_loci, quiet)
        self.lam, self.U = np.linalg.eigh(MMt)                       # the decomposition emma.REMLE needs anyway
        n, self.L = geno["dim_of_ascii_M"]
        self.rcpp_api.spectral_prepare(geno["asciifileMt"], (self.L, n), self.U, availmemGb, device=self.device)
        return MMt

    def find_qtl(self, geno, availmemGb, selected_loci, MMt, invMMt, best_ve, best_vg, currentX, ncpu, quiet, trait):
        res = self.rcpp_api.spectral_scan(self.lam, self.U.T @ currentX, self.U.T @ np.ravel(trait), best_ve, best_vg, self.L,
                                          selected_loci, device=self.device)
        with np.errstate(all="ignore"):
            tsq = res["a"].ravel() ** 2 / res["vara"].ravel()
        return int(np.flatnonzero(tsq == np.nanmax(tsq))[0]) + 1         # find_qtl.R:71-83


def AM(trait, X, geno, availmemGb=8, ncpu=1, maxit=20, quiet=True, backend=None, message=None, algebra=None):
    """E/R/AM.R:400-475 for a complete-data trait vector and a ready design matrix X (n x q, intercept included).

    Returns dict(selected_loci = 1-based marker columns in order of selection, extBIC = list, ve, vg of the last fit).
    selected_loci starts as [NA] exactly like AM.R:260, so the selected_loci masking never fires."""
    backend = backend or HipBackend()
    if algebra is not None:  # "host" (LAPACK, the reference's placement) or "device" (rocSOLVER / the fp64 MFMA GEMM through the C ABI)
        host_model.set_algebra(algebra)
    say = message or (lambda *_: None)
    trait = np.asarray(trait, dtype=np.float64).ravel()
    currentX = np.asarray(X, dtype=np.float64)
    nmarkers = geno["dim_of_ascii_M"][1]
    selected_loci = [np.nan]
    new_selected_locus = np.nan
    extBIC = []
    itnum, cont = 1, True
    MMt = invMMt = eig_L = None
    best = {}
    while cont:
        say("Iteration %d: Searching for most significant marker-trait association" % itnum)
        if not (isinstance(new_selected_locus, float) and math.isnan(new_selected_locus)):  # constructX.R:10-22
            currentX = np.column_stack([currentX, backend.extract_geno(geno, int(new_selected_locus)).astype(np.float64)])
        if itnum == 1:  # AM.R:414-422
            MMt = backend.calcMMt(geno, availmemGb, ncpu, np.array(selected_loci), quiet)
            invMMt = host_model._chol2inv(MMt)
            eig_L = emma_eigen_L_wo_Z(MMt)  # depends on MMt only; the reference recomputes it every iteration
        eig_R = emma_eigen_R_wo_Z(MMt, currentX)  # shared by REMLE and MLE of this iteration (same K, X)
        best = calcVC(trait, currentX, MMt, eig_R=eig_R)
        extBIC.append(calc_extBIC(trait, currentX, MMt, nmarkers, eig_L=eig_L, eig_R=eig_R))
        if int(np.flatnonzero(np.asarray(extBIC) == min(extBIC))[0]) == len(extBIC) - 1:  # AM.R:448
            new_selected_locus = backend.find_qtl(geno=geno, availmemGb=availmemGb, selected_loci=np.array(selected_loci), MMt=MMt,
                                                  invMMt=invMMt, best_ve=best["ve"], best_vg=best["vg"], currentX=currentX,
                                                  ncpu=ncpu, quiet=quiet, trait=trait)
            selected_loci.append(new_selected_locus)
        else:
            cont = False
        itnum += 1
        if itnum > maxit:  # AM.R:463-470
            cont = False
    # AM.R:476-499.  Stopped by maxit: every pick is reported (the in-loop report that drops the last pick is
    # overwritten by the one after the loop).  Stopped by extBIC: the last pick made extBIC worse and is dropped
    # together with its extBIC entry.
    picks = [int(v) for v in selected_loci[1:]]
    if itnum > maxit or len(selected_loci) <= 1:
        loci, ext = picks, list(extBIC)
    else:
        loci = picks[:-1]
        ext = [v for i, v in enumerate(extBIC) if i != len(selected_loci) - 1]
    return {"selected_loci": loci, "all_picks": picks, "extBIC": ext, "extBIC_trace": list(extBIC), "ve": best.get("ve"),
            "vg": best.get("vg")}
