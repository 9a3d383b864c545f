"""The non-R AM() loop driver (eagleeverything_amd/am.py).

CPU: the loop with an oracle-backed backend finds planted QTL and follows the reference's stop/report rules.
GPU (-m gpu): the HIP-backed loop selects exactly the same markers, with the same extBIC trace, as the
oracle-backed loop on the reference's demo data.  parity unpinned (no recorded outputs in the reference)."""
import numpy as np
import pytest

from eagleeverything_amd import am, host_model, synth

NA = np.nan


class OracleBackend:
    """calcMMt / find_qtl / extract_geno through the CPU oracle (tests only)."""

    def __init__(self, oracle):
        self.o = oracle

    def calcMMt(self, geno, availmemGb, ncpu, selected_loci, quiet):
        n, L = geno["dim_of_ascii_M"]
        sel = np.atleast_1d(np.asarray(selected_loci, dtype=np.float64))
        if not np.any(np.isnan(sel)):
            sel = sel - 1  # calculateMMt.R:24
        return self.o.normalise_MMt(self.o.calculateMMt_rcpp(geno["asciifileM"], availmemGb, ncpu, sel, (n, L)))

    def find_qtl(self, geno, availmemGb, selected_loci, MMt, invMMt, best_ve, best_vg, currentX, ncpu, quiet, trait):
        n, L = geno["dim_of_ascii_M"]
        H = host_model.calculateH(MMt, best_ve, best_vg)
        P = host_model.calculateP(H, currentX)
        sq = host_model.calculateMMt_sqrt_and_sqrtinv(MMt, checkres=False)
        hat_a = host_model.calculate_reduced_a(best_vg, P, sq["sqrt_MMt"], trait)
        var_hat_a = host_model.calculate_reduced_vara(currentX, best_ve, best_vg, invMMt, sq["sqrt_MMt"])
        sel = np.atleast_1d(np.asarray(selected_loci, dtype=np.float64))
        if not np.any(np.isnan(sel)):
            sel = sel - 1  # calculate_a_and_vara.R:23
        res = self.o.calculate_a_and_vara_rcpp(geno["asciifileMt"], sel, sq["inverse_sqrt_MMt"], var_hat_a, availmemGb, (L, n), hat_a)
        return self.o.tsq_argmax(res["a"], res["vara"])[1]

    def extract_geno(self, geno, colnum):
        n = geno["dim_of_ascii_M"][0]
        return self.o.ReadBlock(geno["asciifileMt"], colnum - 1, n, 1).ravel().astype(np.int64)


def _planted(tmp_path, n=120, L=600, seed=4):
    Mt8 = synth.genotypes_marker_major(n, L, seed=seed)
    rng = np.random.default_rng(seed)
    qtl = [50, 333]
    y = 1.5 * Mt8[qtl[0]] - 1.2 * Mt8[qtl[1]] + 0.5 * rng.standard_normal(n)
    geno = synth.write_geno_pair(str(tmp_path), Mt8)
    return geno, y.astype(np.float64), np.ones((n, 1)), qtl


def test_am_loop_finds_planted_qtl_cpu(oracle, tmp_path):
    geno, y, X, qtl = _planted(tmp_path)
    res = am.AM(y, X, geno, availmemGb=8, ncpu=2, maxit=10, backend=OracleBackend(oracle))
    assert set(q + 1 for q in qtl) <= set(res["selected_loci"])   # 1-based columns
    tr = res["extBIC_trace"]
    assert len(res["all_picks"]) == len(res["selected_loci"]) + 1   # the pick that made extBIC worse is dropped
    assert tr[-1] > min(tr) and np.argmin(tr) == len(tr) - 2         # stop rule AM.R:448
    assert len(res["extBIC"]) == len(tr) - 1
    assert res["vg"] >= 0 and res["ve"] > 0


def test_am_maxit_reports_every_pick(oracle, tmp_path):
    geno, y, X, _ = _planted(tmp_path)
    res = am.AM(y, X, geno, maxit=2, backend=OracleBackend(oracle))
    assert len(res["extBIC_trace"]) == 2 and res["selected_loci"] == res["all_picks"] and len(res["all_picks"]) == 2


def test_emma_pieces_are_consistent():
    rng = np.random.default_rng(0)
    n = 60
    A = rng.standard_normal((n, 200))
    K = A @ A.T / 200 + 0.95 * np.eye(n)
    X = np.column_stack([np.ones(n), rng.standard_normal(n)])
    y = rng.multivariate_normal(X @ [1.0, 0.5], 0.7 * K + 1.3 * np.eye(n))
    r = am.emma_REMLE(y, X, K)
    m = am.emma_MLE(y, X, K, llim=-100, ulim=100)
    assert r["ve"] > 0 and r["vg"] >= 0 and abs(r["delta"] - r["ve"] / r["vg"]) < 1e-9 * r["delta"]
    # the REML optimum is a stationary point of the restricted likelihood (or sits on the grid boundary)
    eig = am.emma_eigen_R_wo_Z(K, X)
    etas = eig["vectors"].T @ y
    ld = np.log(r["delta"])
    if -10 < ld < 10:
        assert abs(am._reml_dll(ld, eig["values"], etas)) < 1e-3
    assert np.isfinite(m["ML"]) and m["ve"] > 0
    # eigen R: n-q values, vectors orthogonal to X
    assert eig["values"].size == n - 2 and np.abs(X.T @ eig["vectors"]).max() < 1e-8
    # zeroin agrees with a bracketing reference root to its tolerance
    root = am._zeroin(lambda x: x ** 3 - 2.0, 0.0, 3.0)
    assert abs(root - 2.0 ** (1 / 3)) < 2e-4


def test_W_is_varG2_P_inside_AM(golden):
    """What eagle_scan_with_W rests on: with the operands .find_qtl builds (E/R/find_qtl.R:5-49), dim_reduced_vara =
    varG I - C22 = varG^2 Ze P Ze (Henderson; calculate_reduced_vara.R:21-35) and inv_MMt_sqrt = Ze^-1, so the W = S V S the
    reference forms with two n^3 products per call is varG^2 P, and v = S a_hat is varG P y."""
    g = golden("genoDemo_150x4998")
    n = 150
    MMtn = g["MMt"] / g["MMt"].max() + 0.95 * np.eye(n)
    for varE, varG, X in ((1.0, 0.5, g["X"]), (0.3, 2.0, np.column_stack([g["X"], g["M8"][:, [17, 900]].astype(float)]))):
        ops = host_model.scan_operands(MMtn, X, g["y"], varE, varG)
        W = ops["S"] @ (ops["V"] @ ops["S"])
        np.testing.assert_allclose(W, varG ** 2 * ops["P"], rtol=0, atol=1e-10 * np.abs(W).max())
        np.testing.assert_allclose(ops["S"] @ ops["ahat"], varG * (ops["P"] @ np.ravel(g["y"])), rtol=1e-9, atol=1e-12)


@pytest.mark.gpu
def test_scan_with_W_shortcut_matches_the_reference_shaped_scan(golden, tmp_path):
    from eagleeverything_amd import rcpp_api
    g = golden("genoDemo_150x4998")
    n, L = g["M8"].shape
    geno = synth.write_geno_pair(str(tmp_path), np.ascontiguousarray(g["M8"].T))
    MMtn = g["MMt"] / g["MMt"].max() + 0.95 * np.eye(n)
    varE, varG = 1.0, 0.5
    ops = host_model.scan_operands(MMtn, g["X"], g["y"], varE, varG)
    ref = rcpp_api.calculate_a_and_vara_rcpp(geno["asciifileMt"], np.nan, ops["S"], ops["V"], 8.0, (L, n), ops["ahat"])
    idx_ref = rcpp_api.last_scan_argmax()[0]
    res = rcpp_api.scan_with_W(geno["asciifileMt"], np.nan, varG ** 2 * ops["P"], varG * (ops["P"] @ np.ravel(g["y"])), 8.0, (L, n))
    assert rcpp_api.last_scan_argmax()[0] == idx_ref
    np.testing.assert_allclose(res["a"], ref["a"], rtol=1e-8, atol=1e-10 * np.abs(ref["a"]).max())
    vs = np.abs(ref["vara"]).max()
    np.testing.assert_allclose(res["vara"], ref["vara"], rtol=9e-7, atol=1e-10 * vs)
    # masking and a non-symmetric W work as in the reference-shaped call
    W = np.random.default_rng(0).standard_normal((n, n)) * 0.01 + np.eye(n)
    v = np.random.default_rng(1).standard_normal(n)
    sel = np.array([3.0, 77.0])
    res2 = rcpp_api.scan_with_W(geno["asciifileMt"], sel, W, v, 8.0, (L, n))
    Mt = g["M8"].T.astype(np.float64)
    a_ref = Mt @ v
    v_ref = np.einsum("ij,jk,ik->i", Mt, W, Mt)
    a_ref[[3, 77]] = 0.0; v_ref[[3, 77]] = 0.0
    np.testing.assert_allclose(res2["a"].ravel(), a_ref, rtol=1e-9, atol=1e-11 * np.abs(a_ref).max())
    np.testing.assert_allclose(res2["vara"].ravel(), v_ref, rtol=9e-7, atol=1e-9 * np.abs(v_ref).max())
    rcpp_api.drop_cache()


@pytest.mark.gpu
def test_spectral_scan_matches_the_reference_shaped_scan(golden, tmp_path):
    """include/eagle_hip.h section 1d: Z = Mt U once, then a and vara of every marker from one pass over Z -- against the
    reference-shaped scan fed with the S, V, a_hat that find_qtl.R builds from the same K, X, y, varE, varG."""
    from eagleeverything_amd import rcpp_api
    g = golden("genoDemo_150x4998")
    n, L = g["M8"].shape
    geno = synth.write_geno_pair(str(tmp_path), np.ascontiguousarray(g["M8"].T))
    K = g["MMt"] / g["MMt"].max() + 0.95 * np.eye(n)
    lam, U = np.linalg.eigh(K)
    rcpp_api.spectral_prepare(geno["asciifileMt"], (L, n), U, 8.0)
    y = np.ravel(g["y"])
    rng = np.random.default_rng(3)
    cases = [(1.0, 0.5, g["X"]),
             (0.3, 2.0, np.column_stack([g["X"], g["M8"][:, [17, 900, 4000]].astype(float)])),
             (0.7, 0.9, np.column_stack([g["X"], rng.standard_normal((n, 19))]))]        # p = 20: the 32-column form of the kernel
    for varE, varG, X in cases:
        ops = host_model.scan_operands(K, X, y, varE, varG)
        ref = rcpp_api.calculate_a_and_vara_rcpp(geno["asciifileMt"], np.nan, ops["S"], ops["V"], 8.0, (L, n), ops["ahat"])
        idx_ref = rcpp_api.last_scan_argmax()[0]
        res = rcpp_api.spectral_scan(lam, U.T @ X, U.T @ y, varE, varG, L)
        np.testing.assert_allclose(res["a"], ref["a"], rtol=1e-8, atol=1e-10 * np.abs(ref["a"]).max())
        np.testing.assert_allclose(res["vara"], ref["vara"], rtol=9e-7, atol=1e-10 * np.abs(ref["vara"]).max())
        with np.errstate(all="ignore"):
            tsq = res["a"].ravel() ** 2 / res["vara"].ravel()
        assert int(np.nanargmax(tsq)) + 1 == idx_ref
    # masking rule of the reference
    res_m = rcpp_api.spectral_scan(lam, U.T @ g["X"], U.T @ y, 1.0, 0.5, L, selected_loci=np.array([5.0, 4000.0]))
    assert res_m["a"][5, 0] == 0.0 and res_m["vara"][4000, 0] == 0.0 and res_m["a"][6, 0] != 0.0
    rcpp_api.drop_cache()


@pytest.mark.gpu
def test_spectral_zbuild_int8_slices_against_fp64():
    """Z = Mt U from six exact int8 digit slices of U (default) against the fp64 MFMA form and against numpy, within the
    documented bound (sum_j |m_ij|) 2^(e+1-48); ragged marker count (last 384-marker tile short), n not a multiple of 256."""
    import ctypes as C
    import torch
    from eagleeverything_amd.sharded import DeviceShard
    n, L = 1003, 5000
    sh = DeviceShard(n, L)
    sh.fill_synthetic(seed=9)
    rng = np.random.default_rng(4)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    Ur = torch.zeros((sh.np_, sh.np_), dtype=torch.float64, device=sh.dev)
    Ur[:n, :n] = torch.from_numpy(Q).to(sh.dev)
    Z64 = torch.empty((sh.Lp, sh.np_), dtype=torch.float64, device=sh.dev)
    Z8 = torch.full((sh.Lp, sh.np_), float("nan"), dtype=torch.float64, device=sh.dev)
    lib, ctx = sh.L, sh.ctx
    st = C.c_void_p(torch.cuda.current_stream(sh.dev).cuda_stream)
    sh._check(lib.eagle_dev_spectral_zbuild(ctx, sh.Mt8.data_ptr(), sh.Lp, sh.np_, sh.np_, Ur.data_ptr(), Z64.data_ptr(), st))
    ws = torch.empty(int(lib.eagle_spectral_zbuild_i8_workspace_bytes(sh.np_, 6)), dtype=torch.uint8, device=sh.dev)
    sh._check(lib.eagle_dev_spectral_zbuild_i8(ctx, sh.Mt8.data_ptr(), sh.Lp, sh.np_, sh.np_, Ur.data_ptr(), Z8.data_ptr(), ws.data_ptr(), 6, st))
    torch.cuda.synchronize()
    ref = sh.Mt8[:L, :n].double() @ Ur[:n, :n]
    l1 = sh.Mt8[:L, :n].abs().sum(dim=1, dtype=torch.int64).double()
    assert torch.allclose(Z64[:L, :n], ref, rtol=0, atol=1e-12)
    bound = l1[:, None] * 2.0 ** (0 + 1 - 48) + 1e-13                # e <= 0 for an orthogonal matrix
    assert bool(((Z8[:L, :n] - ref).abs() <= bound).all())
    assert bool((Z8[L:, :] == 0).all()) and bool((Z8[:, n:] == 0).all())  # padding rows / columns come out as exact zeros


@pytest.mark.gpu
def test_am_spectral_backend_selects_the_same_markers(golden, tmp_path):
    g = golden("genoDemo_150x4998")
    geno = synth.write_geno_pair(str(tmp_path), np.ascontiguousarray(g["M8"].T))
    ref = am.AM(g["y"], g["X"], geno, maxit=5)
    spec = am.AM(g["y"], g["X"], geno, maxit=5, backend=am.SpectralBackend())
    assert spec["all_picks"] == ref["all_picks"] and spec["selected_loci"] == ref["selected_loci"]
    np.testing.assert_allclose(spec["extBIC_trace"], ref["extBIC_trace"], rtol=1e-12)


@pytest.mark.gpu
def test_am_hip_selects_same_markers_as_oracle(oracle, golden, tmp_path):
    g = golden("genoDemo_150x4998")
    geno = synth.write_geno_pair(str(tmp_path), np.ascontiguousarray(g["M8"].T))
    ref = am.AM(g["y"], g["X"], geno, maxit=6, backend=OracleBackend(oracle))
    hip = am.AM(g["y"], g["X"], geno, maxit=6)
    assert hip["all_picks"] == ref["all_picks"] and hip["selected_loci"] == ref["selected_loci"]
    np.testing.assert_allclose(hip["extBIC_trace"], ref["extBIC_trace"], rtol=1e-9)
    assert len(hip["all_picks"]) >= 2


@pytest.mark.gpu
def test_readmarker_to_am_end_to_end(oracle, golden, tmp_path):
    """BASELINE configs[0] in this build's terms: the reference's demo genotypes as a whitespace-separated 0/1/2 table
    (what MyPackage/genoDemo.dat is) -> ReadMarker() -> AM(), every hot call on the GPU; the oracle-backed loop on
    oracle-converted files must pick the same markers with the same extBIC trace."""
    from eagleeverything_amd import r_api
    g = golden("genoDemo_150x4998")
    raw = tmp_path / "genoDemo.dat"
    with open(raw, "w") as f:
        for row in (g["M8"] + 1):
            f.write(" ".join(map(str, row)) + "\n")
    d_hip, d_ref = tmp_path / "hip", tmp_path / "ref"
    d_hip.mkdir(), d_ref.mkdir()
    geno = r_api.ReadMarker(str(raw), type="text", AA=0, AB=1, BB=2, outdir=str(d_hip))
    assert geno is not None and geno["dim_of_ascii_M"] == [150, 4998]
    dims = oracle.getRowColumn(str(raw))
    ok, _ = oracle.createM_ASCII_rcpp(str(raw), str(d_ref / "M.ascii"), "text", 0, 1, 2, 8, dims)
    assert ok
    oracle.createMt_ASCII_rcpp(str(d_ref / "M.ascii"), str(d_ref / "Mt.ascii"), "text", 8, dims)
    geno_ref = {"asciifileM": str(d_ref / "M.ascii"), "asciifileMt": str(d_ref / "Mt.ascii"), "dim_of_ascii_M": dims}
    ref = am.AM(g["y"], g["X"], geno_ref, maxit=5, backend=OracleBackend(oracle))
    hip = am.AM(g["y"], g["X"], geno, maxit=5)
    assert hip["all_picks"] == ref["all_picks"] and hip["selected_loci"] == ref["selected_loci"]
    np.testing.assert_allclose(hip["extBIC_trace"], ref["extBIC_trace"], rtol=1e-9)


@pytest.mark.gpu
def test_am_device_algebra_matches_host_algebra(golden, tmp_path):
    """eigh / chol2inv / inv / n x n products of the model algebra on the GPU through the C ABI
    (eagle_sym_eig, eagle_chol2inv, eagle_inverse, eagle_matmul, eagle_mmt_sqrt_and_sqrtinv; ctypes, no torch in the path);
    the AM loop picks the same markers and the extBIC trace agrees to 1e-8 with host LAPACK."""
    import inspect
    assert "import torch" not in inspect.getsource(host_model) and "torch" not in inspect.getsource(am)
    g = golden("genoDemo_150x4998")
    geno = synth.write_geno_pair(str(tmp_path), np.ascontiguousarray(g["M8"].T))
    try:
        host = am.AM(g["y"], g["X"], geno, maxit=5, algebra="host")
        dev = am.AM(g["y"], g["X"], geno, maxit=5, algebra="device")
    finally:
        host_model.set_algebra("host")
    assert dev["all_picks"] == host["all_picks"] and dev["selected_loci"] == host["selected_loci"]
    np.testing.assert_allclose(dev["extBIC_trace"], host["extBIC_trace"], rtol=1e-8)
    # the pieces themselves
    MMtn = g["MMt"] / g["MMt"].max() + 0.95 * np.eye(150)
    ops_h = host_model.scan_operands(MMtn, g["X"], g["y"], 1.0, 0.5)
    host_model.set_algebra("device")
    try:
        ops_d = host_model.scan_operands(MMtn, g["X"], g["y"], 1.0, 0.5)
    finally:
        host_model.set_algebra("host")
    for k in ("S", "V", "ahat", "P"):
        np.testing.assert_allclose(ops_d[k], ops_h[k], rtol=1e-8, atol=1e-10 * np.abs(ops_h[k]).max())
