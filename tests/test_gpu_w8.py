"""W = S (V S) on the int8 MFMA from exact digit slices (csrc/eagle_w8.hip; E/src/calculate_a_and_vara_rcpp.cpp:97-98).

What makes the int8 W legitimate is checked here: the folded image it delivers is within its own Frobenius bound of the fp64
product (and the bound within the few per cent of the digit budget it was chosen for), the bits do not depend on the tile engine or on
how the products were cut into row panels, the scan that follows returns the fp64 scan's marker with every vara inside the budget,
markers the certificate re-evaluates carry the fp64 value m^T S V S m, and operands the configurations cannot certify (non-finite,
visibly asymmetric, an overflowing certificate) end on the fp64 products with the fp64 path's results."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _model_operands(torch, sh, seed=7):
    """S = MMt^-1/2, V = Var(a_hat), a_hat from the model algebra on the shard's own MM^T (as bench.py manufactures them)."""
    import bench
    c32 = sh.mmt_partial()
    MMt, _ = sh.mmt_finish(c32, normalise=True)
    gen = torch.Generator(device=sh.dev)
    gen.manual_seed(seed)
    y = torch.randn(sh.n, generator=gen, device=sh.dev, dtype=torch.float64)
    X = torch.ones((sh.n, 1), dtype=torch.float64, device=sh.dev)
    S, V, ahat, _, _ = bench.host_operands_torch(torch, MMt, X, y, 1.0, 0.5)
    return S, V, ahat


def _fold_err(torch, Wa, Wb):
    """|| . ||_F of the symmetric matrix the difference of two folded images stands for (2 W_jk above the diagonal)."""
    d = Wa - Wb
    dd = torch.diagonal(d)
    return float(torch.sqrt((dd * dd).sum() + 0.5 * ((d * d).sum() - (dd * dd).sum())))


def _shard(n, L, seed=11):
    import torch
    from eagleeverything_amd.sharded import DeviceShard
    sh = DeviceShard(n, L)
    sh.fill_synthetic(seed=seed)
    sh.mode = 1
    return torch, sh


def test_int8_w_is_within_its_bound_and_does_not_depend_on_engine_or_panels():
    torch, sh = _shard(1500, 4096)
    S, V, ahat = _model_operands(torch, sh)
    sh.set_operands(S, V, ahat)
    try:
        sh.w_mode = 0
        sh.scan_operands()
        torch.cuda.synchronize()
        assert sh.w_info()["int8"] == 0 and sh.w_info()["declined"] == 7
        W64, v64 = sh.Wu.clone(), sh.v.clone()
        sh.w_mode = 2
        sh.scan_operands()
        torch.cuda.synchronize()
        info = sh.w_info()
        assert info["int8"] == 1 and info["declined"] == 0, info
        W8 = sh.Wu.clone()
        assert torch.equal(sh.v, v64)                                    # v = S a_hat is the same fp64 product either way
        err = _fold_err(torch, W8, W64)
        assert 0.0 < err <= info["eta"], (err, info)
        assert info["eta"] <= 0.05 * 5e-7 * info["mean_diag"] and info["eta"] >= 50 * err    # rigorous, and known to be pessimistic
        assert abs(info["mean_diag"] - float(torch.diagonal(W64)[:sh.n].abs().mean())) <= 1e-9 * info["mean_diag"]
        assert torch.count_nonzero(torch.tril(W8, -1)) == 0              # folded: nothing below the diagonal
        # the same bits from the compiler-scheduled 256 x 256 engine, and with the products cut into row panels
        sh.L.eagle_dev_set_tune(sh.ctx, 31)
        sh.scan_operands()
        torch.cuda.synchronize()
        assert torch.equal(sh.Wu, W8)
        sh.L.eagle_dev_set_tune(sh.ctx, 0)
        os.environ["EAGLE_HIP_W8_LEVEL_MB"] = "24"                      # 1536^2 x 4 bytes = 9.4 MB per level image: two row panels and more
        sh.scan_operands()
        torch.cuda.synchronize()
        assert torch.equal(sh.Wu, W8)
        sh.L.eagle_dev_set_tune(sh.ctx, 31)
        sh.scan_operands()
        torch.cuda.synchronize()
        assert torch.equal(sh.Wu, W8)
    finally:
        os.environ.pop("EAGLE_HIP_W8_LEVEL_MB", None)
        sh.L.eagle_dev_set_tune(sh.ctx, 0)


def test_scan_on_the_int8_w_returns_the_fp64_scan():
    torch, sh = _shard(1500, 8192)
    S, V, ahat = _model_operands(torch, sh)
    sh.set_operands(S, V, ahat)
    L = sh.Lloc
    sh.mode = 0
    sh.scan()
    torch.cuda.synchronize()
    v64, a64, best64 = sh.vara[:L].clone(), sh.a[:L].clone(), sh.best()[:2]
    sh.mode = 1
    res = {}
    for wm in (0, 2):
        sh.w_mode = wm
        sh.scan()
        torch.cuda.synchronize()
        res[wm] = (sh.vara[:L].clone(), sh.a[:L].clone(), sh.best()[:2], sh.certificate(), sh.w_info())
    assert res[2][4]["int8"] == 1 and res[0][4]["int8"] == 0
    for wm in (0, 2):
        assert torch.equal(res[wm][1], a64)
        assert res[wm][2][1] == best64[1]
        rel = ((res[wm][0] - v64).abs() / v64.abs()).max()
        assert float(rel) <= 1e-7, (wm, float(rel))                     # measured ~1e-9: the enforced ceiling is 9e-7
        assert res[wm][3]["overflow"] == 0
    # the marker the certificate re-evaluated carries m^T S V S m in fp64 (m^T (S (V (S m))), not a value read off the int8 W)
    i = res[2][2][1]
    m = sh.Mt8[i, :sh.n].double()
    true = float((S.T @ m) @ (V @ (S @ m)))
    assert abs(float(res[2][0][i]) - true) <= 1e-12 * abs(true)
    assert abs(res[2][2][0] - res[0][2][0]) <= 1e-11 * abs(res[0][2][0])


def test_operands_the_configurations_cannot_certify_end_on_the_fp64_products():
    torch, sh = _shard(1200, 2048)
    S, V, ahat = _model_operands(torch, sh)
    L = sh.Lloc

    def scan(Sx, Vx, wm):
        sh.set_operands(Sx, Vx, ahat)
        sh.w_mode = wm
        sh.scan()
        torch.cuda.synchronize()
        return sh.vara[:L].clone(), sh.Wu.clone(), sh.w_info()

    # visibly asymmetric V: the measured || V - V^T ||_F makes the asymmetry term of the bound too large -> the fp64 general products
    gen = torch.Generator(device=sh.dev)
    gen.manual_seed(1)
    Vasym = V + 1e-7 * torch.triu(torch.randn(V.shape, generator=gen, device=sh.dev, dtype=torch.float64), 1)
    v8, W8, i8 = scan(S, Vasym, 2)
    v0, W0, _ = scan(S, Vasym, 0)
    assert i8["int8"] == 0 and i8["declined"] == 2 and torch.equal(W8, W0) and torch.equal(v8, v0), i8
    # a non-finite entry
    Vnan = V.clone()
    Vnan[3, 5] = float("nan")
    Vnan[5, 3] = float("nan")
    _, W8, i8 = scan(S, Vnan, 2)
    _, W0, _ = scan(S, Vnan, 0)
    assert i8["int8"] == 0 and i8["declined"] == 1
    assert torch.equal(torch.isnan(W8), torch.isnan(W0))
    # rounding-level asymmetry (what R's solve() leaves) is paid for by the bound, not declined
    Vr = V + 1e-17 * torch.triu(torch.randn(V.shape, generator=gen, device=sh.dev, dtype=torch.float64), 1)
    v8, W8, i8 = scan(S, Vr, 2)
    v0, W0, _ = scan(S, Vr, 0)
    assert i8["int8"] == 1 and 0.0 < i8["asym_term"] < 0.25 * i8["target"], i8
    assert _fold_err(torch, W8, W0) <= i8["eta"]
    # a V of mixed sign with large off-diagonal entries (nothing like a variance matrix): certified or declined, never wrong
    g2 = torch.randn((sh.n, 40), generator=gen, device=sh.dev, dtype=torch.float64)
    Vc = 0.3 * torch.eye(sh.n, dtype=torch.float64, device=sh.dev) + (g2[:, :20] @ g2[:, :20].T - g2[:, 20:] @ g2[:, 20:].T) / 40.0
    sh.mode = 0
    vref, _, _ = scan(S, Vc, 0)
    sh.mode = 1
    v8, W8, i8 = scan(S, Vc, 2)
    v0, W0, _ = scan(S, Vc, 0)
    if i8["int8"]:
        assert _fold_err(torch, W8, W0) <= i8["eta"]
    else:
        assert torch.equal(W8, W0)
    ok = vref.abs() > 1e-9 * vref.abs().max()
    assert float(((v8 - vref).abs() / vref.abs())[ok].max()) <= 9e-7


def test_an_overflowing_certificate_replaces_the_int8_w_by_the_fp64_products():
    """More than 2,048 markers outside what any digit count certifies (the panel of test_gpu_spectral): the block is redone in fp64 --
    on the fp64 W, which the library forms in place from the operands on record."""
    import torch
    from eagleeverything_amd.sharded import DeviceShard
    n, L = 2048, 6144
    rng = np.random.default_rng(3)
    Mt8 = (rng.binomial(2, rng.uniform(0.1, 0.5, size=L)[:, None], size=(L, n)) - 1).astype(np.int8)
    quiet = np.arange(n) < n // 2
    Mt8[np.ix_(np.arange(L) % 2 == 1, ~quiet)] = 0
    E = rng.standard_normal((n, n)) * 2e-4
    E[quiet, :] = 0.0
    E[:, quiet] = 0.0
    V = np.diag(np.where(quiet, 1e-6, 1.0) * rng.uniform(0.8, 1.2, size=n)) + 0.5 * (E + E.T)
    sh = DeviceShard(n, L)
    sh.Mt8[:L, :n] = torch.from_numpy(Mt8).to(sh.dev)
    sh.set_operands(0.9 * np.eye(n), V, rng.standard_normal(n))
    try:
        sh.L.eagle_dev_set_spectral(sh.ctx, 1)
        sh.mode = 0
        sh.scan()
        torch.cuda.synchronize()
        v64, best64 = sh.vara[:L].clone(), sh.best()[:2]
        sh.mode = 1
        sh.w_mode = 2
        sh.scan()
        torch.cuda.synchronize()
        c = sh.certificate()
        assert c["overflow"] == 1
        info = sh.w_info()
        assert info["int8"] == 0 and info["declined"] == 8, info
        assert torch.equal(sh.vara[:L], v64) and sh.best()[:2] == best64
    finally:
        sh.L.eagle_dev_set_spectral(sh.ctx, 1)


def test_structured_population_panel():
    """Three sub-populations (Balding-Nichols, Fst 0.1): MM^T has large leading eigenvalues, S and V a dense low-rank part."""
    torch, sh = _shard(1536, 8192)
    n, L, K, fst = sh.n, sh.Lloc, 3, 0.1
    gen = torch.Generator(device=sh.dev)
    gen.manual_seed(5)
    pop = torch.arange(n, device=sh.dev) * K // n
    p0 = 0.05 + 0.9 * torch.rand(L, 1, generator=gen, device=sh.dev)
    a, b = p0 * (1 - fst) / fst, (1 - p0) * (1 - fst) / fst
    ga = torch.distributions.Gamma(a.expand(-1, K), 1.0).sample()
    gb = torch.distributions.Gamma(b.expand(-1, K), 1.0).sample()
    p = (ga / (ga + gb)).clamp(0.001, 0.999)[:, pop]
    g = (torch.rand(p.shape, generator=gen, device=sh.dev) < p).to(torch.int8) + (torch.rand(p.shape, generator=gen, device=sh.dev) < p).to(torch.int8) - 1
    sh.Mt8[:L, :n] = g
    sh.M8 = sh.M4 = sh.Mt8s = None
    S, V, ahat = _model_operands(torch, sh)
    sh.set_operands(S, V, ahat)
    sh.mode = 0
    sh.scan()
    torch.cuda.synchronize()
    v64, best64 = sh.vara[:L].clone(), sh.best()[:2]
    sh.mode = 1
    sh.w_mode = 0
    sh.scan_operands()
    W0 = sh.Wu.clone()
    sh.w_mode = 2
    sh.scan()
    torch.cuda.synchronize()
    info = sh.w_info()
    assert info["int8"] == 1, info
    sh.scan_operands()
    torch.cuda.synchronize()
    assert _fold_err(torch, sh.Wu, W0) <= info["eta"]
    sh.scan()
    torch.cuda.synchronize()
    ok = v64.abs() > 0
    assert float(((sh.vara[:L] - v64).abs() / v64.abs())[ok].max()) <= 9e-7
    assert sh.best()[1] == best64[1]


@pytest.mark.parametrize("n,L", [(700, 3000), (3300, 1024)])   # one block of V's rows; three blocks of 1,536 (the pipelined upload)
def test_reference_shaped_call_on_the_int8_w(tmp_path, n, L):
    """Through the C ABI of the Rcpp surface, resident and streamed in marker blocks: the oracle's marker, vara inside 1e-7 of it, the
    same bits from both paths -- and from a call that forms V S block by block while V is still arriving."""
    from eagleeverything_amd import rcpp_api as api, synth
    from oracle import oracle_c
    oracle_c.build()
    Mt8 = synth.genotypes_marker_major(n, L, seed=4)
    rng = np.random.default_rng(2)
    A = rng.standard_normal((n, 30)) / np.sqrt(n)
    S = np.eye(n) * 0.8 + A @ A.T
    S = 0.5 * (S + S.T)
    B = rng.standard_normal((n, n)) * 1e-3
    V = 0.5 * np.eye(n) - 0.05 * (A[:, :5] @ A[:, :5].T) + 0.5 * (B + B.T)
    ahat = rng.standard_normal(n)
    geno = synth.write_geno_pair(str(tmp_path), Mt8)
    ref = oracle_c.calculate_a_and_vara_rcpp(geno["asciifileMt"], np.nan, S, V, 8.0, (L, n), ahat)
    try:
        api.set_scan_mode(1)
        api.set_w_mode(2)
        r = api.calculate_a_and_vara_rcpp(geno["asciifileMt"], np.nan, S, V, 8.0, (L, n), ahat)
        info = api.last_w_info()
        assert info["int8"] == 1, info
        np.testing.assert_allclose(r["a"], ref["a"], rtol=1e-9, atol=1e-12 * np.abs(ref["a"]).max())
        np.testing.assert_allclose(r["vara"], ref["vara"], rtol=1e-7)
        assert api.last_scan_argmax()[0] == oracle_c.tsq_argmax(ref["a"], ref["vara"])[1]
        # the second call of a context forms V S block by block while V arrives (on the first call's configuration): the same bits
        rp = api.calculate_a_and_vara_rcpp(geno["asciifileMt"], np.nan, S, V, 8.0, (L, n), ahat)
        assert api.last_w_info()["pipelined"] == 1 and info["pipelined"] == 0
        assert np.array_equal(rp["vara"], r["vara"]) and np.array_equal(rp["a"], r["a"])
        # ... and a V that calls for another configuration is not served by the guess: a stronger off-diagonal part, new statistics
        V2 = V + 0.05 * (B + B.T)
        rq = api.calculate_a_and_vara_rcpp(geno["asciifileMt"], np.nan, S, V2, 8.0, (L, n), ahat)
        iq = api.last_w_info()
        refq = oracle_c.calculate_a_and_vara_rcpp(geno["asciifileMt"], np.nan, S, V2, 8.0, (L, n), ahat)
        np.testing.assert_allclose(rq["vara"], refq["vara"], rtol=1e-7)
        api.set_w_mode(0)
        r0q = api.calculate_a_and_vara_rcpp(geno["asciifileMt"], np.nan, S, V2, 8.0, (L, n), ahat)
        api.set_w_mode(2)
        np.testing.assert_allclose(rq["vara"], r0q["vara"], rtol=1e-7)
        assert iq["int8"] == 1
        os.environ["EAGLE_HIP_MAX_RESIDENT_GB"] = "%.6f" % (1.6 * ((n + 255) // 256 * 256) * 768 / 1e9)   # marker blocks of 768: the file is streamed
        api.drop_cache()
        r2 = api.calculate_a_and_vara_rcpp(geno["asciifileMt"], np.nan, S, V, 8.0, (L, n), ahat)
        assert api.last_w_info()["int8"] == 1
        assert np.array_equal(r2["vara"], r["vara"]) and np.array_equal(r2["a"], r["a"])
        api.set_w_mode(0)
        r0 = api.calculate_a_and_vara_rcpp(geno["asciifileMt"], np.nan, S, V, 8.0, (L, n), ahat)
        np.testing.assert_allclose(r0["vara"], r["vara"], rtol=1e-7)
    finally:
        os.environ.pop("EAGLE_HIP_MAX_RESIDENT_GB", None)
        api.set_w_mode(1)
        api.drop_cache()
