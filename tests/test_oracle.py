"""CPU: the C oracle against the golden vectors (known MM^T answers from the reference's demo data,
numpy/OpenBLAS restatement for fp64 outputs) and against the reference's branch rules.

parity unpinned for fp64 outputs: the reference records no expected outputs.
"""
import numpy as np
import pytest

from conftest import GOLDEN_CASES
from eagleeverything_amd import synth

NA = np.nan


def _files(tmp_path, g):
    Mt8 = np.ascontiguousarray(g["M8"].T)
    return synth.write_geno_pair(str(tmp_path), Mt8)


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_readblock_matches_fixture(case, golden, oracle, tmp_path):
    g = golden(case)
    geno = _files(tmp_path, g)
    n, L = g["M8"].shape
    M = oracle.ReadBlock(geno["asciifileM"], 0, L, n)
    assert M.shape == (n, L) and M.flags.f_contiguous
    np.testing.assert_array_equal(M, g["M8"].astype(np.float64))
    # a block in the middle, ragged sizes
    blk = oracle.ReadBlock(geno["asciifileMt"], 7, n - 5, 11)
    np.testing.assert_array_equal(blk, g["M8"].T[7:18, : n - 5].astype(np.float64))


def test_readblock_errors(oracle, tmp_path):
    with pytest.raises(oracle.OracleError, match="Could not open"):
        oracle.ReadBlock(str(tmp_path / "missing.ascii"), 0, 3, 3)
    p = synth.write_ascii(str(tmp_path / "s.ascii"), np.zeros((4, 6), np.int8))
    with pytest.raises(oracle.OracleError):
        oracle.ReadBlock(p, 2, 6, 5)  # fewer lines than asked
    with pytest.raises(oracle.OracleError):
        oracle.ReadBlock(p, 0, 9, 2)  # lines shorter than numcols


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_mmt_known_answers_inmemory_and_blocked(case, golden, oracle, tmp_path):
    g = golden(case)
    geno = _files(tmp_path, g)
    n, L = g["M8"].shape
    mmt, br = oracle.calculateMMt_rcpp(geno["asciifileM"], 8.0, 2, NA, (n, L), return_branch=True)
    assert br == 0
    np.testing.assert_array_equal(mmt, g["MMt"].astype(np.float64))
    # force the row-block branch (calculateMMt_rcpp.cpp:99-174): tiny availmemGb
    need = (n * n * 8 + 2 * n * L * 8) / 1e9
    mem = need / 6.0
    mmt_b, br = oracle.calculateMMt_rcpp(geno["asciifileM"], mem, 2, NA, (n, L), return_branch=True)
    assert 0 < br < n
    np.testing.assert_array_equal(mmt_b, g["MMt"].astype(np.float64))


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_mmt_masking_rule(case, golden, oracle, tmp_path):
    g = golden(case)
    geno = _files(tmp_path, g)
    n, L = g["M8"].shape
    sel = g["sel_masked"]
    m = oracle.calculateMMt_rcpp(geno["asciifileM"], 8.0, 1, sel, (n, L))
    np.testing.assert_array_equal(m, g["MMt_masked"].astype(np.float64))
    # element 0 NA => no masking at all, whatever follows (calculateMMt_rcpp.cpp:88; AM.R:260,455)
    m2 = oracle.calculateMMt_rcpp(geno["asciifileM"], 8.0, 1, np.array([NA, 3.0, 17.0]), (n, L))
    np.testing.assert_array_equal(m2, g["MMt"].astype(np.float64))
    need = (n * n * 8 + 2 * n * L * 8) / 1e9
    m3 = oracle.calculateMMt_rcpp(geno["asciifileM"], need / 5.0, 1, sel, (n, L))
    np.testing.assert_array_equal(m3, g["MMt_masked"].astype(np.float64))


def test_normalise(golden, oracle):
    g = golden("geno_150x100")
    out = oracle.normalise_MMt(g["MMt"].astype(np.float64))
    exp = g["MMt"] / g["MMt"].max() + 0.95 * np.eye(150)
    np.testing.assert_allclose(out, exp, rtol=0, atol=1e-15)


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_scan_matches_numpy_restatement(case, golden, oracle, tmp_path):
    g = golden(case)
    geno = _files(tmp_path, g)
    n, L = g["M8"].shape
    res, br = oracle.calculate_a_and_vara_rcpp(geno["asciifileMt"], NA, g["S"], g["V"], 8.0, (L, n), g["ahat"],
                                               return_branch=True)
    assert br == 0
    assert res["a"].shape == (L, 1) and res["vara"].shape == (L, 1)
    np.testing.assert_allclose(res["a"].ravel(), g["a"], rtol=1e-11, atol=1e-13 * np.abs(g["a"]).max())
    np.testing.assert_allclose(res["vara"].ravel(), g["vara"], rtol=1e-10)
    tsq, idx, mx = oracle.tsq_argmax(res["a"], res["vara"])
    assert idx == int(g["argmax"])
    np.testing.assert_allclose(mx, float(g["tsqmax"]), rtol=1e-9)
    # masked
    resm = oracle.calculate_a_and_vara_rcpp(geno["asciifileMt"], g["sel_masked"], g["S"], g["V"], 8.0, (L, n),
                                            g["ahat"])
    np.testing.assert_allclose(resm["vara"].ravel(), g["vara_masked"], rtol=1e-10, atol=0)
    for s in g["sel_masked"].astype(int):
        assert resm["a"][s, 0] == 0.0 and resm["vara"][s, 0] == 0.0
    tsq, idx, _ = oracle.tsq_argmax(resm["a"], resm["vara"])
    assert np.isnan(tsq[int(g["sel_masked"][0])])  # 0/0 ignored by na.rm=TRUE (find_qtl.R:76)
    assert idx == int(g["argmax_masked"])


def test_scan_blocked_branch_equals_inmemory(golden, oracle, tmp_path):
    g = golden("synth_203x1531")
    geno = _files(tmp_path, g)
    n, L = g["M8"].shape
    full = oracle.calculate_a_and_vara_rcpp(geno["asciifileMt"], NA, g["S"], g["V"], 8.0, (L, n), g["ahat"])
    # 4*n*L*8 bytes = 9.9e6 < 1e9 => integer division gives 0 GB needed; block branch needs availmemGb <= 0?
    # calculate_a_and_vara_rcpp.cpp:74 uses strict '<', so availmemGb = 0 forces blocks of 0 rows (error);
    # a small positive value stays in-memory.  Use a larger problem estimate through a tiny negative test below.
    res, br = oracle.calculate_a_and_vara_rcpp(geno["asciifileMt"], NA, g["S"], g["V"], 1e-9, (L, n), g["ahat"],
                                               return_branch=True)
    assert br == 0  # 0 < 1e-9: still the in-memory branch
    np.testing.assert_array_equal(res["a"], full["a"])
    with pytest.raises(oracle.OracleError, match="zero rows"):
        oracle.calculate_a_and_vara_rcpp(geno["asciifileMt"], NA, g["S"], g["V"], 0.0, (L, n), g["ahat"])
    neg = oracle.calculate_a_and_vara_rcpp(geno["asciifileMt"], NA, g["S"], g["V"], -1.0, (L, n), g["ahat"])
    assert neg["a"].shape == (1,) and neg["a"][0] == 0 and neg["vara"][0] == 0  # sentinel :141-142


def test_reduced_a(golden, oracle, tmp_path):
    g = golden("synth_203x1531")
    geno = _files(tmp_path, g)
    n, L = g["M8"].shape
    ar = oracle.calculate_reduced_a_rcpp(geno["asciifileMt"], float(g["varG"]), g["P"], g["y"], 8.0, (n, L), NA)
    assert ar.shape == (L, 1)
    np.testing.assert_allclose(ar.ravel(), g["ar"], rtol=1e-10, atol=1e-12 * np.abs(g["ar"]).max())
    z = oracle.calculate_reduced_a_rcpp(geno["asciifileMt"], float(g["varG"]), g["P"], g["y"], 0.0, (n, L), NA)
    assert z.shape == (1, 1) and z[0, 0] == 0.0  # calculate_reduced_a_rcpp.cpp:94-103
    arm = oracle.calculate_reduced_a_rcpp(geno["asciifileMt"], float(g["varG"]), g["P"], g["y"], 8.0, (n, L),
                                          np.array([5.0, 9.0]))
    assert arm[5, 0] == 0.0 and arm[9, 0] == 0.0
    np.testing.assert_allclose(np.delete(arm.ravel(), [5, 9]), np.delete(g["ar"], [5, 9]), rtol=1e-10,
                               atol=1e-12 * np.abs(g["ar"]).max())


def test_tsq_argmax_first_of_ties_and_nan(oracle):
    a = np.array([1.0, 2.0, 0.0, 2.0, 0.0])
    v = np.array([1.0, 1.0, 0.0, 1.0, 1.0])
    tsq, idx, mx = oracle.tsq_argmax(a, v)
    assert np.isnan(tsq[2]) and idx == 2 and mx == 4.0  # first of the tied maxima, NaN skipped
    tsq, idx, mx = oracle.tsq_argmax(np.zeros(3), np.zeros(3))
    assert idx == 0 and np.isnan(mx)


def test_inmem_i8_entry_points(golden, oracle):
    g = golden("synth_203x1531")
    Mt8 = np.ascontiguousarray(g["M8"].T)
    a, vara = oracle.scan_from_i8(Mt8, g["S"], g["V"], g["ahat"])
    np.testing.assert_allclose(vara, g["vara"], rtol=1e-10)
    np.testing.assert_allclose(a, g["a"], rtol=1e-10, atol=1e-12)
    mmt = oracle.mmt_from_i8(g["M8"])
    np.testing.assert_array_equal(mmt, g["MMt"].astype(np.float64))


def test_scan_real_blocked_branch(oracle, tmp_path):
    """A problem big enough (4*n*L*8 >= 1e9) for calculate_a_and_vara_rcpp.cpp:74 to take the marker-block branch
    (:117-234), including block-relative masking (:176-190); it must reproduce the in-memory branch exactly."""
    n, L = 48, 700000
    rng = np.random.default_rng(1)
    Mt8 = rng.integers(-1, 2, size=(L, n), dtype=np.int8)
    p = synth.write_ascii(str(tmp_path / "Mt.ascii"), Mt8)
    A = rng.standard_normal((n, n)) / 6.0
    S = A @ A.T + np.eye(n)
    V = 0.3 * np.eye(n)
    ah = rng.standard_normal(n)
    sel = np.array([5.0, 233334.0, 466669.0, float(L - 1)])  # one in every block
    full, br0 = oracle.calculate_a_and_vara_rcpp(p, sel, S, V, 8.0, (L, n), ah, return_branch=True)
    blk, br1 = oracle.calculate_a_and_vara_rcpp(p, sel, S, V, 0.5, (L, n), ah, return_branch=True)
    assert br0 == 0 and 0 < br1 < L  # 0.5e9/(4*48*8) = 325520 rows per block -> 3 blocks
    np.testing.assert_array_equal(blk["a"], full["a"])
    np.testing.assert_array_equal(blk["vara"], full["vara"])
    for s in sel.astype(int):
        assert blk["a"][s, 0] == 0.0 and blk["vara"][s, 0] == 0.0
    a_np = Mt8.astype(np.float64) @ (S @ ah)
    a_np[sel.astype(int)] = 0.0
    np.testing.assert_allclose(blk["a"].ravel(), a_np, rtol=1e-11, atol=1e-12)


def test_extract_geno(golden, oracle, tmp_path):
    g = golden("synth_203x1531")
    geno = _files(tmp_path, g)
    n, L = g["M8"].shape
    for c in (0, 17, L - 1):
        np.testing.assert_array_equal(oracle.extract_geno_rcpp(geno["asciifileM"], 8.0, c, (n, L)), g["M8"][:, c].astype(np.int32))


def test_scan_against_exact_rational_arithmetic(oracle):
    """The mathematical definition, with no rounding at all: a = Mt (S a_hat), vara_i = m_i^T (S (V S)) m_i
    (calculate_a_and_vara_rcpp.cpp:90-112) and MM^T, evaluated in exact rational arithmetic (Python fractions) from the very
    doubles the oracle is given.  What this pins is that the C restatement (and the numpy one the golden vectors come from)
    computes these quantities to fp64 rounding -- not the reference's own rounding, which no file records (parity unpinned)."""
    from fractions import Fraction as F
    from oracle import oracle_np
    rng = np.random.default_rng(11)
    n, L = 7, 9
    Mt8 = rng.integers(-1, 2, size=(L, n)).astype(np.int8)
    A = rng.standard_normal((n, n))
    S = A @ A.T / n + np.eye(n)
    B = rng.standard_normal((n, n))
    V = 0.5 * (B + B.T) / n + 2.0 * np.eye(n)
    ahat = rng.standard_normal(n)
    Sq = [[F(float(x)) for x in row] for row in S]
    Vq = [[F(float(x)) for x in row] for row in V]
    aq = [F(float(x)) for x in ahat]
    Mq = [[int(x) for x in row] for row in Mt8]
    mat = lambda X, Y: [[sum(X[i][k] * Y[k][j] for k in range(n)) for j in range(n)] for i in range(n)]
    Wq = mat(Sq, mat(Vq, Sq))
    vq = [sum(Sq[i][j] * aq[j] for j in range(n)) for i in range(n)]
    a_exact = [sum(Mq[i][j] * vq[j] for j in range(n)) for i in range(L)]
    vara_exact = [sum(Mq[i][j] * Wq[j][k] * Mq[i][k] for j in range(n) for k in range(n)) for i in range(L)]
    for name, (a, vara) in (("C oracle", oracle.scan_from_i8(Mt8, S, V, ahat)), ("numpy", oracle_np.a_and_vara(Mt8, S, V, ahat))):
        a, vara = np.ravel(a), np.ravel(vara)
        scale_a = max(abs(float(x)) for x in a_exact)
        for i in range(L):
            assert abs(F(float(a[i])) - a_exact[i]) <= F(1, 10 ** 14) * F(scale_a), (name, "a", i)
            assert abs(F(float(vara[i])) - vara_exact[i]) <= F(1, 10 ** 13) * abs(vara_exact[i]) + F(1, 10 ** 300), (name, "vara", i)
    G = Mt8.astype(np.int64)
    mmt_exact = [[sum(Mq[l][i] * Mq[l][j] for l in range(L)) for j in range(n)] for i in range(n)]
    assert np.array_equal(oracle_np.mmt_int64(np.ascontiguousarray(Mt8.T)), np.array(mmt_exact, dtype=np.int64))
    assert np.array_equal(G.T @ G, np.array(mmt_exact, dtype=np.int64))
