"""GPU parity tests: every call goes through the C ABI of libeaglehip.so (rcpp_api mirrors the reference's
exported functions) and is compared with the CPU oracle / the golden vectors on the same inputs.

Bars:  MM^T and ReadBlock bit-exact (integer data) ; a, vara, tsq within 1e-9 relative (north_star allows
1e-6; the fp64 kernels only differ from the oracle in summation order) ; selected index identical.
parity unpinned for fp64 outputs (the reference records none; see oracle/eagle_oracle.c).
"""
import numpy as np
import pytest

from conftest import GOLDEN_CASES
from eagleeverything_amd import synth

pytestmark = pytest.mark.gpu
NA = np.nan
RTOL = 1e-9        # a, and vara from the fp64 MFMA kernel (mode 0)
RTOL_DIGITS = 1e-7  # vara from the digit-slice kernels against the oracle: the level actually MEASURED on these cases (1e-8 and below), so that
# an order of magnitude of drift shows; what the certificate ENFORCES per marker is RTOL_ENFORCED = 1.8 x the budget in force (9e-7 at most; north_star: 1e-6)
RTOL_ENFORCED = 9e-7


@pytest.fixture(scope="module")
def api():
    from eagleeverything_amd import rcpp_api
    info = rcpp_api.device_info()
    assert info["arch"].startswith("gfx950")
    yield rcpp_api
    rcpp_api.close_all()


@pytest.fixture(scope="module")
def files(tmp_path_factory, golden):
    out = {}
    for case in GOLDEN_CASES:
        g = golden(case)
        d = tmp_path_factory.mktemp(case)
        out[case] = (g, synth.write_geno_pair(str(d), np.ascontiguousarray(g["M8"].T)))
    return out


def _r_which_max(a, vara):
    """which(tsq == max(tsq, na.rm=TRUE))[1] of find_qtl.R:71-83, 1-based (0: every tsq NaN)."""
    with np.errstate(all="ignore"):
        tsq = np.ravel(a) ** 2 / np.ravel(vara)
    if np.all(np.isnan(tsq)):
        return 0
    return int(np.flatnonzero(tsq == np.nanmax(tsq))[0]) + 1


def _selection_rests_on_noise(ref, idx_ref):
    """True when the reference's own pick is a (nearly) constant marker: its vara = c^2 1'W1 is rounding noise of the
    reference's summation order (tsq = a^2 / noise), which no other order can reproduce."""
    if idx_ref <= 0:
        return True
    v = np.abs(np.ravel(ref["vara"]))
    return bool(v[idx_ref - 1] <= 1e-8 * np.median(v))


def _close(x, ref, rtol=RTOL, scale=None):
    x = np.ravel(x); ref = np.ravel(ref)
    s = np.abs(ref).max() if scale is None else scale
    np.testing.assert_allclose(x, ref, rtol=rtol, atol=1e-12 * s)


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_readblock_bit_exact(case, files, api, oracle):
    g, geno = files[case]
    n, L = g["M8"].shape
    M = api.ReadBlock(geno["asciifileM"], 0, L, n)
    assert M.shape == (n, L) and M.flags.f_contiguous
    np.testing.assert_array_equal(M, g["M8"].astype(np.float64))
    for (start, cols, rows) in [(7, n - 5, 11), (0, 1, 1), (L - 3, n, 3), (5, min(64, n), min(128, L - 5))]:
        blk = api.ReadBlock(geno["asciifileMt"], start, cols, rows)
        np.testing.assert_array_equal(blk, oracle.ReadBlock(geno["asciifileMt"], start, cols, rows))


def test_readblock_errors(api, tmp_path):
    from eagleeverything_amd._lib import EagleError
    with pytest.raises(EagleError, match="Could not open"):
        api.ReadBlock(str(tmp_path / "missing.ascii"), 0, 3, 3)
    p = synth.write_ascii(str(tmp_path / "s.ascii"), np.zeros((4, 6), np.int8))
    with pytest.raises(EagleError):
        api.ReadBlock(p, 2, 6, 5)
    with pytest.raises(EagleError):
        api.ReadBlock(p, 0, 9, 2)
    assert api.ReadBlock(p, 1, 0, 0).shape == (0, 0)
    # a file that is not fixed-width is still read line by line (ReadBlock uses getline)
    q = tmp_path / "ragged.ascii"
    q.write_text("0120\n21012\n000\n")
    np.testing.assert_array_equal(api.ReadBlock(str(q), 0, 3, 3), np.array([[-1, 0, 1], [1, 0, -1], [-1, -1, -1.0]]))
    # characters outside '0'..'2' are rejected (the reference would silently produce other integers)
    r = tmp_path / "bad.ascii"
    r.write_text("01a\n012\n")
    with pytest.raises(EagleError, match="outside"):
        api.ReadBlock(str(r), 0, 3, 2)


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_mmt_bit_exact_and_masking_rule(case, files, api):
    g, geno = files[case]
    n, L = g["M8"].shape
    msgs = []
    mmt = api.calculateMMt_rcpp(geno["asciifileM"], 8.0, 2, NA, (n, L), quiet=False, message=msgs.append)
    assert mmt.shape == (n, n)
    np.testing.assert_array_equal(mmt, g["MMt"].astype(np.float64))
    assert any("Number of cores" in m for m in msgs)
    # tiny availmemGb: the reference would take its row-block branch; results are identical
    need = (n * n * 8 + 2 * n * L * 8) / 1e9
    np.testing.assert_array_equal(api.calculateMMt_rcpp(geno["asciifileM"], need / 6, 2, NA, (n, L)), mmt)
    # masking fires iff element 0 is not NA
    np.testing.assert_array_equal(api.calculateMMt_rcpp(geno["asciifileM"], 8.0, 1, g["sel_masked"], (n, L)),
                                  g["MMt_masked"].astype(np.float64))
    np.testing.assert_array_equal(api.calculateMMt_rcpp(geno["asciifileM"], 8.0, 1, np.array([NA, 3.0, 17.0]), (n, L)), mmt)
    dup = np.concatenate([g["sel_masked"], g["sel_masked"][:1]])  # zeroing a column twice = once
    np.testing.assert_array_equal(api.calculateMMt_rcpp(geno["asciifileM"], 8.0, 1, dup, (n, L)),
                                  g["MMt_masked"].astype(np.float64))
    # calcMMt.R:13 on the device
    from eagleeverything_amd import r_api
    norm = r_api.calcMMt(geno, 8.0, 2, NA, True)
    exp = g["MMt"] / g["MMt"].max() + 0.95 * np.eye(n)
    np.testing.assert_allclose(norm, exp, rtol=0, atol=2e-16)


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_scan_matches_oracle(case, mode, files, api, oracle, request):
    g, geno = files[case]
    n, L = g["M8"].shape
    request.addfinalizer(lambda: api.set_scan_mode(1))
    api.set_scan_mode(mode)
    ref = oracle.calculate_a_and_vara_rcpp(geno["asciifileMt"], NA, g["S"], g["V"], 8.0, (L, n), g["ahat"])
    res = api.calculate_a_and_vara_rcpp(geno["asciifileMt"], NA, g["S"], g["V"], 8.0, (L, n), g["ahat"])
    assert res["a"].shape == (L, 1) and res["vara"].shape == (L, 1)
    _close(res["a"], ref["a"])
    # markers that are constant over individuals have vara == 0 up to rounding noise (X holds the intercept):
    # an absolute floor of 1e-12 * max|vara| keeps those out of the relative comparison
    vs = np.abs(ref["vara"]).max()
    rt = RTOL_DIGITS if mode == 1 else RTOL
    np.testing.assert_allclose(res["vara"].ravel(), ref["vara"].ravel(), rtol=rt, atol=1e-12 * vs)
    np.testing.assert_allclose(res["vara"].ravel(), g["vara"], rtol=rt, atol=1e-12 * vs)  # golden (numpy restatement)
    tsq_ref, idx_ref, mx_ref = oracle.tsq_argmax(ref["a"], ref["vara"])
    idx, mx, near = api.last_scan_argmax()
    assert idx == idx_ref == int(g["argmax"])
    np.testing.assert_allclose(mx, mx_ref, rtol=2 * rt)
    # identical marker columns (perfect LD) tie exactly; the first one is selected (find_qtl.R:80)
    assert near == int(np.sum(tsq_ref >= mx_ref * (1 - 1e-9)))
    # masked rows: a = vara = 0 there, tsq NaN there and ignored by the arg-max
    resm = api.calculate_a_and_vara_rcpp(geno["asciifileMt"], g["sel_masked"], g["S"], g["V"], 8.0, (L, n), g["ahat"])
    np.testing.assert_allclose(resm["vara"].ravel(), g["vara_masked"], rtol=rt, atol=1e-12 * vs)
    for s in g["sel_masked"].astype(int):
        assert resm["a"][s, 0] == 0.0 and resm["vara"][s, 0] == 0.0
    idx, _, _ = api.last_scan_argmax()
    assert idx == int(g["argmax_masked"])
    api.set_scan_mode(1)


def test_scan_branch_rules_and_sentinels(files, api):
    from eagleeverything_amd._lib import EagleError
    g, geno = files["synth_203x1531"]
    n, L = g["M8"].shape
    with pytest.raises(EagleError, match="zero rows"):
        api.calculate_a_and_vara_rcpp(geno["asciifileMt"], NA, g["S"], g["V"], 0.0, (L, n), g["ahat"])
    neg = api.calculate_a_and_vara_rcpp(geno["asciifileMt"], NA, g["S"], g["V"], -1.0, (L, n), g["ahat"])
    assert neg["a"].shape == (1,) and neg["a"][0] == 0 and neg["vara"][0] == 0
    with pytest.raises(EagleError, match="Could not open"):
        api.calculate_a_and_vara_rcpp(geno["asciifileMt"] + ".nope", NA, g["S"], g["V"], 8.0, (L, n), g["ahat"])
    with pytest.raises(EagleError):
        api.calculate_a_and_vara_rcpp(geno["asciifileMt"], np.array([5.0, NA]), g["S"], g["V"], 8.0, (L, n), g["ahat"])


def test_reduced_a(files, api, oracle):
    g, geno = files["synth_203x1531"]
    n, L = g["M8"].shape
    ref = oracle.calculate_reduced_a_rcpp(geno["asciifileMt"], float(g["varG"]), g["P"], g["y"], 8.0, (n, L), NA)
    ar = api.calculate_reduced_a_rcpp(geno["asciifileMt"], float(g["varG"]), g["P"], g["y"], 8.0, (n, L), NA)
    assert ar.shape == (L, 1)
    _close(ar, ref)
    z = api.calculate_reduced_a_rcpp(geno["asciifileMt"], float(g["varG"]), g["P"], g["y"], 0.0, (n, L), NA)
    assert z.shape == (1, 1) and z[0, 0] == 0.0
    arm = api.calculate_reduced_a_rcpp(geno["asciifileMt"], float(g["varG"]), g["P"], g["y"], 8.0, (n, L), np.array([5.0, 9.0]))
    assert arm[5, 0] == 0.0 and arm[9, 0] == 0.0
    _close(np.delete(arm.ravel(), [5, 9]), np.delete(ref.ravel(), [5, 9]), scale=np.abs(ref).max())


def test_extract_geno_resident_and_from_file(files, api, oracle):
    g, geno = files["synth_203x1531"]
    n, L = g["M8"].shape
    api.drop_cache()
    for c in (0, 17, L - 1):  # not resident: one character per line from the file
        np.testing.assert_array_equal(api.extract_geno_rcpp(geno["asciifileM"], 8.0, c, (n, L)),
                                      oracle.extract_geno_rcpp(geno["asciifileM"], 8.0, c, (n, L)))
    api.calculateMMt_rcpp(geno["asciifileM"], 8.0, 2, NA, (n, L))  # makes M.ascii resident in HBM
    for c in (0, 17, L - 1):
        np.testing.assert_array_equal(api.extract_geno_rcpp(geno["asciifileM"], 8.0, c, (n, L)), g["M8"][:, c].astype(np.int32))
    from eagleeverything_amd._lib import EagleError
    with pytest.raises(EagleError):
        api.extract_geno_rcpp(geno["asciifileM"], 8.0, L, (n, L))


def test_find_qtl_mirror_selects_same_marker(files, api, oracle):
    from eagleeverything_amd import host_model, r_api
    g, geno = files["genoDemo_150x4998"]
    n, L = g["M8"].shape
    MMt = r_api.calcMMt(geno, 8.0, 2, NA, True)
    invMMt = np.linalg.inv(MMt)
    sel = np.array([NA])  # AM.R:260: selected_loci starts as NA, so masking never fires
    idx, st = r_api.find_qtl(geno, 8.0, sel, MMt, invMMt, float(g["varE"]), float(g["varG"]), g["X"], 2, True, g["y"],
                             return_stats=True)
    assert idx == int(g["argmax"])
    np.testing.assert_allclose(st["tsqmax"], float(g["tsqmax"]), rtol=RTOL_DIGITS)


# ---- larger seeded case, ragged sizes, cache reuse, cross-check by properties ---------------------------
@pytest.fixture(scope="module")
def big(tmp_path_factory):
    n, L = 1003, 20011
    Mt8 = synth.genotypes_marker_major(n, L, seed=5)
    d = tmp_path_factory.mktemp("big")
    geno = synth.write_geno_pair(str(d), Mt8)
    rng = np.random.default_rng(3)
    A = rng.standard_normal((n, n)) / np.sqrt(n)
    S = A @ A.T + np.eye(n)            # SPD like MMt^-1/2
    Bm = rng.standard_normal((n, 8))
    V = 0.7 * np.eye(n) - 0.01 * (Bm @ Bm.T)
    ahat = rng.standard_normal(n)
    return Mt8, geno, S, V, ahat


def test_big_mmt_exact(big, api):
    Mt8, geno, *_ = big
    L, n = Mt8.shape
    mmt = api.calculateMMt_rcpp(geno["asciifileM"], 8.0, 4, NA, (n, L))
    G = Mt8.astype(np.float64)
    np.testing.assert_array_equal(mmt, G.T @ G)
    assert np.array_equal(mmt, mmt.T)


def test_big_scan(big, api, oracle):
    Mt8, geno, S, V, ahat = big
    L, n = Mt8.shape
    res = api.calculate_a_and_vara_rcpp(geno["asciifileMt"], NA, S, V, 8.0, (L, n), ahat)
    a_ref, v_ref = oracle.scan_from_i8(Mt8, S, V, ahat)
    _close(res["a"], a_ref)
    np.testing.assert_allclose(res["vara"].ravel(), v_ref, rtol=RTOL_DIGITS)
    # second call hits the HBM-resident tile and is bit-identical (deterministic reduction order)
    res2 = api.calculate_a_and_vara_rcpp(geno["asciifileMt"], NA, S, V, 8.0, (L, n), ahat)
    np.testing.assert_array_equal(res["a"], res2["a"])
    np.testing.assert_array_equal(res["vara"], res2["vara"])
    # linearity of a in a_hat, quadratic scaling of vara in S
    res3 = api.calculate_a_and_vara_rcpp(geno["asciifileMt"], NA, 2.0 * S, V, 8.0, (L, n), ahat)
    np.testing.assert_allclose(res3["a"], 2.0 * res["a"], rtol=1e-12)
    np.testing.assert_allclose(res3["vara"], 4.0 * res["vara"], rtol=1e-12)
    idx, mx, near = api.last_scan_argmax()
    tsq, idx_ref, mx_ref = oracle.tsq_argmax(a_ref * 2, v_ref * 4)
    assert idx == idx_ref


def test_big_scan_int8_slices_match_fp64_kernel(big, api, oracle):
    """The int8-slice vara kernel (exact integer partial sums) against the fp64 MFMA kernel and the oracle."""
    Mt8, geno, S, V, ahat = big
    L, n = Mt8.shape
    a_ref, v_ref = oracle.scan_from_i8(Mt8, S, V, ahat)
    api.set_scan_mode(0)
    r0 = api.calculate_a_and_vara_rcpp(geno["asciifileMt"], NA, S, V, 8.0, (L, n), ahat)
    api.set_scan_mode(1)
    r1 = api.calculate_a_and_vara_rcpp(geno["asciifileMt"], NA, S, V, 8.0, (L, n), ahat)
    r1b = api.calculate_a_and_vara_rcpp(geno["asciifileMt"], NA, S, V, 8.0, (L, n), ahat)
    idx1, mx1, _ = api.last_scan_argmax()
    # fewer digits: the documented bound (sum|m|)^2 * 2^(e+1-8S) must hold (e from the off-diagonal fold of W)
    Wexact = S @ (V @ S)
    off = np.triu(Wexact + Wexact.T, 1)
    e = int(np.floor(np.log2(np.abs(off).max()))) + 1
    for S_ in (3, 4, 5, 6):
        api.set_scan_slices(S_)
        rs = api.calculate_a_and_vara_rcpp(geno["asciifileMt"], NA, S, V, 8.0, (L, n), ahat)
        bound = (np.abs(Mt8).sum(axis=1).astype(np.float64) ** 2) * 2.0 ** (e + 1 - 8 * S_)
        assert np.all(np.abs(rs["vara"].ravel() - v_ref) <= bound + 1e-9 * np.abs(v_ref)), S_
    api.set_scan_slices(0)  # automatic choice must stay inside the tolerance by a wide margin
    ra = api.calculate_a_and_vara_rcpp(geno["asciifileMt"], NA, S, V, 8.0, (L, n), ahat)
    np.testing.assert_allclose(ra["vara"].ravel(), v_ref, rtol=RTOL_DIGITS)
    api.set_scan_slices(0)
    np.testing.assert_array_equal(r1["vara"], r1b["vara"])  # integer atomics: bitwise reproducible
    np.testing.assert_allclose(r0["vara"].ravel(), v_ref, rtol=RTOL)
    np.testing.assert_allclose(r1["vara"].ravel(), v_ref, rtol=RTOL_DIGITS)
    np.testing.assert_allclose(r1["vara"].ravel(), r0["vara"].ravel(), rtol=RTOL_DIGITS)
    np.testing.assert_array_equal(r1["a"], r0["a"])
    _, idx_ref, _ = oracle.tsq_argmax(a_ref, v_ref)
    assert idx1 == idx_ref


@pytest.mark.parametrize("mode", [0, 1])
def test_scan_nonsymmetric_operands(big, api, oracle, mode, request):
    """S and V are symmetric in every Eagle run (the library then computes half of W = S V S); an arbitrary caller may
    pass anything, and m^T S V S m must still match calculate_a_and_vara_rcpp.cpp:97-112."""
    Mt8, geno, S, V, ahat = big
    L, n = Mt8.shape
    rng = np.random.default_rng(11)
    S2 = S + 0.05 * rng.standard_normal((n, n)) / np.sqrt(n)
    V2 = V + 0.05 * rng.standard_normal((n, n)) / np.sqrt(n)
    request.addfinalizer(lambda: api.set_scan_mode(1))
    api.set_scan_mode(mode)
    a_ref, v_ref = oracle.scan_from_i8(Mt8[:4096], S2, V2, ahat)
    res = api.calculate_a_and_vara_rcpp(geno["asciifileMt"], NA, S2, V2, 8.0, (L, n), ahat)
    _close(res["a"][:4096], a_ref)
    np.testing.assert_allclose(res["vara"].ravel()[:4096], v_ref, rtol=RTOL_DIGITS)


def test_streamed_paths_match_resident(big, api, oracle, monkeypatch):
    """Files larger than the resident budget are streamed through HBM in marker chunks (the out-of-core case, BASELINE
    config 4); forced here with a 5 MB budget.  Results must equal the resident path bit for bit (same kernels, same
    per-marker order; MM^T is an exact integer sum over windows)."""
    Mt8, geno, S, V, ahat = big
    L, n = Mt8.shape
    rng = np.random.default_rng(2)
    P = rng.standard_normal((n, n)) / n
    y = rng.standard_normal(n)
    sel = np.array([7.0, 5000.0, float(L - 1)])
    api.drop_cache()
    res = {}
    for tag, budget in (("resident", None), ("streamed", "0.005")):
        if budget:
            monkeypatch.setenv("EAGLE_HIP_MAX_RESIDENT_GB", budget)
        api.drop_cache()
        msgs = []
        res[tag] = dict(
            mmt=api.calculateMMt_rcpp(geno["asciifileM"], 8.0, 4, NA, (n, L), quiet=False, message=msgs.append),
            mmt_m=api.calculateMMt_rcpp(geno["asciifileM"], 8.0, 4, sel, (n, L)),
            scan=api.calculate_a_and_vara_rcpp(geno["asciifileMt"], sel, S, V, 8.0, (L, n), ahat, quiet=False, message=msgs.append),
            idx=api.last_scan_argmax(),
            cert=api.last_scan_certificate(),
            ar=api.calculate_reduced_a_rcpp(geno["asciifileMt"], 0.7, P, y, 8.0, (n, L), NA))
        assert any("streamed" in m for m in msgs) == (budget is not None)
        if budget:  # the library's out-of-core books (eagle_last_stream_stats) of the last streamed call: the reduced-a pass
            st = api.last_stream_stats()
            assert st["chunks"] >= 2 and st["file_bytes"] >= L * n // 4 and st["pread_s"] > 0.0
            assert 0.0 < st["load_first_s"] <= st["load_s"] <= st["wall_s"] + 1e-3 and st["kernel_s"] > 0.0
            assert 0.0 <= st["starved_s"] <= st["wall_s"] and 0.0 <= st["load_hidden_frac"] <= 1.0
    monkeypatch.delenv("EAGLE_HIP_MAX_RESIDENT_GB")
    api.drop_cache()
    for k in ("mmt", "mmt_m", "ar"):
        np.testing.assert_array_equal(res["streamed"][k], res["resident"][k])
    np.testing.assert_array_equal(res["streamed"]["scan"]["a"], res["resident"]["scan"]["a"])
    # vara: bit for bit as well -- the blocks of a streamed file are certified against ONE lower bound of the maximum over all of
    # them (their per-marker bounds stay on the device; the candidates' rows are read back from the file), so the markers handed
    # to the fp64 kernel, and with them every returned bit, are those of the resident scan: residency never changes a result
    np.testing.assert_array_equal(res["streamed"]["scan"]["vara"], res["resident"]["scan"]["vara"])
    assert res["streamed"]["cert"] == res["resident"]["cert"]
    assert res["streamed"]["idx"] == res["resident"]["idx"]
    G = Mt8.astype(np.float64)
    np.testing.assert_array_equal(res["streamed"]["mmt"], G.T @ G)


def test_large_n_device_path(api, oracle):
    """n = 12000 (47 column tiles: odd count; more individuals than one LDS image of the genotype pass holds, so it sweeps
    two column chunks), device-resident entry points only, checked against the oracle on a marker sample and through
    exact properties of MM^T."""
    import torch
    from eagleeverything_amd.sharded import DeviceShard
    n, L = 12000, 4096
    sh = DeviceShard(n, L)
    sh.fill_synthetic(seed=99)
    gen = torch.Generator(device=sh.dev)
    gen.manual_seed(5)
    A = torch.randn((n, 32), generator=gen, device=sh.dev, dtype=torch.float64) / 32.0
    S = 0.4 * torch.eye(n, dtype=torch.float64, device=sh.dev) + A @ A.T
    V = 0.5 * torch.eye(n, dtype=torch.float64, device=sh.dev) - 0.02 * (A[:, :4] @ A[:, :4].T)
    ahat = torch.randn(n, generator=gen, device=sh.dev, dtype=torch.float64)
    sh.set_operands(S, V, ahat)
    Mt_s = sh.Mt8[:384, :n].cpu().numpy()
    a_ref, v_ref = oracle.scan_from_i8(Mt_s, S.cpu().numpy(), V.cpu().numpy(), ahat.cpu().numpy())
    for mode in (1, 0):
        sh.mode = mode
        sh.scan()
        torch.cuda.synchronize()
        np.testing.assert_allclose(sh.a[:384].cpu().numpy(), a_ref, rtol=1e-9, atol=1e-12 * np.abs(a_ref).max())
        np.testing.assert_allclose(sh.vara[:384].cpu().numpy(), v_ref, rtol=RTOL_DIGITS if mode else RTOL)
    # MM^T: exact trace (sum of squares) and symmetry at this size
    c32 = sh.mmt_partial()
    MMt, mx = sh.mmt_finish(c32)
    Mt = sh.Mt8[:L, :n]
    diag_ref = (Mt.to(torch.int32) ** 2).sum(dim=0).double()
    assert torch.equal(torch.diagonal(MMt), diag_ref)
    assert torch.equal(MMt, MMt.T)
    cols = torch.tensor([0, 1, 255, 256, 6000, n - 1], device=sh.dev)
    ref_cols = Mt.double().T @ Mt[:, cols].double()
    assert torch.equal(MMt[:, cols], ref_cols)
    assert float(mx) == float(MMt.max())


@pytest.mark.parametrize("n,L", [(1, 1), (2, 7), (15, 33), (17, 300), (255, 64), (256, 257), (257, 1000), (300, 1), (511, 513), (640, 129)])
def test_ragged_small_shapes(n, L, api, oracle, tmp_path):
    """Sizes around every padding / tile boundary (1, 16, 256) through the reference-shaped entry points, all modes:
    MM^T and ReadBlock bit-exact, a / vara / reduced a against the oracle, same selected marker."""
    rng = np.random.default_rng(1000 * n + L)
    maf = rng.uniform(0.0, 0.5, size=L)
    Mt8 = (rng.binomial(2, maf[:, None], size=(L, n)) - 1).astype(np.int8)
    geno = synth.write_geno_pair(str(tmp_path), Mt8)
    A = rng.standard_normal((n, max(1, n // 3))) / 4.0
    S = np.eye(n) + A @ A.T
    V = 0.7 * np.eye(n) - 0.05 * (A[:, :1] @ A[:, :1].T)
    ahat = rng.standard_normal(n)
    P = np.eye(n) * 0.3 + 0.01 * (A @ A.T)
    y = rng.standard_normal((n, 1))
    assert np.array_equal(api.calculateMMt_rcpp(geno["asciifileM"], 8.0, 2, NA, (n, L)),
                          oracle.calculateMMt_rcpp(geno["asciifileM"], 8.0, 2, NA, (n, L)))
    np.testing.assert_array_equal(api.ReadBlock(geno["asciifileMt"], 0, n, L), Mt8.astype(np.float64))
    ref = oracle.calculate_a_and_vara_rcpp(geno["asciifileMt"], NA, S, V, 8.0, (L, n), ahat)
    tsq_ref, idx_ref, _ = oracle.tsq_argmax(ref["a"], ref["vara"])
    vs = np.abs(ref["vara"]).max()
    picked = {}
    try:
        for mode in (1, 0):
            api.set_scan_mode(mode)
            res = api.calculate_a_and_vara_rcpp(geno["asciifileMt"], NA, S, V, 8.0, (L, n), ahat)
            _close(res["a"], ref["a"])
            rt = RTOL_DIGITS if mode == 1 else RTOL
            # monomorphic markers have vara = c^2 1'W1 in both; everything is compared with the absolute floor of the budget
            np.testing.assert_allclose(res["vara"].ravel(), ref["vara"].ravel(), rtol=rt, atol=rt * 1e-3 * vs)
            picked[mode] = api.last_scan_argmax()[0]
            # R takes the arg-max itself on the returned arrays (find_qtl.R:71-83): same marker
            assert picked[mode] == _r_which_max(res["a"], res["vara"])
    finally:
        api.set_scan_mode(1)
    # the digit-slice scan is certified: it selects the marker the fp64 scan selects, by construction
    assert picked[1] == picked[0]
    if not _selection_rests_on_noise(ref, idx_ref):
        assert picked[1] == idx_ref
    ar = api.calculate_reduced_a_rcpp(geno["asciifileMt"], 0.8, P, y, 8.0, (n, L), NA)
    ar_ref = oracle.calculate_reduced_a_rcpp(geno["asciifileMt"], 0.8, P, y, 8.0, (n, L), NA)
    _close(ar, ar_ref)
    api.drop_cache()


def test_fuzz_random_shapes(api, oracle, tmp_path):
    """Random (n, L) below (900, 3000), random allele frequencies incl. monomorphic markers, either coding of the common
    allele, and every fourth case a W that annihilates constants: MM^T bit-exact, a to 1e-9, vara to the digit budget."""
    rng0 = np.random.default_rng(2024)
    for it in range(16):
        n, L = int(rng0.integers(1, 900)), int(rng0.integers(1, 3000))
        rng = np.random.default_rng(it)
        maf = rng.uniform(0.0, 0.5, size=L)
        Mt8 = (rng.binomial(2, maf[:, None], size=(L, n)) - 1).astype(np.int8)
        if it % 3 == 0:
            Mt8 = -Mt8
        d = tmp_path / ("c%d" % it)
        d.mkdir()
        geno = synth.write_geno_pair(str(d), Mt8)
        A = rng.standard_normal((n, max(1, n // 3))) / 4.0
        S = np.eye(n) + A @ A.T
        V = 0.7 * np.eye(n) - 0.05 * (A[:, :1] @ A[:, :1].T)
        if it % 4 == 1:
            P1 = np.eye(n) - np.ones((n, n)) / n
            V, S = P1 @ V @ P1, 0.8 * np.eye(n)
        ahat = rng.standard_normal(n)
        assert np.array_equal(api.calculateMMt_rcpp(geno["asciifileM"], 8.0, 2, NA, (n, L)),
                              oracle.calculateMMt_rcpp(geno["asciifileM"], 8.0, 2, NA, (n, L))), (n, L)
        ref = oracle.calculate_a_and_vara_rcpp(geno["asciifileMt"], NA, S, V, 8.0, (L, n), ahat)
        res = api.calculate_a_and_vara_rcpp(geno["asciifileMt"], NA, S, V, 8.0, (L, n), ahat)
        _close(res["a"], ref["a"])
        vs = np.abs(ref["vara"]).max()
        np.testing.assert_allclose(res["vara"], ref["vara"], rtol=RTOL_DIGITS, atol=RTOL_DIGITS * 1e-4 * vs, err_msg=str((n, L)))
        api.drop_cache()


def test_vara_rare_variants_and_monomorphic_markers(api, oracle):
    """Markers that are almost constant over the individuals, against a W that annihilates constants (a model with an
    intercept): their vara is orders of magnitude below the diagonal term of the raw g-1 coding.  The digit-slice kernel
    works on the re-centred genotypes, so the relative budget (1e-7) holds for them too, with the automatic digit count."""
    import torch
    from eagleeverything_amd.sharded import DeviceShard
    n, L = 700, 4096
    rng = np.random.default_rng(17)
    maf = np.concatenate([np.zeros(64), rng.uniform(0.002, 0.02, 2048), rng.uniform(0.05, 0.5, L - 64 - 2048)])
    g = rng.binomial(2, maf[:, None], size=(L, n)).astype(np.int8)
    flip = rng.random(L) < 0.5           # half of the files code the common allele as 2
    g[flip] = 2 - g[flip]
    Mt8 = (g - 1).astype(np.int8)
    sh = DeviceShard(n, L)
    sh.Mt8.zero_()
    sh.Mt8[:L, :n] = torch.from_numpy(Mt8).to(sh.dev)
    A = rng.standard_normal((n, 40)) / 8.0
    V0 = np.eye(n) + A @ A.T
    P1 = np.eye(n) - np.ones((n, n)) / n
    V = P1 @ V0 @ P1                     # V 1 = 0
    S = 0.7 * np.eye(n)
    ahat = rng.standard_normal(n)
    sh.set_operands(S, V, ahat)
    a_ref, v_ref = oracle.scan_from_i8(Mt8, S, V, ahat)
    sh.mode = 1
    sh.scan()
    torch.cuda.synchronize()
    used, bound, _ = sh.vara_i8_info()
    vara = sh.vara[:L].cpu().numpy()
    cs = sh.cshift[:L].cpu().numpy()
    # every marker is centred on a homozygote (the majority genotype, or the commoner homozygote when the heterozygote is the
    # majority), so that the stored rows have no negative entries; only an all-heterozygote marker would keep c = 0
    assert set(np.unique(cs)) == {-1, 1}
    typical = np.median(v_ref[maf > 0.05])
    mono = (Mt8 == Mt8[:, :1]).all(axis=1)   # includes rare markers that drew no minor allele
    assert mono.sum() >= 64
    assert np.all(np.abs(vara[mono]) <= 1e-10 * typical) and np.all(np.abs(v_ref[mono]) <= 1e-10 * typical)
    rare = (maf > 0) & (maf <= 0.02) & ~mono
    assert v_ref[rare].max() < 0.2 * typical                       # far below the common markers
    np.testing.assert_allclose(vara[~mono], v_ref[~mono], rtol=RTOL_DIGITS)
    np.testing.assert_allclose(sh.a[:L].cpu().numpy(), a_ref, rtol=1e-9, atol=1e-12 * np.abs(a_ref).max())
    # the same kernel on the raw coding (no re-centring) stays inside the absolute bound, but loses the rare markers' relative accuracy
    sh.vara_prepare()                    # zeroes the integer accumulators again
    sh._check(sh.L.eagle_dev_vara_i8_mfma(sh.ctx, sh.Mt8.data_ptr(), sh.Lp, sh.np_, sh.np_, sh.nslices, sh._ws().data_ptr(),
                                          sh.vara.data_ptr(), None, sh._stream()))
    torch.cuda.synchronize()
    raw = sh.vara[:L].cpu().numpy()
    assert np.abs(raw - v_ref).max() <= bound
    assert np.abs(raw - v_ref)[rare].max() >= np.abs(vara - v_ref)[rare].max()


def test_certified_argmax_planted_near_tie_and_cancelling_form(api, oracle, tmp_path):
    """VERDICT r1 item 1.  Two markers whose tsq differ by ~1e-10 relative -- far inside the digit kernel's error bound -- must
    be told apart exactly as the fp64 scan tells them apart, and a marker whose quadratic form cancels against its diagonal
    term (vara = 1e-6 of the diagonal term) must come back with its fp64 value: both are re-evaluated by the certification
    step of the digit-slice scan (eagle_dev_scan_certify), through the reference-shaped entry point."""
    import ctypes as C
    from eagleeverything_amd import _lib
    n, L = 600, 3000
    rng = np.random.default_rng(99)
    maf = rng.uniform(0.1, 0.5, size=L)
    Mt8 = (rng.binomial(2, maf[:, None], size=(L, n)) - 1).astype(np.int8)
    E = rng.standard_normal((n, n)) * 1e-3
    E = 0.5 * (E + E.T)
    d = rng.uniform(0.8, 1.2, size=n)
    j0 = n - 1
    E[j0, :] = 0.0; E[:, j0] = 0.0
    d[j0] = 3e-8        # individual j0 moves vara by ~1e-10 relative (vara ~ 0.81 * 300)
    V = np.diag(d) + E
    S = 0.9 * np.eye(n)
    ahat = rng.standard_normal(n)
    ahat[j0] = 1e-14    # ... and a not at all
    # scenario 1: plant the pair at the top.  A = the strongest marker with genotype 0 at j0, B = A with +1 there:
    # vara_B = vara_A + W[j0][j0] > vara_A, so tsq_B = tsq_A (1 - ~1e-10); B sits at the LOWER index.
    Mt8[:, j0] = 0
    a0, v0 = oracle.scan_from_i8(Mt8, S, V, ahat)
    t = int(np.argmax(a0 ** 2 / v0))
    iB, iA = 17, 2500
    Mt8[iA] = Mt8[t]
    Mt8[iB] = Mt8[t]
    Mt8[iB, j0] = 1
    Mt8[t] = 0          # the original becomes a dead marker (a = vara = 0: tsq NaN, skipped)
    d1 = tmp_path / "s1"; d1.mkdir()
    geno = synth.write_geno_pair(str(d1), Mt8)
    ref = oracle.calculate_a_and_vara_rcpp(geno["asciifileMt"], NA, S, V, 8.0, (L, n), ahat)
    tsq_ref, idx_ref, mx_ref = oracle.tsq_argmax(ref["a"], ref["vara"])
    gap = (tsq_ref[iA] - tsq_ref[iB]) / tsq_ref[iA]
    assert idx_ref == iA + 1 and 1e-11 < gap < 1e-9, (idx_ref, gap)
    out = {}
    try:
        for mode in (1, 0):
            api.set_scan_mode(mode)
            res = api.calculate_a_and_vara_rcpp(geno["asciifileMt"], NA, S, V, 8.0, (L, n), ahat)
            out[mode] = res
            assert api.last_scan_argmax()[0] == iA + 1 == _r_which_max(res["a"], res["vara"]), mode
    finally:
        api.set_scan_mode(1)
    # the candidates carry the fp64 kernel's values bit for bit
    assert out[1]["vara"][iA, 0] == out[0]["vara"][iA, 0] and out[1]["vara"][iB, 0] == out[0]["vara"][iB, 0]
    np.testing.assert_allclose(out[1]["vara"].ravel(), ref["vara"].ravel(), rtol=RTOL_DIGITS, atol=1e-12)
    nre, nfl, fell = C.c_long(), C.c_long(), C.c_int()
    ctx = api.context()
    api.set_scan_mode(1)
    api.calculate_a_and_vara_rcpp(geno["asciifileMt"], NA, S, V, 8.0, (L, n), ahat)
    assert _lib.load().eagle_last_scan_certificate(ctx, C.byref(nre), C.byref(nfl), C.byref(fell)) == 0
    assert 2 <= nre.value <= 64 and fell.value == 0, (nre.value, nfl.value)
    api.drop_cache()

    # scenario 2: W = (1 + eps) I - s s^T / k, marker C = s: vara_C = eps k against a diagonal term of (1 + eps) k
    k, eps, iC = 200, 1e-6, 1234
    s_vec = np.zeros(n); s_vec[rng.choice(n, k, replace=False)] = 1.0
    V2 = (1.0 + eps) * np.eye(n) - np.outer(s_vec, s_vec) / k
    S2 = np.eye(n)
    Mt8b = (rng.binomial(2, maf[:, None], size=(L, n)) - 1).astype(np.int8)
    Mt8b[iC] = s_vec.astype(np.int8)
    d2 = tmp_path / "s2"; d2.mkdir()
    geno2 = synth.write_geno_pair(str(d2), Mt8b)
    ref2 = oracle.calculate_a_and_vara_rcpp(geno2["asciifileMt"], NA, S2, V2, 8.0, (L, n), ahat)
    assert abs(ref2["vara"][iC, 0] / (eps * k) - 1.0) < 1e-6
    res2 = {}
    try:
        for mode in (1, 0):
            api.set_scan_mode(mode)
            res2[mode] = api.calculate_a_and_vara_rcpp(geno2["asciifileMt"], NA, S2, V2, 8.0, (L, n), ahat)
            if mode == 1:
                assert _lib.load().eagle_last_scan_certificate(ctx, C.byref(nre), C.byref(nfl), C.byref(fell)) == 0
                assert nfl.value >= 1 and fell.value == 0
    finally:
        api.set_scan_mode(1)
    assert res2[1]["vara"][iC, 0] == res2[0]["vara"][iC, 0]                     # flagged -> the fp64 kernel's value
    np.testing.assert_allclose(res2[1]["vara"][iC, 0], ref2["vara"][iC, 0], rtol=1e-7)
    np.testing.assert_allclose(res2[1]["vara"].ravel(), ref2["vara"].ravel(), rtol=RTOL_DIGITS)
    assert api.last_scan_argmax()[0] == oracle.tsq_argmax(ref2["a"], ref2["vara"])[1]
    api.drop_cache()


def test_certified_candidates_are_bitwise_fp64_device_api(api, oracle):
    """Device-resident form (what bench.py and the sharded driver call): after certify() the re-evaluated markers hold
    bit for bit what eagle_dev_vara_f64 computes for them, the rest the digit-slice values, and the arg-max of the
    certified arrays is the arg-max of the fp64 scan.  Also the overflow fallback: with one digit slice forced nearly
    every marker is flagged, more than the 2048 the candidate path takes, and the whole block is redone in fp64."""
    import torch
    from eagleeverything_amd.sharded import DeviceShard
    n, L = 700, 9000
    rng = np.random.default_rng(5)
    Mt8 = synth.genotypes_marker_major(n, L, seed=11)
    Mt8[4000] = Mt8[100]                      # an exact duplicate pair: identical tsq, the first index wins in both modes
    sh = DeviceShard(n, L)
    sh.Mt8.zero_()
    sh.Mt8[:L, :n] = torch.from_numpy(Mt8).to(sh.dev)
    A = rng.standard_normal((n, 50)) / 8.0
    S = np.eye(n) + A @ A.T
    V = 0.6 * np.eye(n) - 0.02 * (A[:, :4] @ A[:, :4].T)
    ahat = rng.standard_normal(n)
    sh.set_operands(S, V, ahat)
    sh.mode = 0
    sh.scan()
    torch.cuda.synchronize()
    v64 = sh.vara[:L].clone()
    best64 = sh.best()
    sh.mode = 1
    sh.certified = False
    sh.scan()
    torch.cuda.synchronize()
    vdig = sh.vara[:L].clone()
    sh.certified = True
    sh.scan()
    torch.cuda.synchronize()
    vcert = sh.vara[:L].clone()
    info = sh.certificate()
    assert info["overflow"] == 0 and 1 <= info["reevaluated"] <= 64
    idx = np.frombuffer(sh.cert_ws[256:256 + 8 * info["reevaluated"]].cpu().numpy().tobytes(), dtype=np.int64)
    mask = torch.zeros(L, dtype=torch.bool, device=sh.dev)
    mask[torch.from_numpy(idx.copy()).to(sh.dev)] = True
    assert torch.equal(vcert[mask], v64[mask])            # bitwise: the fp64 kernel's values
    assert torch.equal(vcert[~mask], vdig[~mask])         # untouched digit-slice values
    assert sh.best()[:2] == best64[:2]
    # exact tie of the duplicate pair is kept (same bits in either coding), the first index is reported
    assert vcert[100].item() == vcert[4000].item() and sh.a[100].item() == sh.a[4000].item()
    # overflow fallback
    sh.nslices = 1
    sh.ws = None
    sh.scan()
    torch.cuda.synchronize()
    info = sh.certificate()
    assert info["overflow"] == 1
    assert torch.equal(sh.vara[:L], v64)
    assert sh.best()[:2] == best64[:2]
    sh.nslices = 0


def test_overflow_fallback_through_the_reference_shaped_call_resident_streamed_and_sharded(api, tmp_path, monkeypatch):
    """More candidates than the re-evaluation buffer holds (one digit slice forced: nearly every marker's bound exceeds 1e-7 of
    its vara) must end in the fp64 scan's values for the whole file -- in the one-block flow (device-side gate), in the flow of
    a streamed file (per-block bounds, ONE selection after the last block, the file streamed a second time through the fp64
    kernel) and with the markers on two sub-contexts of the card.  Expected: bit for bit the arrays of scan mode 0."""
    n, L = 600, 9000
    rng = np.random.default_rng(17)
    Mt8 = synth.genotypes_marker_major(n, L, seed=23)
    geno = synth.write_geno_pair(str(tmp_path), Mt8)
    A = rng.standard_normal((n, 40)) / 8.0
    S = np.eye(n) + A @ A.T
    V = 0.6 * np.eye(n) - 0.02 * (A[:, :4] @ A[:, :4].T)
    ahat = rng.standard_normal(n)
    call = lambda dev: api.calculate_a_and_vara_rcpp(geno["asciifileMt"], NA, S, V, 8.0, (L, n), ahat, device=dev)
    api.set_scan_mode(0)
    ref = call(0)
    best_ref = api.last_scan_argmax()
    api.set_scan_mode(1)
    try:
        api.set_scan_slices(1)
        res = call(0)
        assert api.last_scan_certificate()[2] is True                       # fell back
        np.testing.assert_array_equal(res["vara"], ref["vara"])
        np.testing.assert_array_equal(res["a"], ref["a"])
        assert api.last_scan_argmax()[:2] == best_ref[:2]
        monkeypatch.setenv("EAGLE_HIP_MAX_RESIDENT_GB", "0.0012")           # 768-byte rows: chunks of 768 markers, 12 blocks
        api.drop_cache()
        res = call(0)
        assert api.last_stream_stats()["chunks"] >= 4 and api.last_scan_certificate()[2] is True
        np.testing.assert_array_equal(res["vara"], ref["vara"])
        np.testing.assert_array_equal(res["a"], ref["a"])
        assert api.last_scan_argmax()[:2] == best_ref[:2]
        monkeypatch.delenv("EAGLE_HIP_MAX_RESIDENT_GB")
        dev = (0, 0)
        api.set_scan_slices(1, device=dev)
        res = call(dev)
        np.testing.assert_array_equal(res["vara"], ref["vara"])
        assert api.last_scan_argmax(device=dev)[:2] == best_ref[:2]
        api.set_scan_slices(0, device=dev)
        api.drop_cache(device=dev)
    finally:
        api.set_scan_slices(0)
        api.drop_cache()


def test_stochastic_rounding_option(api, oracle):
    """eagle_set_scan_rounding(1) / EAGLE_SLICES_STOCHASTIC (opt-in): the digits of W rounded at random (unbiased, keyed by
    position).  One digit fewer than round-to-nearest at n = 5000; every error inside the Hoeffding radius
    8.355 q2 2^(e+1-8S) that the certification uses and inside the digit budget; reproducible; same selected marker."""
    import torch
    from eagleeverything_amd.sharded import DeviceShard
    n, L = 5000, 32768
    sh = DeviceShard(n, L)
    sh.fill_synthetic(seed=77)
    gen = torch.Generator(device=sh.dev)
    gen.manual_seed(3)
    A = torch.randn((n, 48), generator=gen, device=sh.dev, dtype=torch.float64) / 40.0
    S = 0.5 * torch.eye(n, dtype=torch.float64, device=sh.dev) + A @ A.T
    V = 0.6 * torch.eye(n, dtype=torch.float64, device=sh.dev) - 0.03 * (A[:, :6] @ A[:, :6].T)
    ahat = torch.randn(n, generator=gen, device=sh.dev, dtype=torch.float64)
    sh.set_operands(S, V, ahat)
    rows = torch.arange(0, 2048, device=sh.dev)
    Mt_s = sh.Mt8[rows][:, :n].cpu().numpy()
    a_ref, v_ref = oracle.scan_from_i8(Mt_s, S.cpu().numpy(), V.cpu().numpy(), ahat.cpu().numpy())
    sh.mode = 1
    sh.L.eagle_dev_set_tune(sh.ctx, 29)        # round to nearest WITHOUT the spectral digit saving: the worst-case digit count
    sh.scan()
    torch.cuda.synchronize()
    sh.L.eagle_dev_set_tune(sh.ctx, 0)
    S_near, _, maxoff = sh.vara_i8_info()
    best_near = sh.best()[:2]
    sh.stochastic = True
    sh.certified = False                       # raw digit values first: the bound must hold for them
    sh.scan()
    torch.cuda.synchronize()
    S_rand = sh.vara_i8_info()[0]
    assert S_rand == S_near - 1, (S_near, S_rand)
    raw = sh.vara[rows].cpu().numpy()
    e = sh.last_e                              # the digits' scale exponent (max |off-diagonal| < 2^e, or < 1.96 * 2^e)
    assert maxoff < 1.96 * 2.0 ** e and maxoff >= 0.49 * 2.0 ** e
    cs = sh.cshift[rows].cpu().numpy().astype(np.int64)
    q2 = ((Mt_s.astype(np.int64) - cs[:, None]) ** 2).sum(axis=1)
    assert np.array_equal(sh.l1[rows, 1].cpu().numpy(), q2)
    radius = 8.355 * q2 * 2.0 ** (e + 1 - 8 * S_rand)
    err = np.abs(raw - v_ref)
    assert np.all(err <= radius + 1e-12 * np.abs(v_ref))
    assert np.max(radius / np.abs(v_ref)) < 9e-7          # what the certification checks per marker (1.8 x the budget of 5e-7)
    assert np.max(err / np.abs(v_ref)) < 1e-8             # the Hoeffding radius is generous: typical errors are ~ a tenth of it
    sh.certified = True
    sh.scan()
    torch.cuda.synchronize()
    v1 = sh.vara[:L].clone()
    assert sh.best()[:2] == best_near                      # same marker, and its fp64 value, whichever rounding
    assert sh.certificate()["overflow"] == 0
    sh.scan()
    torch.cuda.synchronize()
    assert torch.equal(sh.vara[:L], v1)                    # position-keyed random bits: reproducible
    np.testing.assert_allclose(v1[rows].cpu().numpy(), v_ref, rtol=RTOL_DIGITS)
    # the reference-shaped switch
    api.set_scan_rounding(1)
    api.set_scan_rounding(0)


def test_vara_fp4_fp6_engine(api, oracle):
    """The block-scaled form of the vara kernel (genotypes fp4, base-33 digits of W as fp6, exact fp32 sums): against the
    oracle with the automatic digit count, and inside its documented bound n_pad^2 * 2^(e-5S) with fewer digits."""
    import torch
    from eagleeverything_amd.sharded import DeviceShard
    n, L = 700, 3000
    sh = DeviceShard(n, L)
    sh.fill_synthetic(seed=3)
    gen = torch.Generator(device=sh.dev)
    gen.manual_seed(5)
    A = torch.randn((n, 48), generator=gen, device=sh.dev, dtype=torch.float64) / 32.0
    S = 0.5 * torch.eye(n, dtype=torch.float64, device=sh.dev) + A @ A.T
    V = 0.6 * torch.eye(n, dtype=torch.float64, device=sh.dev) - 0.03 * (A[:, :6] @ A[:, :6].T)
    ahat = torch.randn(n, generator=gen, device=sh.dev, dtype=torch.float64)
    sh.set_operands(S, V, ahat)
    a_ref, v_ref = oracle.scan_from_i8(sh.Mt8[:L, :n].cpu().numpy(), S.cpu().numpy(), V.cpu().numpy(), ahat.cpu().numpy())
    sh.mode = 2
    sh.scan()
    torch.cuda.synchronize()
    first = sh.vara[:L].cpu().numpy()
    np.testing.assert_allclose(first, v_ref, rtol=RTOL_DIGITS)
    np.testing.assert_allclose(sh.a[:L].cpu().numpy(), a_ref, rtol=1e-9, atol=1e-12 * np.abs(a_ref).max())
    sh.scan()
    torch.cuda.synchronize()
    np.testing.assert_array_equal(sh.vara[:L].cpu().numpy(), first)  # integer atomics: bitwise reproducible
    for S_ in (4, 6, 8, 10):
        sh.nslices, sh.ws = S_, None
        sh.scan()
        torch.cuda.synchronize()
        used, bound, _ = sh.vara_i8_info()
        assert used == S_ and np.abs(sh.vara[:L].cpu().numpy() - v_ref).max() <= bound + 1e-12 * np.abs(v_ref).max()


def test_config_C3_shape_single_gpu(api, oracle):
    """BASELINE.json configs[2] at its full shape on ONE card (10,000 individuals x 1,000,000 markers, resident): the scan
    against the oracle on a marker sample, MM^T through exact size-independent properties (diagonal = sums of squares,
    symmetry, selected columns against an int-exact torch product), arg-max against the full tsq vector."""
    import torch
    from eagleeverything_amd.sharded import DeviceShard
    n, L = 10000, 1000000
    sh = DeviceShard(n, L)
    sh.fill_synthetic(seed=20240601)
    gen = torch.Generator(device=sh.dev)
    gen.manual_seed(11)
    A = torch.randn((n, 48), generator=gen, device=sh.dev, dtype=torch.float64) / 40.0
    S = 0.5 * torch.eye(n, dtype=torch.float64, device=sh.dev) + A @ A.T
    V = 0.6 * torch.eye(n, dtype=torch.float64, device=sh.dev) - 0.03 * (A[:, :6] @ A[:, :6].T)
    ahat = torch.randn(n, generator=gen, device=sh.dev, dtype=torch.float64)
    sh.set_operands(S, V, ahat)
    del A
    sh.mode = 1
    sh.scan()
    torch.cuda.synchronize()
    rows = torch.cat([torch.arange(0, 96), torch.arange(499968, 500064), torch.arange(L - 96, L)]).to(sh.dev)
    Mt_s = sh.Mt8[rows][:, :n].cpu().numpy()
    a_ref, v_ref = oracle.scan_from_i8(Mt_s, S.cpu().numpy(), V.cpu().numpy(), ahat.cpu().numpy())
    np.testing.assert_allclose(sh.a[rows].cpu().numpy(), a_ref, rtol=1e-9, atol=1e-12 * np.abs(a_ref).max())
    np.testing.assert_allclose(sh.vara[rows].cpu().numpy(), v_ref, rtol=RTOL_DIGITS)
    tsq = sh.a[:L] ** 2 / sh.vara[:L]
    tsqmax, gidx, near = sh.best()
    assert gidx == int(torch.argmax(tsq)) and tsqmax == float(tsq.max())
    del S, V, tsq
    sh.release_operands()
    # MM^T at 2 x 10^14 multiply-adds: exact integer properties
    sh.individual_major()
    c32 = sh.mmt_partial()
    MMt, mx = sh.mmt_finish(c32)
    del c32
    cols = torch.tensor([0, 1, 255, 256, 4999, 5120, n - 1], device=sh.dev)
    diag_ref = torch.zeros(n, dtype=torch.float64, device=sh.dev)
    cols_ref = torch.zeros((n, cols.numel()), dtype=torch.float64, device=sh.dev)
    for r0 in range(0, L, 65536):
        blk = sh.Mt8[r0:min(L, r0 + 65536), :n]
        diag_ref += (blk.to(torch.int16) ** 2).sum(dim=0, dtype=torch.int64).double()
        cols_ref += blk.double().T @ blk[:, cols].double()
    assert torch.equal(torch.diagonal(MMt), diag_ref)
    assert torch.equal(MMt, MMt.T)
    assert torch.equal(MMt[:, cols], cols_ref)
    assert float(mx) == float(MMt.max())


@pytest.mark.parametrize("n", [256, 384, 768, 1280, 2816])
def test_dev_gemm_f64_layout(api, n):
    """The fp64 MFMA fragment maps, the LDS-DMA swizzles and the split-K tail of the GEMM kernels, checked with asymmetric
    integer-valued data (exact in fp64, so any summation order must give the same bits): k_gemm_f64_dma (256 x 128 tiles) for
    multiples of 256 -- 2816 is 242 tiles, all of them in the split-K tail; 1280 has whole tiles only -- and the 128 x 128 form
    for 384."""
    import torch
    from eagleeverything_amd import _lib
    L = _lib.load()
    ctx = api.context(0)
    rng = np.random.default_rng(n)
    A = rng.integers(-8, 9, size=(n, n)).astype(np.float64)
    B = rng.integers(-8, 9, size=(n, n)).astype(np.float64)
    dA = torch.from_numpy(A).cuda(); dB = torch.from_numpy(B).cuda(); dC = torch.zeros((n, n), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    rc = L.eagle_dev_gemm_f64(ctx, dA.data_ptr(), dB.data_ptr(), dC.data_ptr(), n, None)
    assert rc == 0
    torch.cuda.synchronize()
    np.testing.assert_array_equal(dC.cpu().numpy(), A @ B)


@pytest.mark.parametrize("n", [500, 1400])
def test_scan_operands_products_exact_on_integer_operands(api, n):
    """W = S (V S) and v = S a_hat through eagle_dev_scan_operands (symmetric operands: upper tiles + fold), through the general
    path (non-symmetric V: both triangles from memory) and through eagle_dev_scan_operands_rows (row blocks whose length is an odd
    multiple of 128: the half tile at the end of a range) on small-integer matrices, where every product and sum is exact in fp64:
    the folded image must equal numpy's bit for bit whatever the tile shape or summation order."""
    import torch
    from eagleeverything_amd.sharded import DeviceShard
    sh = DeviceShard(n, 256)
    np_ = sh.np_
    rng = np.random.default_rng(n)
    S = rng.integers(-3, 4, size=(n, n)).astype(np.float64); S = S + S.T
    V = rng.integers(-3, 4, size=(n, n)).astype(np.float64); V = V + V.T
    ahat = rng.integers(-4, 5, size=n).astype(np.float64)
    for sym in (True, False):
        Vx = V.copy()
        if not sym:
            Vx[1, 0] += 5.0
        W = S @ Vx @ S
        fold = np.triu(W, 1) + np.tril(W, -1).T + np.diag(np.diag(W))
        sh.set_operands(S, Vx, ahat)
        sh.scan_operands(None)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(sh.Wu[:n, :n].cpu().numpy(), fold)
        assert float(sh.Wu[n:, :].abs().max()) == 0.0 and float(sh.Wu[:, n:].abs().max()) == 0.0
        np.testing.assert_array_equal(sh.v[:n].cpu().numpy(), S @ ahat)
        # rows of the W^T image in three ranges with odd 128-row counts
        Wt = torch.zeros((np_, np_), dtype=torch.float64, device=sh.dev)
        cuts = [0, 128, np_ - 384 if np_ >= 768 else 256, np_]
        for r0, r1 in zip(cuts[:-1], cuts[1:]):
            if r0 >= r1:
                continue
            sh._check(sh.L.eagle_dev_scan_operands_rows(sh.ctx, sh.Sa.data_ptr(), sh.Va.data_ptr(), sh.ahat.data_ptr(), n, np_, r0, r1,
                                                        sh.v.data_ptr(), Wt.data_ptr(), sh.tmp.data_ptr(), sh._stream()))
        torch.cuda.synchronize()
        np.testing.assert_array_equal(Wt[:n, :n].cpu().numpy(), W.T)


@pytest.mark.parametrize("n", [3001, 3100, 3500])
def test_mmt_on_384_row_tiles_every_last_tile_shape(n):
    """k_syrk_f4w (384 x 256 tiles on the symmetric output; from 3,072 padded individuals): padded sizes 3072 = 8 x 384, 3328
    (last row tile 256 rows) and 3584 (128 rows), a marker count that splits K unevenly: the live 256-tiles of the exact int32
    accumulator against an fp64 product of the same integers (exact: far below 2^53), the dead tiles untouched."""
    import torch
    from eagleeverything_amd.sharded import DeviceShard
    L = 9000
    sh = DeviceShard(n, L)
    sh.fill_synthetic(seed=n)
    c32 = torch.full((sh.np_, sh.np_), -7, dtype=torch.int32, device=sh.dev)
    upper = torch.triu(torch.ones((sh.np_ // 256, sh.np_ // 256), dtype=torch.bool, device=sh.dev)).repeat_interleave(256, 0).repeat_interleave(256, 1)
    c32[upper] = 0
    M4 = sh.individual_major_fp4()
    assert sh.L.eagle_dev_mmt_accumulate_f4(sh.ctx, M4.data_ptr(), sh.np_, sh.Lp, sh.Lp // 2, c32.data_ptr(), sh._stream()) == 0
    G = sh.Mt8.to(torch.float64)                               # padded rows / columns are zero
    ref = (G.T @ G).to(torch.int32)
    assert torch.equal(c32[upper], ref[upper])
    assert bool((c32[~upper] == -7).all())


@pytest.mark.parametrize("n,L", [(150, 1000), (1000, 5003), (3100, 9000)])
def test_fp4_operand_image_in_one_pass_equals_transpose_then_pack(n, L):
    """k_transpose_pack_fp4 (marker-major int8 -> individual-major fp4 in one pass, the MM^T operand image bench.py times inside
    mmt_build_s) writes byte for byte what eagle_dev_transpose_i8 + eagle_dev_pack_fp4 write, and decodes to the genotypes."""
    import torch
    from eagleeverything_amd.sharded import DeviceShard
    sh = DeviceShard(n, L)
    sh.fill_synthetic(seed=n + L)
    one = sh.individual_major_fp4().clone()        # fused (sh.M8 is None)
    sh.M4 = None
    sh.individual_major()                          # int8 individual-major image ...
    two = sh.individual_major_fp4()                # ... packed
    assert torch.equal(one, two)
    lo, hi = (one & 0xF).to(torch.int16), (one >> 4).to(torch.int16)
    dec = lambda c: torch.where(c == 0x2, 1, torch.where(c == 0xA, -1, 0)).to(torch.int8)
    assert bool(((lo == 0) | (lo == 0x2) | (lo == 0xA)).all()) and bool(((hi == 0) | (hi == 0x2) | (hi == 0xA)).all())
    M = torch.stack([dec(lo), dec(hi)], dim=2).reshape(sh.np_, sh.Lp)
    assert torch.equal(M, sh.Mt8.T)


def test_cached_S_is_verified_and_never_changes_a_result(api, tmp_path, monkeypatch):
    """calculate_a_and_vara keeps the last call's S on the device and starts the next call's products on it while the caller's S is
    uploaded and compared under them (AM(): MMt^-1/2 is the same matrix in every find_qtl call).  Same S -> a hit; one entry of S
    changed -> the products start over; either way the arrays are bit for bit those of a call with the mechanism switched off."""
    n, L = 900, 5000
    rng = np.random.default_rng(8)
    Mt8 = synth.genotypes_marker_major(n, L, seed=31)
    geno = synth.write_geno_pair(str(tmp_path), Mt8)
    A = rng.standard_normal((n, 30)) / 6.0
    S = np.asfortranarray(np.eye(n) + A @ A.T)
    V1 = np.asfortranarray(0.6 * np.eye(n) - 0.02 * (A[:, :5] @ A[:, :5].T))
    V2 = np.asfortranarray(0.9 * np.eye(n) - 0.01 * (A[:, 5:9] @ A[:, 5:9].T))
    ahat = rng.standard_normal(n)
    S2 = S.copy(order="F")
    S2[n // 2, n // 3] += 1e-3                                       # one entry, far from the corner
    call = lambda Sx, Vx: api.calculate_a_and_vara_rcpp(geno["asciifileMt"], NA, Sx, Vx, 8.0, (L, n), ahat)
    monkeypatch.setenv("EAGLE_HIP_NO_SCACHE", "1")
    ref = {k: call(*k_args) for k, k_args in (("S,V1", (S, V1)), ("S,V2", (S, V2)), ("S2,V2", (S2, V2)))}
    monkeypatch.delenv("EAGLE_HIP_NO_SCACHE")
    h0, m0 = api.scan_operand_cache_stats()
    first = call(S, V1)                       # whatever the cache held before: filled or started over
    h1, m1 = api.scan_operand_cache_stats()
    again = call(S, V2)                       # the same S: computed on the device copy, verified under the product
    h2, m2 = api.scan_operand_cache_stats()
    other = call(S2, V2)                      # another S: detected, started over
    h3, m3 = api.scan_operand_cache_stats()
    back = call(S2, V2)                       # and now that one is the cached S
    h4, m4 = api.scan_operand_cache_stats()
    assert (h2 - h1, m2 - m1) == (1, 0) and (h3 - h2, m3 - m2) == (0, 1) and (h4 - h3, m4 - m3) == (1, 0)
    for got, key in ((first, "S,V1"), (again, "S,V2"), (other, "S2,V2"), (back, "S2,V2")):
        np.testing.assert_array_equal(got["a"], ref[key]["a"])
        np.testing.assert_array_equal(got["vara"], ref[key]["vara"])
    assert not np.array_equal(again["a"], other["a"])     # the changed entry does reach the result
    api.drop_cache()


@pytest.mark.parametrize("streamed", [False, True])
def test_S_kept_in_its_arena_slot_above_the_cache_limit(api, tmp_path, monkeypatch, streamed):
    """Above 16,384 padded individuals no second device copy of S is kept (20 GB at n = 50,000): the last scan's S is still in its arena
    slot when nothing has re-laid the arena since, the next scan computes on it, and the caller's matrix is compared with a host copy
    while the card works (a streamed file too).  EAGLE_HIP_SCACHE_MAX_NP=0 forces that form here.  Same S -> a hit and no upload;
    a changed S, an MM^T call in between (its background allocation may replace the arena) or a dropped cache -> an upload; always the
    bits of a call with the mechanism switched off."""
    n, L = 900, 5000
    rng = np.random.default_rng(8)
    Mt8 = synth.genotypes_marker_major(n, L, seed=31)
    geno = synth.write_geno_pair(str(tmp_path), Mt8)
    A = rng.standard_normal((n, 30)) / 6.0
    S = np.asfortranarray(np.eye(n) + A @ A.T)
    V1 = np.asfortranarray(0.6 * np.eye(n) - 0.02 * (A[:, :5] @ A[:, :5].T))
    V2 = np.asfortranarray(0.9 * np.eye(n) - 0.01 * (A[:, 5:9] @ A[:, 5:9].T))
    ahat = rng.standard_normal(n)
    S2 = S.copy(order="F")
    S2[n // 2, n // 3] += 1e-3
    call = lambda Sx, Vx: api.calculate_a_and_vara_rcpp(geno["asciifileMt"], NA, Sx, Vx, 8.0, (L, n), ahat)
    api.drop_cache()
    if streamed:
        monkeypatch.setenv("EAGLE_HIP_MAX_RESIDENT_GB", "%.6f" % (1.6 * 1024 * 1024 / 1e9))   # marker blocks of 1,024
    monkeypatch.setenv("EAGLE_HIP_NO_SCACHE", "1")
    ref = {k: call(*k_args) for k, k_args in (("S,V1", (S, V1)), ("S,V2", (S, V2)), ("S2,V2", (S2, V2)))}
    monkeypatch.delenv("EAGLE_HIP_NO_SCACHE")
    monkeypatch.setenv("EAGLE_HIP_SCACHE_MAX_NP", "0")
    try:
        api.drop_cache()
        stats = [api.scan_operand_cache_stats()]
        got = []
        for Sx, Vx, key in ((S, V1, "S,V1"), (S, V2, "S,V2"), (S2, V2, "S2,V2"), (S2, V2, "S2,V2")):
            got.append((call(Sx, Vx), key))
            stats.append(api.scan_operand_cache_stats())
            if streamed:
                assert api.last_stream_stats()["chunks"] > 1
        d = [(b[0] - a[0], b[1] - a[1]) for a, b in zip(stats, stats[1:])]
        assert d == [(0, 0), (1, 0), (0, 1), (1, 0)], d            # filled; hit; another S detected; hit on that one
        api.calculateMMt_rcpp(geno["asciifileM"], 8.0, 4, NA, (n, L))     # (may start a background allocation that replaces the arena)
        got.append((call(S2, V1), None))
        got.append((call(S2, V2), "S2,V2"))
        api.drop_cache()
        got.append((call(S2, V2), "S2,V2"))                       # arena gone: uploaded again
        h, m = api.scan_operand_cache_stats()
        assert m == stats[-1][1]                                   # none of these was a wrong guess
        for r, key in got:
            if key:
                np.testing.assert_array_equal(r["a"], ref[key]["a"])
                np.testing.assert_array_equal(r["vara"], ref[key]["vara"])
    finally:
        api.drop_cache()


def test_mmt_of_many_individuals_comes_back_through_the_staged_download(api, tmp_path):
    """n = 6,000: the 288 MB result returns through the two pinned staging buffers in 64 MiB pieces (csrc/eagle_api.cpp, download_big),
    a path the smaller shapes above never take; the normalised form too.  Exact against the integer product."""
    n, L = 6000, 192
    rng = np.random.default_rng(41)
    Mt8 = (rng.binomial(2, rng.uniform(0.1, 0.5, size=L)[:, None], size=(L, n)) - 1).astype(np.int8)
    geno = synth.write_geno_pair(str(tmp_path), Mt8)
    G = Mt8.astype(np.float64)
    want = G.T @ G
    got = api.calculateMMt_rcpp(geno["asciifileM"], 8.0, 4, NA, (n, L))
    np.testing.assert_array_equal(got, want)
    norm, mx = api.last_mmt_normalised(n)
    assert mx == want.max()
    np.testing.assert_allclose(norm, want / mx + 0.95 * np.eye(n), rtol=1e-15, atol=1e-15)
    api.drop_cache()
