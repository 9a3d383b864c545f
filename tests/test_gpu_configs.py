"""GPU parity tests at the shapes BASELINE.json's configs name (one card; the 8-GPU split of configs 2-4 only changes which
rank holds which marker shard, tests/test_sharded_gloo.py):

  C2  5,000 x 500,000 resident                      -> test_config_shape_single_gpu[C2]
  C3  10,000 x 1,000,000 resident                   -> tests/test_gpu_parity.py::test_config_C3_shape_single_gpu
  C4  n = 50,000, out-of-core marker blocks         -> test_config_C4_n50000_streamed (16,384 of the 5,000,000 markers: the
                                                       n-dependent part -- 80 GB of fp64 operands, 17 GB of digit slices,
                                                       int32 / index limits -- is the full size; more markers are more
                                                       blocks of the same pass)
  C5  2,000 individuals x 1,000,000 (Z-matrix run)  -> test_config_shape_single_gpu[C5] (the scan sees n = 2,000; the Z
                                                       incidence matrix only enters the host algebra, SURVEY 8a note)

Full-size cases cannot be checked against a full oracle run (the CPU port needs hours): the scan is compared with the
oracle on marker samples, MM^T through exact integer properties, the arg-max against the full tsq vector.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RTOL_DIGITS = 1e-7  # vara from the digit-slice kernels against the oracle: the level actually MEASURED on these cases (1e-8 and below), so that
# an order of magnitude of drift shows; what the certificate ENFORCES per marker is RTOL_ENFORCED = 1.8 x the budget in force (9e-7 at most; north_star: 1e-6)
RTOL_ENFORCED = 9e-7


@pytest.fixture(scope="module")
def api():
    from eagleeverything_amd import rcpp_api
    assert rcpp_api.device_info()["arch"].startswith("gfx950")
    yield rcpp_api
    rcpp_api.close_all()


@pytest.mark.parametrize("name,n,L,nsample", [("C2", 5000, 500000, 4096), ("C5", 2000, 1000000, 8192)])
def test_config_shape_single_gpu(name, n, L, nsample, api, oracle):
    import torch
    from eagleeverything_amd.sharded import DeviceShard
    sh = DeviceShard(n, L)
    sh.fill_synthetic(seed=20240601)
    gen = torch.Generator(device=sh.dev)
    gen.manual_seed(11 + n)
    A = torch.randn((n, 48), generator=gen, device=sh.dev, dtype=torch.float64) / 40.0
    S = 0.5 * torch.eye(n, dtype=torch.float64, device=sh.dev) + A @ A.T
    V = 0.6 * torch.eye(n, dtype=torch.float64, device=sh.dev) - 0.03 * (A[:, :6] @ A[:, :6].T)
    ahat = torch.randn(n, generator=gen, device=sh.dev, dtype=torch.float64)
    sh.set_operands(S, V, ahat)
    del A
    sh.mode = 1
    sh.scan()
    torch.cuda.synchronize()
    third = nsample // 4
    rows = torch.cat([torch.arange(0, 2 * third), torch.arange(L // 2 - third // 2, L // 2 + third // 2),
                      torch.arange(L - third, L)]).to(sh.dev)
    Mt_s = sh.Mt8[rows][:, :n].cpu().numpy()
    a_ref, v_ref = oracle.scan_from_i8(Mt_s, S.cpu().numpy(), V.cpu().numpy(), ahat.cpu().numpy())
    assert rows.numel() >= 4000
    np.testing.assert_allclose(sh.a[rows].cpu().numpy(), a_ref, rtol=1e-9, atol=1e-12 * np.abs(a_ref).max())
    np.testing.assert_allclose(sh.vara[rows].cpu().numpy(), v_ref, rtol=RTOL_DIGITS)
    tsq = sh.a[:L] ** 2 / sh.vara[:L]
    tsqmax, gidx, near = sh.best()
    assert gidx == int(torch.argmax(tsq)) and tsqmax == float(tsq.max())
    info = sh.certificate()
    assert info["overflow"] == 0 and info["reevaluated"] >= 1
    # the fp64 scan picks the same marker (and its value for that marker is what the certified scan reports)
    v_cert = sh.vara[gidx].item()
    sh.mode = 0
    sh.scan()
    torch.cuda.synchronize()
    assert sh.best()[1] == gidx and sh.vara[gidx].item() == v_cert
    del S, V, tsq
    sh.release_operands()
    # MM^T: exact integer properties
    c32 = sh.mmt_partial()
    MMt, mx = sh.mmt_finish(c32)
    del c32
    cols = torch.tensor([0, 1, 255, 256, n // 2, n - 1], device=sh.dev)
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


def test_config_C4_n50000_streamed(api, tmp_path, monkeypatch):
    """n = 50,000 individuals through the reference-shaped entry points with the marker files streamed in blocks
    (EAGLE_HIP_MAX_RESIDENT_GB forces the out-of-core path of BASELINE configs[3]).  Operands with a low-rank structure,
    S = s I + P P^T, V = D + U U^T, so that v = S a_hat, z_i = S m_i and vara_i = z_i^T V z_i have O(n r) closed forms in
    numpy -- the library receives them as dense 50,000 x 50,000 matrices and does the full n^3 products.  MM^T (20 GB of
    doubles back to the host): diagonal, symmetry of sampled blocks and selected columns exact."""
    import torch
    from eagleeverything_amd import synth
    from eagleeverything_amd.sharded import DeviceShard
    n, L = 50000, 16384
    gsh = DeviceShard(n, L)                       # only used to draw the genotypes on the device
    gsh.fill_synthetic(seed=4)
    Mt8 = gsh.Mt8[:L, :n].cpu().numpy()
    del gsh
    torch.cuda.empty_cache()
    geno = synth.write_geno_pair(str(tmp_path), Mt8)
    rng = np.random.default_rng(50)
    r1, r2 = 6, 5
    P = rng.standard_normal((n, r1)) / np.sqrt(n) * 0.5
    U = rng.standard_normal((n, r2)) / np.sqrt(n) * 0.7
    s, d = 0.8, rng.uniform(0.5, 1.5, size=n)
    S = (P @ P.T).T                               # symmetric: the transposed view is the column-major matrix, no copy
    S[np.diag_indices(n)] += s
    V = (U @ U.T).T
    V[np.diag_indices(n)] += d
    ahat = rng.standard_normal(n)
    monkeypatch.setenv("EAGLE_HIP_MAX_RESIDENT_GB", "0.4")
    msgs = []
    try:
        res = api.calculate_a_and_vara_rcpp(geno["asciifileMt"], np.nan, S, V, 1000.0, (L, n), ahat, quiet=False, message=msgs.append)
        idx, tsqmax, _ = api.last_scan_argmax()
    finally:
        del S, V
    assert any("streamed" in m for m in msgs)
    rows = np.r_[0:32, 3830:3850, L - 16:L]       # across the first block boundary (3,840 markers per block) and the ragged tail
    M = Mt8[rows].astype(np.float64)
    Z = s * M + (M @ P) @ P.T                     # z_i = S m_i
    a_ref = Z @ ahat
    v_ref = (Z * Z) @ d + ((Z @ U) ** 2).sum(axis=1)
    np.testing.assert_allclose(res["a"].ravel()[rows], a_ref, rtol=1e-9, atol=1e-11 * np.abs(a_ref).max())
    np.testing.assert_allclose(res["vara"].ravel()[rows], v_ref, rtol=RTOL_DIGITS)
    with np.errstate(all="ignore"):
        tsq = res["a"].ravel() ** 2 / res["vara"].ravel()
    assert idx == int(np.nanargmax(tsq)) + 1 and tsqmax == np.nanmax(tsq)
    del res, Z, M
    # MM^T streamed in marker windows of M.ascii
    msgs.clear()
    mmt = api.calculateMMt_rcpp(geno["asciifileM"], 1000.0, 8, np.nan, (n, L), quiet=False, message=msgs.append)
    assert any("streamed" in m for m in msgs)
    G = Mt8.astype(np.float32)                    # sums stay below 2^24: exact in float32
    cols = np.array([0, 1, 255, 256, 25000, 49999])
    np.testing.assert_array_equal(mmt[:, cols], (G.T @ G[:, cols]).astype(np.float64))
    np.testing.assert_array_equal(np.diagonal(mmt), (G * G).sum(axis=0, dtype=np.float64))
    for b0 in (0, 12288, 49000):
        blk = mmt[b0:b0 + 1000, 20000:21000]
        np.testing.assert_array_equal(blk, mmt[20000:21000, b0:b0 + 1000].T)
    api.drop_cache()
