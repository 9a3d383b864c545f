"""SURVEY 8 f-4 through the C ABI (include/eagle_hip.h section 1c): the dense model algebra on the device, against
numpy / LAPACK on the host.  Opt-in entry points; rocSOLVER for the factorisations, the library's fp64 MFMA GEMM for products.
parity unpinned (the reference records no outputs); the R semantics mirrored are cited in the header."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    from eagleeverything_amd import rcpp_api
    assert rcpp_api.device_info()["arch"].startswith("gfx950")
    yield rcpp_api
    rcpp_api.close_all()


@pytest.mark.parametrize("n", [1, 7, 150, 513, 1200])
def test_sym_eig_chol2inv_inverse(n, api):
    rng = np.random.default_rng(n)
    B = rng.standard_normal((n, n + 3))
    A = B @ B.T / n + 0.5 * np.eye(n)
    w, U = api.sym_eig(A)
    w_ref = np.linalg.eigvalsh(A)[::-1]
    np.testing.assert_allclose(w, w_ref, rtol=1e-11, atol=1e-13)       # decreasing, as R's eigen()
    np.testing.assert_allclose(A @ U, U * w, rtol=0, atol=1e-10 * w[0])
    np.testing.assert_allclose(U.T @ U, np.eye(n), atol=1e-11)
    w2, none = api.sym_eig(A, only_values=True)
    assert none is None
    np.testing.assert_allclose(w2, w_ref, rtol=1e-11, atol=1e-13)
    Ai = api.chol2inv(A)
    np.testing.assert_allclose(Ai, np.linalg.inv(A), rtol=1e-9, atol=1e-11)
    assert np.array_equal(Ai, Ai.T)
    G = rng.standard_normal((n, n)) + n * np.eye(n)                    # general, well conditioned
    np.testing.assert_allclose(api.inverse(G), np.linalg.inv(G), rtol=1e-9, atol=1e-12)
    # either memory order goes in without a host-side layout change (row-major = the transpose read column-major) and a strided view is copied
    np.testing.assert_allclose(api.inverse(np.asfortranarray(G)), np.linalg.inv(G), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(api.inverse(G[::-1, ::-1]), np.linalg.inv(G[::-1, ::-1]), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(api.chol2inv(np.asfortranarray(A)), Ai, rtol=1e-12, atol=1e-14)


def test_chol2inv_and_inverse_failures(api):
    from eagleeverything_amd._lib import EagleError
    A = np.eye(5)
    A[3, 3] = -1.0
    with pytest.raises(EagleError, match="leading minor of order 4 is not positive"):
        api.chol2inv(A)
    with pytest.raises(EagleError, match="singular"):
        api.inverse(np.zeros((4, 4)))


@pytest.mark.parametrize("m,k,n", [(1, 1, 1), (150, 150, 150), (150, 3, 150), (3, 150, 1), (700, 130, 257)])
def test_matmul(m, k, n, api):
    rng = np.random.default_rng(m + 10 * k + 100 * n)
    A, B = rng.standard_normal((m, k)), rng.standard_normal((k, n))
    np.testing.assert_allclose(api.matmul(A, B), A @ B, rtol=0, atol=1e-12 * k)
    Ai = rng.integers(-9, 10, size=(m, k)).astype(np.float64)         # integer-valued: exact in fp64
    Bi = rng.integers(-9, 10, size=(k, n)).astype(np.float64)
    np.testing.assert_array_equal(api.matmul(Ai, Bi), Ai @ Bi)
    for Ax, Bx in ((np.asfortranarray(Ai), Bi), (Ai, np.asfortranarray(Bi)), (np.asfortranarray(Ai), np.asfortranarray(Bi)), (Ai[::-1], Bi[:, ::-1])):
        np.testing.assert_array_equal(api.matmul(Ax, Bx), Ax @ Bx)   # any memory order / strides


def test_mmt_sqrt_and_sqrtinv(api, golden):
    g = golden("genoDemo_150x4998")
    n = 150
    MMt = g["MMt"] / g["MMt"].max() + 0.95 * np.eye(n)                # calcMMt.R:13
    sq, inv, tr = api.mmt_sqrt_and_sqrtinv(MMt)
    np.testing.assert_allclose(sq @ sq, MMt, rtol=0, atol=1e-12)
    np.testing.assert_allclose(sq @ inv, np.eye(n), atol=1e-11)
    assert np.array_equal(sq, sq.T) and np.array_equal(inv, inv.T)
    assert int(np.trunc(tr + 1e-9)) == n and abs(tr - n) < 1e-9       # the reference's own check, :35-46
    w, U = np.linalg.eigh(MMt)
    np.testing.assert_allclose(sq, (U * np.sqrt(w)) @ U.T, rtol=0, atol=1e-12)
    # not positive definite (two identical individuals without the 0.95 ridge): the R function returns NULL
    M = g["M8"].astype(np.float64)
    M[1] = M[0]
    assert api.mmt_sqrt_and_sqrtinv(M @ M.T) is None
