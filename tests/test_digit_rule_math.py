"""The arithmetic behind the digit rule of the int8 scan (csrc/eagle_i8mfma.hip: w_scale_exp, k_slice_w, k_spectral_decide, k_gram_hi_i8),
restated in numpy on small matrices: what the device code relies on must hold exactly, whatever the GPU does.  (The device side is
compared with the same restatement in tests/test_gpu_spectral.py.)"""
import numpy as np


def _scale_exp(mx):
    f, e = np.frexp(mx)
    return int(e) - (1 if f <= 0.98 else 0)


def _digits(Q, S):
    out = []
    Q = Q.astype(np.int64).copy()
    for _ in range(S):                       # least significant first, as k_slice_w peels them
        d = np.mod(Q + 128, 256) - 128
        out.append(d)
        Q = (Q - d) >> 8
    assert not Q.any()                       # S balanced digits hold the whole number
    return out[::-1]                         # most significant first: digit s has weight 256^(S-1-s)


def test_scaled_integers_fit_the_balanced_digits():
    rng = np.random.default_rng(1)
    for S in range(1, 8):
        top = 127 * (256 ** S - 1) // 255    # the largest balanced S-digit number
        for f in np.concatenate([rng.uniform(0.5, 1.0, 200), [0.5, 0.98, 0.9800001, 0.999999]]):
            mx = f * 2.0 ** int(rng.integers(-40, 40))
            e = _scale_exp(mx)
            assert mx <= (1.96 if np.frexp(mx)[0] <= 0.98 else 1.0) * 2.0 ** e
            q = int(np.rint(np.ldexp(mx, 8 * S - e - 2)))
            assert q + 1 <= top, (S, f, q, top)          # + 1: stochastic rounding may round up once more
            for d in _digits(np.array([q, -q]), S):
                assert np.all((d >= -128) & (d <= 127))


def test_truncation_error_is_the_quadratic_form_of_the_last_digit():
    rng = np.random.default_rng(2)
    n, S = 192, 3
    W = rng.standard_normal((n, n)) * 1e-3
    Wu = np.triu(W + W.T, 1)
    e = _scale_exp(np.abs(Wu).max())
    u = 2.0 ** (e + 2 - 8 * S)
    Q = np.rint(np.ldexp(Wu, 8 * S - e - 2))
    rho = Wu / u - Q
    assert np.abs(rho).max() <= 0.5
    d = _digits(Q, S)[-1]                    # the last digit
    T = (Q - d) * u                          # what the leading S - 1 digits represent
    Ds, P = d + d.T, rho + rho.T
    H = 0.5 * u * (Ds + P)
    normH = np.abs(np.linalg.eigvalsh(H)).max()
    for _ in range(50):
        m = rng.integers(0, 3, size=n).astype(np.float64)      # a re-centred marker row: 0, 1, 2
        err = m @ Wu @ m - m @ T @ m
        assert abs(err - m @ H @ m) <= 1e-9 * abs(err) + 1e-18
        assert abs(err) <= normH * (m @ m) * (1 + 1e-12)
        assert abs(err) <= 0.5 * m.sum() ** 2 * 128.5 * u      # the worst-case bound of the same truncation


def _level1(Ds):
    return np.sqrt(float(np.abs(Ds @ Ds).sum(axis=1).max()))


def _level2(Ds, shift):
    G = Ds @ Ds
    E = G - np.diag(np.diag(G))
    hi = (E + (1 << (shift - 1))) >> shift
    lo = E - (hi << shift)
    assert np.abs(lo).max() <= 1 << (shift - 1)
    if hi.max() > 127 or hi.min() < -128:
        return None
    return np.sqrt(float(np.diag(G).max()) + 2.0 ** shift * _level1(hi) + np.sqrt(float((lo * lo).sum())))


def test_both_levels_bound_the_spectral_norm_and_the_second_is_tighter_on_random_digits():
    rng = np.random.default_rng(3)
    for n in (256, 768):
        d = np.triu(rng.integers(-128, 128, size=(n, n)), 1)
        Ds = (d + d.T).astype(np.int64)
        true = np.abs(np.linalg.eigvalsh(Ds.astype(np.float64))).max()
        shift = 8
        while shift < 23 and 127.0 * (1 << shift) < 8.0 * 5476.0 * np.sqrt(n):
            shift += 1
        b1, b2 = _level1(Ds), _level2(Ds, shift)
        assert b2 is not None and true <= b2 <= b1 and true <= b1 <= 8.0 * true, (n, true, b1, b2)
    # a structured last digit (rank one): both levels still bound it; the second may decline (high part outside int8)
    v = rng.integers(-11, 12, size=512)
    Ds = np.clip(np.outer(v, v), -128, 127).astype(np.int64)
    np.fill_diagonal(Ds, 0)
    true = np.abs(np.linalg.eigvalsh(Ds.astype(np.float64))).max()
    assert true <= _level1(Ds)
    b2 = _level2(Ds, 15)
    assert b2 is None or true <= b2
