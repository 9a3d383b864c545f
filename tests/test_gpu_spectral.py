"""The spectral bound of the last digit (csrc/eagle_i8mfma.hip: k_last_digit_sym, k_gram_rowabs_i8, k_spectral_decide; round 3).

A digit-slice scan may run on ONE digit fewer than the worst-case bound asks for when
    |digit error of marker i| <= specH * sum_j m'_ij^2,   specH = (u/2) (||Ds||_2 bound + (n_pad-1)/2)
keeps a typical marker inside the budget.  The tests check what makes that legitimate: the bound is an upper bound of the true
spectral norm of the residual (recomputed on the host from the folded W), every raw digit value sits inside its per-marker bound
against the fp64 kernel, the certified result and the selected marker are those of the scan without the saving, and a context whose
certificate overflowed under the saving stops taking the digit off."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
# W_kk of the "quiet" individuals in the constructed panels below: markers that are non-zero only on them have vara = q2 * 0.81 * W_kk.
QUIET_HOPELESS = 1e-6   # below what ANY digit count of the automatic rule certifies: > 2,048 flagged markers -> fp64 fallback of the block
QUIET_RESCUED = 5e-3    # above their budget under the spectral bound (one digit fewer), inside it once they get the dropped digit back


def _operands(torch, dev, n, off_scale, diag_scale, seed):
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    E = torch.randn((n, n), generator=gen, device=dev, dtype=torch.float64) * off_scale
    V = torch.diag(diag_scale * (0.8 + 0.4 * torch.rand(n, generator=gen, device=dev, dtype=torch.float64))) + 0.5 * (E + E.T)
    S = 0.9 * torch.eye(n, dtype=torch.float64, device=dev)
    ahat = torch.randn(n, generator=gen, device=dev, dtype=torch.float64)
    return S, V, ahat


def _residual_norm(Wu, e, S_cut):
    """||H||_2 of H = sym(R)/2, R = Wu - (leading S_cut - 1 digits of round(Wu 2^(8 S_cut - e - 2))), off-diagonal part only."""
    off = np.triu(Wu, 1)
    Q = np.rint(np.ldexp(off, 8 * S_cut - (e + 2)))
    d = np.mod(Q + 128, 256) - 128
    u = 2.0 ** (e + 2 - 8 * S_cut)
    R = off - (Q - d) * u
    H = 0.5 * (R + R.T)
    ev = np.linalg.eigvalsh(H)
    return max(abs(ev[0]), abs(ev[-1])), d


def test_one_digit_fewer_under_a_rigorous_spectral_bound():
    import torch
    from eagleeverything_amd.sharded import DeviceShard
    n, L = 2048, 8192
    sh = DeviceShard(n, L)
    sh.fill_synthetic(seed=11)
    # off-diagonal ~1e-3, diagonal ~2: the worst-case rule needs 4 digits with room to spare, the spectral bound certifies 3
    S, V, ahat = _operands(torch, sh.dev, n, 1e-3, 2.0, 5)
    sh.set_operands(S, V, ahat)
    sh.mode = 0
    sh.scan()
    torch.cuda.synchronize()
    v64 = sh.vara[:L].cpu().numpy().copy()
    best64 = sh.best()[:2]
    sh.mode = 1
    sh.L.eagle_dev_set_tune(sh.ctx, 29)            # the saving switched off: the worst-case digit count
    sh.scan()
    torch.cuda.synchronize()
    sh.L.eagle_dev_set_tune(sh.ctx, 0)
    S_wc = sh.vara_i8_info()[0]
    assert sh.last_specH == 0.0 and sh.last_sliced == S_wc
    v_wc = sh.vara[:L].cpu().numpy().copy()
    best_wc = sh.best()[:2]
    sh.certified = False                            # raw digit values first
    sh.scan()
    torch.cuda.synchronize()
    S_used, bound_abs, maxoff = sh.vara_i8_info()
    assert (S_used, sh.last_sliced) == (S_wc - 1, S_wc) and sh.last_specH > 0.0, (S_used, sh.last_sliced, sh.last_specH)
    raw = sh.vara[:L].cpu().numpy().copy()
    # (1) the bound is an upper bound of the true norm, and not absurdly loose
    true_norm, d = _residual_norm(sh.Wu.cpu().numpy(), sh.last_e, S_wc)
    assert true_norm <= sh.last_specH <= 12.0 * true_norm, (true_norm, sh.last_specH)
    # (2) every raw value inside its own bound against the fp64 kernel
    q2 = sh.l1[:L, 1].cpu().numpy().astype(np.float64)
    l1 = sh.l1[:L, 0].cpu().numpy().astype(np.float64)
    b = np.minimum(sh.last_specH * q2, 0.5 * l1 * l1 * 2.0 ** (sh.last_e + 1 - 8 * S_used) * (1 + 2.0 ** -8))
    err = np.abs(raw - v64)
    assert np.all(err <= b + 1e-12 * np.abs(v64)), float((err / np.maximum(b, 1e-300)).max())
    assert np.abs(raw - v64).max() <= bound_abs
    assert (sh.last_specH * q2 < 0.5 * l1 * l1 * 2.0 ** (sh.last_e + 1 - 8 * S_used)).mean() > 0.9    # the spectral term is the one that binds
    # (3) certified: inside the enforced budget, same marker and same fp64 value for it as the scan without the saving
    sh.certified = True
    sh.scan()
    torch.cuda.synchronize()
    v3 = sh.vara[:L].cpu().numpy()
    np.testing.assert_allclose(v3, v64, rtol=9e-7)
    np.testing.assert_allclose(v_wc, v64, rtol=9e-7)
    assert sh.best()[:2] == best_wc == best64
    cert = sh.certificate()
    assert cert["overflow"] == 0 and cert["flagged"] == 0 and 1 <= cert["reevaluated"] <= 16
    assert sh.last_level == 1
    # (4) the Gram row sums on the device are those of the symmetrised last digit (exact integers)
    Ds = (d + d.T).astype(np.int64)
    g = np.abs(Ds @ Ds).sum(axis=1).max()
    u = 2.0 ** (sh.last_e + 2 - 8 * S_wc)
    H_host = 0.5 * u * (np.sqrt(float(g)) + 0.5 * (sh.np_ - 1))
    assert H_host <= sh.last_specH <= H_host * (1 + 1e-12)


def _host_decision(Wu, n_pad, budget=5e-7, tight=1e-7, w_err=0.0):
    """The library's digit rule restated on the host (round 4: the tight budget is tried first, at both levels, then the default):
    (worst-case digit count, level of the spectral bound that takes one off or 0, H, budget in force)."""
    off = np.triu(Wu, 1)
    mx = np.abs(off).max()
    f, e = np.frexp(mx)
    e = int(e) - (1 if f <= 0.98 else 0)
    half_sumdiag = 0.5 * np.abs(np.diag(Wu)).sum()
    wc = lambda c: float(n_pad) ** 2 * 2.0 ** (e + 1 - 8 * c) + w_err * n_pad
    S_wc = next((c for c in range(3, 8) if wc(c) <= budget * half_sumdiag), 7)
    in_force = tight if wc(S_wc) <= tight * half_sumdiag else budget
    Q = np.rint(np.ldexp(off, 8 * S_wc - (e + 2)))
    d = (np.mod(Q + 128, 256) - 128).astype(np.int64)
    Ds = d + d.T
    G = Ds @ Ds
    u = 2.0 ** (e + 2 - 8 * S_wc)
    H1 = 0.5 * u * (np.sqrt(float(np.abs(G).sum(axis=1).max())) + 0.5 * (n_pad - 1))
    ok = lambda H, b: H > 0.0 and (H + w_err) * n_pad <= b * half_sumdiag
    if ok(H1, tight):
        return S_wc, 1, H1, tight
    # level 2: lambda_max(G) <= max G_jj + 2^s ||E_hi||_2 + ||E_lo||_F
    shift = 8
    while shift < 23 and 127.0 * (1 << shift) < 8.0 * 5476.0 * np.sqrt(n_pad):
        shift += 1
    E = G - np.diag(np.diag(G))
    hi = (E + (1 << (shift - 1))) >> shift
    lo = E - (hi << shift)
    H2 = 0.0
    if not (hi.min() < -128 or hi.max() > 127):
        g2 = np.abs(hi @ hi).sum(axis=1).max()
        normsq = float(np.diag(G).max()) + 2.0 ** shift * np.sqrt(float(g2)) + np.sqrt(float((lo * lo).sum()))
        H2 = 0.5 * u * (np.sqrt(normsq) + 0.5 * (n_pad - 1))
    if ok(H2, tight):
        return S_wc, 2, H2, tight
    if ok(H1, budget):
        return S_wc, 1, H1, budget
    if ok(H2, budget):
        return S_wc, 2, H2, budget
    return S_wc, 0, 0.0, in_force


def test_saving_is_not_taken_when_it_does_not_pay_or_is_not_allowed():
    import torch
    from eagleeverything_amd.sharded import DeviceShard
    n, L = 1024, 2048
    sh = DeviceShard(n, L)
    sh.fill_synthetic(seed=12)
    sh.mode = 1
    seen = []
    # diagonal 0.02: the worst-case count has no room for a whole digit at this n; 0.08: room for the second level of the bound only; 1.0: for the first
    for diag in (0.02, 0.08, 1.0):
        S, V, ahat = _operands(torch, sh.dev, n, 1e-3, diag, 6)
        sh.set_operands(S, V, ahat)
        sh.scan()
        torch.cuda.synchronize()
        S_used = sh.vara_i8_info()[0]
        S_wc, level, H, in_force = _host_decision(sh.Wu.cpu().numpy(), sh.np_)
        assert sh.last_sliced == S_wc and S_used == S_wc - (1 if level else 0) and sh.last_level == level, (diag, S_used, S_wc, level, sh.last_level)
        assert sh.last_budget == in_force, (diag, sh.last_budget, in_force)
        if level:
            assert H <= sh.last_specH <= H * (1 + 1e-12)
        else:
            assert sh.last_specH == 0.0
        seen.append(level)
    assert seen == [0, 2, 1], seen
    # a forced digit count and stochastic rounding never take it (operands of the second case)
    sh.nslices, sh.ws = 4, None
    sh.scan()
    torch.cuda.synchronize()
    assert sh.vara_i8_info()[0] == 4 and sh.last_specH == 0.0
    sh.nslices, sh.ws, sh.stochastic = 0, None, True
    sh.scan()
    torch.cuda.synchronize()
    sh.vara_i8_info()
    assert sh.last_specH == 0.0
    sh.stochastic = False


def test_context_stops_saving_after_a_fallback(tmp_path):
    """Markers whose quadratic form is far below q2 * mean(W_kk) (all their non-zero genotypes on individuals with a tiny W_kk) are
    outside what the digits certify to the budget, with or without the dropped digit: more than 2,048 of them overflow the
    re-evaluation buffer, the block is redone in fp64 (results still right), and the context keeps the worst-case digit count from
    then on."""
    from eagleeverything_amd import rcpp_api as api, synth
    from oracle import oracle_c
    oracle_c.build()
    n, L = 2048, 6144
    rng = np.random.default_rng(3)
    Mt8 = (rng.binomial(2, rng.uniform(0.1, 0.5, size=L)[:, None], size=(L, n)) - 1).astype(np.int8)
    quiet = np.arange(n) < n // 2                      # individuals with a tiny diagonal entry of W and no off-diagonal ones
    odd = np.arange(L) % 2 == 1
    Mt8[np.ix_(odd, ~quiet)] = 0                       # 3,072 markers are non-zero only on the quiet individuals (0 = their majority genotype)
    E = rng.standard_normal((n, n)) * 2e-4
    E[quiet, :] = 0.0
    E[:, quiet] = 0.0
    V = np.diag(np.where(quiet, QUIET_HOPELESS, 1.0) * rng.uniform(0.8, 1.2, size=n)) + 0.5 * (E + E.T)
    S = 0.9 * np.eye(n)
    ahat = rng.standard_normal(n)
    geno = synth.write_geno_pair(str(tmp_path), Mt8)
    ref = oracle_c.calculate_a_and_vara_rcpp(geno["asciifileMt"], np.nan, S, V, 8.0, (L, n), ahat)
    try:
        api.set_scan_mode(1)
        api.set_scan_budget(5e-7)                      # re-arms the saving
        r1 = api.calculate_a_and_vara_rcpp(geno["asciifileMt"], np.nan, S, V, 8.0, (L, n), ahat)
        used1, cut1, H1 = api.last_scan_digits()
        nre1, nfl1, fell1 = api.last_scan_certificate()
        assert (used1, cut1) == (cut1 - 1, cut1) and H1 > 0.0 and fell1 and nfl1 > 2048, (used1, cut1, H1, nre1, nfl1, fell1)
        np.testing.assert_allclose(r1["vara"].ravel(), ref["vara"].ravel(), rtol=9e-7, atol=1e-12 * np.abs(ref["vara"]).max())
        r2 = api.calculate_a_and_vara_rcpp(geno["asciifileMt"], np.nan, S, V, 8.0, (L, n), ahat)
        used2, cut2, H2 = api.last_scan_digits()
        assert used2 == cut2 == cut1 and H2 == 0.0
        np.testing.assert_allclose(r2["vara"].ravel(), ref["vara"].ravel(), rtol=9e-7, atol=1e-12 * np.abs(ref["vara"]).max())
        assert api.last_scan_argmax()[0] == oracle_c.tsq_argmax(ref["a"], ref["vara"])[1]
        api.set_scan_budget(5e-7)
        api.calculate_a_and_vara_rcpp(geno["asciifileMt"], np.nan, S, V, 8.0, (L, n), ahat)
        assert api.last_scan_digits()[2] > 0.0         # re-armed
    finally:
        api.set_scan_budget(0)                         # the default policy again (1e-7 first, then 5e-7) for whoever shares the context
        api.drop_cache()


def test_device_resident_driver_stops_saving_after_a_fallback():
    """The same rule for callers of the device-resident entry points (sharded.py: bench.py --gpus N): best() sees the certificate."""
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
    V = np.diag(np.where(quiet, QUIET_HOPELESS, 1.0) * rng.uniform(0.8, 1.2, size=n)) + 0.5 * (E + E.T)
    sh = DeviceShard(n, L)
    sh.Mt8[:L, :n] = torch.from_numpy(Mt8).to(sh.dev)
    sh.set_operands(0.9 * np.eye(n), V, rng.standard_normal(n))
    try:
        sh.L.eagle_dev_set_spectral(sh.ctx, 1)
        sh.mode = 0
        sh.scan()
        best64 = sh.best()[:2]
        sh.mode = 1
        sh.scan()
        b1 = sh.best()[:2]
        S1 = sh.vara_i8_info()[0]
        c1 = sh.certificate()
        assert sh.last_specH > 0.0 and S1 == sh.last_sliced - 1 and c1["overflow"] == 1 and c1["flagged"] > 2048
        sh.scan()
        b2 = sh.best()[:2]
        S2 = sh.vara_i8_info()[0]
        assert sh.last_specH == 0.0 and S2 == sh.last_sliced == S1 + 1
        assert b1 == b2 == best64
    finally:
        sh.L.eagle_dev_set_spectral(sh.ctx, 1)


def _panel(n, L, quiet_diag, seed=3):
    rng = np.random.default_rng(seed)
    Mt8 = (rng.binomial(2, rng.uniform(0.1, 0.5, size=L)[:, None], size=(L, n)) - 1).astype(np.int8)
    quiet = np.arange(n) < n // 2
    odd = np.arange(L) % 2 == 1
    Mt8[np.ix_(odd, ~quiet)] = 0
    E = rng.standard_normal((n, n)) * 2e-4
    E[quiet, :] = 0.0
    E[:, quiet] = 0.0
    V = np.diag(np.where(quiet, quiet_diag, 1.0) * rng.uniform(0.8, 1.2, size=n)) + 0.5 * (E + E.T)
    return Mt8, odd, 0.9 * np.eye(n), V, rng.standard_normal(n)


def test_markers_outside_the_spectral_bound_get_the_dropped_digit_back():
    """eagle_dev_vara_i8_extend: 3,072 markers fail their budget under the spectral bound; each gets the last digit's term exactly and then
    carries, bit for bit, the value of the scan on all cut digits -- no fp64 fallback, nobody else touched."""
    import torch
    from eagleeverything_amd.sharded import DeviceShard
    n, L = 2048, 6144
    Mt8, odd, S, V, ahat = _panel(n, L, QUIET_RESCUED)
    sh = DeviceShard(n, L)
    sh.Mt8[:L, :n] = torch.from_numpy(Mt8).to(sh.dev)
    sh.set_operands(S, V, ahat)
    sh.L.eagle_dev_set_spectral(sh.ctx, 1)
    sh.mode = 0
    sh.scan()
    v64 = sh.vara[:L].cpu().numpy().copy()
    best64 = sh.best()[:2]
    sh.mode = 1
    sh.certified = False
    sh.L.eagle_dev_set_tune(sh.ctx, 29)
    sh.scan()
    torch.cuda.synchronize()
    sh.L.eagle_dev_set_tune(sh.ctx, 0)
    S_wc = sh.vara_i8_info()[0]
    v_all_digits = sh.vara[:L].cpu().numpy().copy()
    sh.extend = False
    sh.scan()
    torch.cuda.synchronize()
    assert sh.vara_i8_info()[0] == S_wc - 1 and sh.last_specH > 0.0
    v_fewer = sh.vara[:L].cpu().numpy().copy()
    sh.extend = True
    sh.scan()
    torch.cuda.synchronize()
    v_ext = sh.vara[:L].cpu().numpy().copy()
    changed = v_ext != v_fewer
    assert changed.sum() > 2048 and not changed[~odd].any() and changed[odd].mean() > 0.9    # the constructed markers, nobody else
    np.testing.assert_array_equal(v_ext[changed], v_all_digits[changed])                       # the value of the scan on all cut digits
    np.testing.assert_array_equal(v_ext[~changed], v_fewer[~changed])
    sh.certified = True
    sh.scan()
    b = sh.best()[:2]
    cert = sh.certificate()
    assert cert["overflow"] == 0 and cert["flagged"] <= 64 and cert["reevaluated"] <= 80, cert   # a few stragglers go to the fp64 kernel
    assert b == best64
    np.testing.assert_allclose(sh.vara[:L].cpu().numpy(), v64, rtol=9e-7)
    sh.scan()
    sh.best()
    assert sh.vara_i8_info()[0] == S_wc - 1 and sh.last_specH > 0.0                          # no fallback: the saving stays
    # a block with more qualifying markers than the compact image holds is worked in several passes, not given up --
    # with room for 768 of the 3,072 per pass the extended values are the very same bits
    import os
    os.environ["EAGLE_HIP_EXT_CAP"] = "768"
    try:
        sh.ws = None                                    # the workspace layout follows the cap
        sh.certified = False
        sh.scan()
        torch.cuda.synchronize()
        np.testing.assert_array_equal(sh.vara[:L].cpu().numpy(), v_ext)
    finally:
        os.environ.pop("EAGLE_HIP_EXT_CAP")
        sh.ws = None
        sh.certified = True


def test_extension_is_a_per_marker_decision_resident_and_streamed(tmp_path, monkeypatch):
    """Through the reference-shaped call: the panel resident and streamed in 256-marker blocks returns the same bits."""
    from eagleeverything_amd import rcpp_api as api, synth
    n, L = 2048, 3072
    Mt8, odd, S, V, ahat = _panel(n, L, QUIET_RESCUED)
    geno = synth.write_geno_pair(str(tmp_path), Mt8)
    try:
        api.set_scan_mode(1)
        api.set_scan_budget(5e-7)
        r1 = api.calculate_a_and_vara_rcpp(geno["asciifileMt"], np.nan, S, V, 8.0, (L, n), ahat)
        d1, c1 = api.last_scan_digits(), api.last_scan_certificate()
        assert d1[0] == d1[1] - 1 and d1[2] > 0.0 and not c1[2] and c1[1] <= 64, (d1, c1)
        api.drop_cache()
        monkeypatch.setenv("EAGLE_HIP_MAX_RESIDENT_GB", "%.6f" % (2.0 * 256 * 2048 / 1e9))
        r2 = api.calculate_a_and_vara_rcpp(geno["asciifileMt"], np.nan, S, V, 8.0, (L, n), ahat)
        assert api.last_stream_stats()["chunks"] >= 12
        np.testing.assert_array_equal(r2["vara"], r1["vara"])
        np.testing.assert_array_equal(r2["a"], r1["a"])
        assert api.last_scan_digits() == d1 and api.last_scan_certificate() == c1
    finally:
        monkeypatch.delenv("EAGLE_HIP_MAX_RESIDENT_GB", raising=False)
        api.set_scan_budget(0)
        api.drop_cache()


def test_extension_over_two_contexts_sharing_one_card(tmp_path):
    """The shards of a multi-device context (one card named twice): every device takes the same digit decision from W alone and gives
    the same markers the dropped digit back -- the bits of the single-device call."""
    from eagleeverything_amd import rcpp_api as api, synth
    n, L = 2048, 3072
    Mt8, odd, S, V, ahat = _panel(n, L, QUIET_RESCUED)
    geno = synth.write_geno_pair(str(tmp_path), Mt8)
    try:
        api.set_scan_mode(1)
        api.set_scan_budget(5e-7)
        r1 = api.calculate_a_and_vara_rcpp(geno["asciifileMt"], np.nan, S, V, 8.0, (L, n), ahat)
        d1, b1 = api.last_scan_digits(), api.last_scan_argmax()[:2]
        api.drop_cache()
        api.set_scan_mode(1, device=(0, 0))
        api.set_scan_budget(5e-7, device=(0, 0))
        r2 = api.calculate_a_and_vara_rcpp(geno["asciifileMt"], np.nan, S, V, 8.0, (L, n), ahat, device=(0, 0))
        np.testing.assert_array_equal(r2["vara"], r1["vara"])
        np.testing.assert_array_equal(r2["a"], r1["a"])
        assert api.last_scan_digits(device=(0, 0)) == d1 and d1[0] == d1[1] - 1
        assert api.last_scan_argmax(device=(0, 0))[:2] == b1
        assert not api.last_scan_certificate(device=(0, 0))[2]
    finally:
        api.set_scan_budget(0)
        api.set_scan_budget(0, device=(0, 0))
        api.drop_cache()
        api.drop_cache(device=(0, 0))
