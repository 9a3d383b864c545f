"""Several GPUs behind the reference-shaped C ABI (eagle_open_devices; the reference's unused AM(..., ngpu) hook,
E/R/AM.R:185-196): every call shards the file's markers over the devices of the context, one worker thread per device.

The build box has ONE card, so the device list names it two / three times: the same code path -- marker ranges, worker
threads, rendezvous, partial MM^T packed and summed, lower bounds of the shards' maxima exchanged for the certificate,
per-device results merged -- with the device-copy sum standing in for ncclReduce and W computed on every device instead of
shared through ncclAllGather (a transfer over RCCL needs distinct devices; the calls themselves run on a communicator of ONE
device in the last test).
Everything must come back bit for bit as from a single-device context.
"""
import ctypes as C

import numpy as np
import pytest

from eagleeverything_amd import synth

pytestmark = pytest.mark.gpu
NA = np.nan


@pytest.fixture(scope="module")
def api():
    from eagleeverything_amd import rcpp_api
    assert rcpp_api.device_info()["arch"].startswith("gfx950")
    yield rcpp_api
    rcpp_api.close_all()


def _all_calls(api, geno, n, L, S, V, ahat, P, y, sel, device):
    out = {}
    out["mmt"] = api.calculateMMt_rcpp(geno["asciifileM"], 8.0, 4, NA, (n, L), device=device)
    out["mmt_masked"] = api.calculateMMt_rcpp(geno["asciifileM"], 8.0, 4, sel, (n, L), device=device)
    for mode in (1, 0):
        api.set_scan_mode(mode, device=device)
        r = api.calculate_a_and_vara_rcpp(geno["asciifileMt"], NA, S, V, 8.0, (L, n), ahat, device=device)
        out["a%d" % mode], out["vara%d" % mode] = r["a"], r["vara"]
        out["best%d" % mode] = api.last_scan_argmax(device=device)
    api.set_scan_mode(1, device=device)
    r = api.calculate_a_and_vara_rcpp(geno["asciifileMt"], sel, S, V, 8.0, (L, n), ahat, device=device)
    out["a_masked"], out["vara_masked"] = r["a"], r["vara"]
    out["best_masked"] = api.last_scan_argmax(device=device)
    out["ar"] = api.calculate_reduced_a_rcpp(geno["asciifileMt"], 0.7, P, y, 8.0, (n, L), sel, device=device)
    out["geno"] = [api.extract_geno_rcpp(geno["asciifileM"], 8.0, c, (n, L), device=device) for c in (0, L // 2 + 3, L - 1)]
    return out


@pytest.mark.parametrize("devices", [(0, 0), (0, 0, 0)])
def test_contexts_sharing_one_card_reproduce_the_single_device_results(devices, api, oracle, tmp_path):
    from eagleeverything_amd import _lib
    n, L = 700, 6001
    rng = np.random.default_rng(3)
    Mt8 = synth.genotypes_marker_major(n, L, seed=21)
    A = rng.standard_normal((n, 40)) / 6.0
    S = np.eye(n) + A @ A.T
    V = 0.7 * np.eye(n) - 0.03 * (A[:, :3] @ A[:, :3].T)
    ahat = rng.standard_normal(n)
    # the strongest marker twice, in the first and in the last shard: an exact tie across devices, the smaller global index wins
    a0, v0 = oracle.scan_from_i8(Mt8, S, V, ahat)
    t = int(np.argmax(a0 ** 2 / v0))
    top = Mt8[t].copy()
    Mt8[t] = 0
    Mt8[40] = top
    Mt8[5000] = top
    geno = synth.write_geno_pair(str(tmp_path), Mt8)
    P = 0.3 * np.eye(n) + 0.01 * (A @ A.T)
    y = rng.standard_normal((n, 1))
    sel = np.array([7.0, 3000.0, 5999.0])     # masked markers in different shards
    one = _all_calls(api, geno, n, L, S, V, ahat, P, y, sel, device=0)
    api.drop_cache(device=0)
    many = _all_calls(api, geno, n, L, S, V, ahat, P, y, sel, device=devices)
    assert _lib.load().eagle_device_count(api.context(devices)) == len(devices)
    G = Mt8.astype(np.float64)
    np.testing.assert_array_equal(one["mmt"], G.T @ G)
    for key in one:
        if key == "geno":
            for x, z in zip(one[key], many[key]):
                np.testing.assert_array_equal(x, z)
        elif key.startswith("best"):
            assert one[key][:2] == many[key][:2], key
        else:
            np.testing.assert_array_equal(one[key], many[key], err_msg=key)
    # and both agree with the oracle
    ref = oracle.calculate_a_and_vara_rcpp(geno["asciifileMt"], NA, S, V, 8.0, (L, n), ahat)
    np.testing.assert_allclose(many["vara1"].ravel(), ref["vara"].ravel(), rtol=9e-7)
    assert many["best1"][0] == oracle.tsq_argmax(ref["a"], ref["vara"])[1] == 41
    # certification counters are summed over the devices
    nre, nfl, fell = C.c_long(), C.c_long(), C.c_int()
    api.calculate_a_and_vara_rcpp(geno["asciifileMt"], NA, S, V, 8.0, (L, n), ahat, device=devices)
    assert _lib.load().eagle_last_scan_certificate(api.context(devices), C.byref(nre), C.byref(nfl), C.byref(fell)) == 0
    assert nre.value >= 2 and fell.value == 0   # the tied pair at least
    api.drop_cache(device=devices)


def test_spectral_scan_over_two_contexts(api, tmp_path):
    """The opt-in scan in the eigenbasis of MM^T (header section 1d) over a multi-device context: Z shards by markers, no
    exchange step; a and vara bit for bit as from one device, masking rule included."""
    n, L = 300, 5000
    rng = np.random.default_rng(12)
    Mt8 = synth.genotypes_marker_major(n, L, seed=5)
    geno = synth.write_geno_pair(str(tmp_path), Mt8)
    G = Mt8.astype(np.float64)
    K = G.T @ G
    K = K / K.max() + 0.95 * np.eye(n)
    lam, U = np.linalg.eigh(K)
    X = np.column_stack([np.ones(n), Mt8[[10, 4000]].T.astype(np.float64)])
    y = rng.standard_normal(n)
    sel = np.array([3.0, 4500.0])
    out = {}
    for dev in (0, (0, 0)):
        api.spectral_prepare(geno["asciifileMt"], (L, n), U, 8.0, device=dev)
        out[dev] = (api.spectral_scan(lam, U.T @ X, U.T @ y, 0.8, 0.6, L, device=dev),
                    api.spectral_scan(lam, U.T @ X, U.T @ y, 0.8, 0.6, L, selected_loci=sel, device=dev))
        api.drop_cache(device=dev)
    for a, b in zip(out[0], out[(0, 0)]):
        np.testing.assert_array_equal(a["a"], b["a"])
        np.testing.assert_array_equal(a["vara"], b["vara"])
    masked = out[(0, 0)][1]
    assert masked["a"][3, 0] == 0.0 and masked["vara"][4500, 0] == 0.0
    # against the definition: a_i = varG m_i^T P y, vara_i = varG^2 m_i^T P m_i
    H = 0.8 * np.eye(n) + 0.6 * K
    Hi = np.linalg.inv(H)
    P = Hi - Hi @ X @ np.linalg.solve(X.T @ Hi @ X, X.T @ Hi)
    rows = np.r_[0:50, 2500:2600, L - 50:L]
    rows = rows[~np.isin(rows, [10, 4000])]            # markers in the model: masked values (a = vara = 0)
    M = G[rows]
    np.testing.assert_allclose(out[(0, 0)][0]["a"].ravel()[rows], 0.6 * (M @ (P @ y)), rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(out[(0, 0)][0]["vara"].ravel()[rows], 0.36 * np.einsum("ij,jk,ik->i", M, P, M), rtol=1e-8)
    assert out[0][0]["a"][10, 0] == 0.0 and out[0][0]["vara"][4000, 0] == 0.0


def test_multi_device_streamed_shards(api, oracle, tmp_path, monkeypatch):
    """Shards that may not stay resident are streamed per device; MM^T, a, vara and the selected marker come back bit for bit as
    from one device holding the file (the blocks' certification uses one lower bound over all blocks of all devices)."""
    n, L = 520, 9000
    rng = np.random.default_rng(8)
    Mt8 = synth.genotypes_marker_major(n, L, seed=33)
    geno = synth.write_geno_pair(str(tmp_path), Mt8)
    A = rng.standard_normal((n, 30)) / 6.0
    S = np.eye(n) + A @ A.T
    V = 0.7 * np.eye(n) - 0.03 * (A[:, :3] @ A[:, :3].T)
    ahat = rng.standard_normal(n)
    one = api.calculate_a_and_vara_rcpp(geno["asciifileMt"], NA, S, V, 8.0, (L, n), ahat, device=0)
    best_one = api.last_scan_argmax(device=0)
    mmt_one = api.calculateMMt_rcpp(geno["asciifileM"], 8.0, 4, NA, (n, L), device=0)
    monkeypatch.setenv("EAGLE_HIP_MAX_RESIDENT_GB", "0.0008")   # 3072 x 768 bytes per shard do not fit: two blocks each
    dev = (0, 0)
    api.drop_cache(device=dev)
    many = api.calculate_a_and_vara_rcpp(geno["asciifileMt"], NA, S, V, 8.0, (L, n), ahat, device=dev)
    best_many = api.last_scan_argmax(device=dev)
    mmt_many = api.calculateMMt_rcpp(geno["asciifileM"], 8.0, 4, NA, (n, L), device=dev)
    monkeypatch.delenv("EAGLE_HIP_MAX_RESIDENT_GB")
    np.testing.assert_array_equal(mmt_one, mmt_many)
    np.testing.assert_array_equal(one["a"], many["a"])
    np.testing.assert_array_equal(one["vara"], many["vara"])   # one lower bound over every block of every device
    assert best_one[0] == best_many[0] and best_one[1] == best_many[1]
    api.drop_cache(device=dev)


def test_exchange_steps_on_an_rccl_communicator_of_one(api, oracle, tmp_path, monkeypatch):
    """EAGLE_HIP_COLLECTIVES=rccl with ONE device: the library dlopen()s RCCL, builds a communicator of one rank and runs the
    exchange steps of the multi-device path on it -- ncclReduce of the packed int32 tiles, W's rows + ncclAllGather, the
    rendezvous -- which is everything of the RCCL leg a one-GPU box can execute (the calls, their types and stream order; not a
    transfer between devices).  Against the plain single-device context: MM^T exact, a exact, vara to rounding (W's row-block
    form sums in another order than the upper-tile form), same marker."""
    n, L = 700, 3000   # np = 768 = 6 row tiles of 128
    rng = np.random.default_rng(5)
    Mt8 = synth.genotypes_marker_major(n, L, seed=23)
    A = rng.standard_normal((n, 40)) / 6.0
    S = np.eye(n) + A @ A.T
    V = 0.7 * np.eye(n) - 0.03 * (A[:, :3] @ A[:, :3].T)
    ahat = rng.standard_normal(n)
    geno = synth.write_geno_pair(str(tmp_path), Mt8)
    P = 0.3 * np.eye(n) + 0.01 * (A @ A.T)
    y = rng.standard_normal((n, 1))
    sel = np.array([7.0, 1500.0, 2999.0])
    one = _all_calls(api, geno, n, L, S, V, ahat, P, y, sel, device=0)
    api.drop_cache(device=0)
    monkeypatch.setenv("EAGLE_HIP_COLLECTIVES", "rccl")
    forced = _all_calls(api, geno, n, L, S, V, ahat, P, y, sel, device=(0,))   # opened now, with the variable set
    monkeypatch.delenv("EAGLE_HIP_COLLECTIVES")
    for key in one:
        if key == "geno":
            for x, z in zip(one[key], forced[key]):
                np.testing.assert_array_equal(x, z)
        elif key.startswith("best"):
            assert one[key][0] == forced[key][0], key
        elif key.startswith("vara"):
            np.testing.assert_allclose(forced[key], one[key], rtol=1e-10, err_msg=key)
        else:
            np.testing.assert_array_equal(one[key], forced[key], err_msg=key)
    api.drop_cache(device=(0,))


def test_a_collective_that_cannot_be_enqueued_aborts_the_communicators_instead_of_hanging(api, tmp_path, monkeypatch):
    """a device that fails at the enqueue of a collective, after the rendezvous that precedes it, must not leave its peers
    blocked inside theirs.  EAGLE_HIP_FAULT makes the named collective report a failure (a communicator of ONE rank is what this box
    has): the call returns the error, the communicators are aborted, and the context carries on with the host-staged exchange."""
    from eagleeverything_amd._lib import EagleError
    n, L = 300, 1200
    rng = np.random.default_rng(8)
    Mt8 = synth.genotypes_marker_major(n, L, seed=29)
    A = rng.standard_normal((n, 20)) / 6.0
    S = np.eye(n) + A @ A.T
    V = 0.7 * np.eye(n) - 0.03 * (A[:, :3] @ A[:, :3].T)
    ahat = rng.standard_normal(n)
    geno = synth.write_geno_pair(str(tmp_path), Mt8)
    G = Mt8.astype(np.float64)
    monkeypatch.setenv("EAGLE_HIP_COLLECTIVES", "rccl")
    dev = (0,)
    ok = api.calculate_a_and_vara_rcpp(geno["asciifileMt"], NA, S, V, 8.0, (L, n), ahat, device=dev)   # RCCL leg alive (or already retired)
    monkeypatch.setenv("EAGLE_HIP_FAULT", "allgather")
    try:
        api.calculate_a_and_vara_rcpp(geno["asciifileMt"], NA, S, V, 8.0, (L, n), ahat, device=dev)
        raised = False
    except EagleError as e:
        raised = "ncclAllGather" in str(e)
    monkeypatch.delenv("EAGLE_HIP_FAULT")
    # either the fault fired (RCCL leg was alive) or an earlier test of this module had retired it already: both leave a working context
    again = api.calculate_a_and_vara_rcpp(geno["asciifileMt"], NA, S, V, 8.0, (L, n), ahat, device=dev)
    np.testing.assert_array_equal(again["a"], ok["a"])
    np.testing.assert_allclose(again["vara"], ok["vara"], rtol=1e-10)
    monkeypatch.setenv("EAGLE_HIP_FAULT", "reduce")
    mmt = api.calculateMMt_rcpp(geno["asciifileM"], 8.0, 4, NA, (n, L), device=dev)     # the RCCL leg is retired: no collective is issued
    monkeypatch.delenv("EAGLE_HIP_FAULT")
    np.testing.assert_array_equal(mmt, G.T @ G)
    assert raised
    api.drop_cache(device=dev)


def test_int8_w_is_replicated_bit_for_bit_and_every_device_keeps_its_copy_of_S(api, oracle, tmp_path, monkeypatch):
    """Round 4: with W on the int8 engine every device of a multi-device context forms the WHOLE W from exact
    digit slices -- a deterministic function of S and V, so no rows are shared, no all-gather runs, and the arrays are bit for bit the
    single-device ones with RCCL in the context as well as without; each device's copy of S is cached and verified (hits on every
    sub-context), and a changed S is picked up by all of them (the restart of a device never passes a rendezvous twice)."""
    n, L = 700, 3000
    rng = np.random.default_rng(6)
    Mt8 = synth.genotypes_marker_major(n, L, seed=31)
    A = rng.standard_normal((n, 40)) / 6.0
    S = np.eye(n) + A @ A.T
    V = 0.7 * np.eye(n) - 0.03 * (A[:, :3] @ A[:, :3].T)
    ahat = rng.standard_normal(n)
    geno = synth.write_geno_pair(str(tmp_path), Mt8)
    ref = oracle.calculate_a_and_vara_rcpp(geno["asciifileMt"], NA, S, V, 8.0, (L, n), ahat)
    monkeypatch.setenv("EAGLE_HIP_W_MODE", "2")          # contexts opened from here on form W on the int8 engine at any size
    api.close_all()
    try:
        one = api.calculate_a_and_vara_rcpp(geno["asciifileMt"], NA, S, V, 8.0, (L, n), ahat, device=0)
        assert api.last_w_info(device=0)["int8"] == 1
        best_one = api.last_scan_argmax(device=0)
        np.testing.assert_allclose(one["vara"], ref["vara"], rtol=1e-7)
        for devs in ((0, 0), (0, 0, 0)):
            h0, m0 = api.scan_operand_cache_stats(device=devs)
            for call in range(3):
                r = api.calculate_a_and_vara_rcpp(geno["asciifileMt"], NA, S, V, 8.0, (L, n), ahat, device=devs)
                np.testing.assert_array_equal(r["a"], one["a"])
                np.testing.assert_array_equal(r["vara"], one["vara"])
                assert api.last_scan_argmax(device=devs)[:2] == best_one[:2]
            h1, m1 = api.scan_operand_cache_stats(device=devs)
            assert (h1 - h0, m1 - m0) == (2 * len(devs), 0)            # calls 2 and 3: every sub-context verified its cached copy
            S2 = S + 1e-3 * np.eye(n)                                   # another S: every device starts over with it, once
            r2 = api.calculate_a_and_vara_rcpp(geno["asciifileMt"], NA, S2, V, 8.0, (L, n), ahat, device=devs)
            ref2 = oracle.calculate_a_and_vara_rcpp(geno["asciifileMt"], NA, S2, V, 8.0, (L, n), ahat)
            np.testing.assert_allclose(r2["vara"], ref2["vara"], rtol=1e-7)
            h2, m2 = api.scan_operand_cache_stats(device=devs)
            assert (h2 - h1, m2 - m1) == (0, len(devs))
            api.set_scan_mode(0, device=devs)                           # fp64 scan, changed S again, no RCCL: ADVICE r3 (a deferred restart hung)
            r3 = api.calculate_a_and_vara_rcpp(geno["asciifileMt"], NA, S, V, 8.0, (L, n), ahat, device=devs)
            api.set_scan_mode(1, device=devs)
            np.testing.assert_allclose(r3["vara"], ref["vara"], rtol=1e-9)
            api.drop_cache(device=devs)
        # with RCCL in the context (a communicator of one rank is what this box has): still replicated, still the same bits
        monkeypatch.setenv("EAGLE_HIP_COLLECTIVES", "rccl")
        api.close_all()
        forced = api.calculate_a_and_vara_rcpp(geno["asciifileMt"], NA, S, V, 8.0, (L, n), ahat, device=(0,))
        assert api.last_w_info(device=(0,))["int8"] == 1
        np.testing.assert_array_equal(forced["vara"], one["vara"])
        np.testing.assert_array_equal(forced["a"], one["a"])
        forced2 = api.calculate_a_and_vara_rcpp(geno["asciifileMt"], NA, S, V, 8.0, (L, n), ahat, device=(0,))
        np.testing.assert_array_equal(forced2["vara"], one["vara"])
        assert api.scan_operand_cache_stats(device=(0,))[0] >= 1
    finally:
        monkeypatch.delenv("EAGLE_HIP_COLLECTIVES", raising=False)
        monkeypatch.delenv("EAGLE_HIP_W_MODE", raising=False)
        api.close_all()
