"""Host logic of W = S (V S) on the int8 engine (csrc/eagle_w8.hip), no GPU: the work lists cover every (tile, accumulation group) of a
product exactly once -- both engines, full and upper-triangular, whole and in row panels -- the groups hold every digit pair of a
configuration once and never more pairs than an int32 accumulator takes, and the Frobenius bound of a product holds against an exact
emulation of the digit arithmetic in numpy (the bound's formula is restated nowhere: the library's own function is called)."""
import ctypes as C

import numpy as np
import pytest

from eagleeverything_amd import _lib

KMAX = 6


class W8Stats(C.Structure):
    _fields_ = [("maxd", C.c_double), ("fro2", C.c_double), ("es2", C.c_double), ("wdsum", C.c_double), ("phi2", C.c_double * KMAX),
                ("asym", C.c_double), ("bad", C.c_int), ("pad", C.c_int)]


class W8Group(C.Structure):
    _fields_ = [("level", C.c_int), ("npairs", C.c_int), ("p", C.c_ubyte * 8), ("q", C.c_ubyte * 8)]


def _work_list(nt, rt0, rt1, upper, piped, k, T, maxp):
    L = _lib.load()
    cap = 1 << 22
    out = (C.c_uint * cap)()
    groups = (W8Group * 64)()
    maxlen, ng = C.c_int(), C.c_int()
    assert L.eagle_w8_host_work_list(nt, rt0, rt1, int(upper), int(piped), k, T, maxp, out, cap, C.byref(maxlen), groups, C.byref(ng)) == 0
    wl = np.frombuffer(out, dtype=np.uint32, count=8 * maxlen.value).reshape(8, maxlen.value).copy()
    return wl, [groups[g] for g in range(ng.value)]


@pytest.mark.parametrize("piped", [False, True])
@pytest.mark.parametrize("upper", [False, True])
@pytest.mark.parametrize("nt,k,T,maxp", [(40, 5, 6, 12), (6, 6, 7, 2), (1, 4, 5, 8), (9, 6, 12, 3)])
def test_work_list_covers_every_tile_of_every_group_once(nt, k, T, maxp, upper, piped):
    ti_rows, tj_rows = 256, (384 if piped else 256)
    ntj = (nt * 256 + tj_rows - 1) // tj_rows
    for rt0, rt1 in ((0, nt), (0, (nt + 1) // 2), ((nt + 1) // 2, nt)):
        if rt0 >= rt1:
            continue
        wl, groups = _work_list(nt, rt0, rt1, upper, piped, k, T, maxp)
        # groups: every pair (p, q) <= k with p + q <= T exactly once, level = p + q, at most maxp pairs each, levels ascending
        pairs = [(g.p[i] + 1, g.q[i] + 1, g.level) for g in groups for i in range(g.npairs)]
        want = [(p, q) for p in range(1, k + 1) for q in range(1, k + 1) if p + q <= T]
        assert sorted((p, q) for p, q, _ in pairs) == sorted(want)
        assert all(p + q == lv for p, q, lv in pairs) and all(1 <= g.npairs <= min(maxp, 8) for g in groups)
        assert [g.level for g in groups] == sorted(g.level for g in groups)
        items = wl[wl != 0xFFFFFFFF]
        seen = {}
        for x in items:
            key = (int(x >> 20), int((x >> 8) & 0xFFF), int(x & 0xFF))
            seen[key] = seen.get(key, 0) + 1
        assert all(c == 1 for c in seen.values())
        need = {(i, j, g) for i in range(rt0, rt1) for j in range(ntj) for g in range(len(groups))
                if not upper or j * tj_rows + tj_rows - 1 >= i * ti_rows}
        assert set(seen) == need
        if upper:   # every element on or above the diagonal lies in a listed tile
            for i in range(rt0 * 256, rt1 * 256, 97):
                for j in range(i, nt * 256, 131):
                    assert (i // ti_rows, j // tj_rows, 0) in seen
        # the eight XCD lists carry about the same work (what the longest-first deal is for)
        cost = np.array([[groups[int(x & 0xFF)].npairs if x != 0xFFFFFFFF else 0 for x in row] for row in wl]).sum(axis=1)
        if len(need) >= 64 * len(groups):
            assert cost.max() <= 1.15 * cost.mean() + 8 * max(g.npairs for g in groups)


def _slices(F, k=KMAX):
    mx = np.abs(F).max(axis=1)
    f, e = np.frexp(mx)
    e = np.where(mx > 0, np.where(f <= 0.98, e - 1, e), 0).astype(np.int64)
    Q = np.rint(np.ldexp(F, (8 * k - e - 2)[:, None])).astype(np.int64)
    Q[mx == 0] = 0
    digs = []
    for _ in range(k):
        d = ((Q + 128) & 255) - 128
        Q = (Q - d) >> 8
        digs.append(d.astype(np.float64))
    assert np.all(Q == 0)
    return e, digs[::-1], mx


def _stats(M):
    d = np.diag(M).copy()
    F = M - np.diag(d)
    e, digs, mx = _slices(F)
    st = W8Stats()
    st.maxd = float(np.abs(d).max())
    st.fro2 = float((F * F).sum())
    sc = np.where(mx > 0, np.ldexp(1.0, 2 * (e + 2)), 0.0)
    st.es2 = float(sc.sum())
    for p in range(KMAX):
        st.phi2[p] = float((sc * (digs[p] ** 2).sum(axis=1)).sum())
    return st, F, e, digs


@pytest.mark.parametrize("seed,scale", [(1, 1e-3), (2, 0.3), (3, 1e-6)])
def test_product_bound_holds_against_the_emulated_digit_arithmetic(seed, scale):
    L = _lib.load()
    rng = np.random.default_rng(seed)
    n = 192
    A = np.diag(rng.uniform(0.5, 1.5, n)) + scale * rng.standard_normal((n, n)) * rng.uniform(0.1, 1.0, (n, 1))
    B = np.diag(rng.uniform(0.1, 0.3, n)) + scale * rng.standard_normal((n, n)) + scale * 0.5   # a flat part on top of the noise
    sa, Fa, ea, da = _stats(A)
    sb, Fb, eb, db = _stats(B)
    G = Fa @ Fb.T
    sA, sB = np.ldexp(1.0, ea + 2), np.ldexp(1.0, eb + 2)
    last = None
    for k, T in ((3, 4), (3, 5), (4, 5), (4, 6), (5, 6), (5, 7), (6, 7), (6, 12)):
        acc = np.zeros((n, n))
        for p in range(1, k + 1):
            for q in range(1, k + 1):
                if p + q <= T:
                    acc += (da[p - 1] @ db[q - 1].T) * 256.0 ** -(p + q)
        err = np.sqrt((((acc * sA[:, None] * sB[None, :]) - G) ** 2).sum())
        bound = L.eagle_w8_host_bound(C.byref(sa), C.byref(sb), k, T, n)
        assert err <= bound + 1e-15 * np.sqrt((G * G).sum()), (k, T, err, bound)   # (+ the fp64 rounding of this emulation itself)
        if last is not None:
            assert bound <= last * (1 + 1e-12)                                     # more pairs never loosen it
        last = bound
