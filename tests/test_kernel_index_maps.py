"""Host emulation of device-side index arithmetic (no GPU): the scalar restatement of a kernel's work decomposition must cover
every unit of work exactly once.  The kernels themselves are checked against the oracle on the GPU box; these tests catch a
wrong map before a box is spent on it."""
import random


def vara_i8w_cover(ntm, S, npair, smax=7, cut=True):
    """k_vara_i8p / k_vara_i8w (csrc/eagle_i8mfma.hip): block b -> (xcd = b & 7, slot = b >> 3) -> worker -> (marker tile, slice) and, for the
    workers of a last round that is less than half full, a piece [pair0, pair1) of the column-tile pairs."""
    groups = (ntm + 7) >> 3
    wx = groups * S
    full = (wx >> 5) << 5
    tail = wx - full
    psplit0 = 1
    if cut and 0 < tail <= 16:
        psplit0 = min(32 // tail, npair)
    seen = {}
    for xcd in range(8):
        for slot in range(groups * smax + 32):          # the launch: 8 * (groups * smax + 32) blocks
            worker, piece, psplit = slot, 0, psplit0
            if slot >= full:
                u = slot - full
                worker, piece = full + u // psplit, u % psplit
            else:
                psplit = 1
            if worker >= wx:
                continue
            mt, sl = (worker // S) * 8 + xcd, worker % S
            if mt >= ntm:
                continue
            for p in range(npair * piece // psplit, npair * (piece + 1) // psplit):
                seen[(mt, sl, p)] = seen.get((mt, sl, p), 0) + 1
    return len(seen) == ntm * S * npair and all(v == 1 for v in seen.values())


def test_vara_i8w_every_marker_tile_slice_and_column_pair_exactly_once():
    rng = random.Random(1)
    cases = [(53, 5, 2), (53, 4, 2), (1303, 4, 10), (326, 4, 20), (7, 3, 1), (1, 4, 3), (2605, 4, 20), (8, 7, 1), (9, 1, 40)]
    cases += [(rng.randint(1, 700), rng.randint(1, 7), rng.randint(1, 25)) for _ in range(400)]
    for ntm, S, npair in cases:
        assert vara_i8w_cover(ntm, S, npair), (ntm, S, npair)
        assert vara_i8w_cover(ntm, S, npair, cut=False), (ntm, S, npair)


def test_split_markers_ranges():
    """eagle_api.cpp split_markers: contiguous ranges, boundaries at multiples of 256, every marker once."""
    for L in (1, 255, 256, 257, 6001, 1000000, 5000000):
        for nd in (1, 2, 3, 8):
            tiles = (L + 255) // 256
            edge = [0] + [min(L, (tiles * k // nd) * 256) for k in range(1, nd)] + [L]
            assert all(a <= b for a, b in zip(edge, edge[1:])) and edge[0] == 0 and edge[-1] == L
            assert all(e % 256 == 0 for e in edge[1:-1])
