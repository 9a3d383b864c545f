"""Host emulation of device-side index arithmetic (no GPU): the scalar restatement of a kernel's work decomposition must cover
every unit of work exactly once.  The kernels themselves are checked against the oracle on the GPU box; these tests catch a
wrong map before a box is spent on it."""
import random


PMAX = 16  # VARA_TAIL_PMAX


def vara_tail_pieces(tail, npair):
    """csrc/eagle_i8mfma.hip vara_tail_pieces: pieces per worker of the last, partly filled round of an XCD."""
    if tail <= 0:
        return 1
    best, bn, bd = 1, 1, 1
    for p in range(2, min(npair, PMAX) + 1):
        rounds = (tail * p + 31) >> 5
        if rounds * bd < bn * p:
            best, bn, bd = p, rounds, p
    return best


def vara_i8w_cover(ntm, S, npair, smax=7, cut=True):
    """k_vara_i8p / k_vara_i8w (csrc/eagle_i8mfma.hip): block b -> (xcd = b & 7, slot = b >> 3) -> worker -> (marker tile, slice) and, for the
    workers of the last, partly filled round, a piece [pair0, pair1) of the column-tile pairs."""
    groups = (ntm + 7) >> 3
    wx = groups * S
    full = (wx >> 5) << 5
    tail = wx - full
    psplit0 = vara_tail_pieces(tail, npair) if cut else 1
    seen = {}
    for xcd in range(8):
        for slot in range(groups * smax + 31 * PMAX):     # the launch: 8 * (groups * smax + 31 * VARA_TAIL_PMAX) blocks
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


def test_tail_pieces_rule():
    for npair in (1, 2, 5, 20, 100):
        for tail in range(0, 32):
            p = vara_tail_pieces(tail, npair)
            assert 1 <= p <= max(1, min(npair, PMAX))
            cost = lambda q: ((tail * q + 31) >> 5) / q
            if tail:
                assert all(cost(p) <= cost(q) + 1e-12 for q in range(1, min(npair, PMAX) + 1))   # the minimum ...
                assert all(cost(q) > cost(p) - 1e-12 or q >= p for q in range(1, min(npair, PMAX) + 1))  # ... with the fewest pieces
    assert vara_tail_pieces(3, 20) == 10


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


def test_slice_permutation_matches_the_transposed_tile_layout():
    """k_slice_w (perm128) against k_vara_i8p's epilogue (csrc/eagle_i8mfma.hip): inside every block of 128, column c of Wu is
    stored at row n*32 + 8*(x>>2) + (x&3) + 4*h with h = c>>6, n = (c>>4)&3, x = c&15 -- the row of the wave's W-digit tile that the
    32 x 32 x 32 MFMA (W digits as SrcA) turns into register x of column tile n in lane half h.  The epilogue reads, for lane
    half h and column tile n, the 16 genotype bytes at columns 64 h + 16 n + x of the wave's 128: byte x must belong to
    register x.  A bijection on 0..127, block by block."""
    def stored_row(c):                       # k_slice_w
        c7, cx = c & 127, c & 15
        return (c & ~127) | (((c7 >> 4) & 3) << 5) | ((cx >> 2) << 3) | (cx & 3) | ((c7 >> 6) << 2)
    assert sorted(stored_row(c) for c in range(256)) == list(range(256))
    for c in range(256):
        assert stored_row(c) >> 7 == c >> 7                                    # stays inside its block of 128
    for h in range(2):
        for n in range(4):
            for x in range(16):
                mfma_row = n * 32 + 8 * (x >> 2) + (x & 3) + 4 * h             # D-tile row of register x in lane half h (32 x 32 layout)
                col_read = 64 * h + 16 * n + x                                 # epilogue: g[n] byte x at column offset h*64 + n*16
                assert stored_row(col_read) == mfma_row
