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


def test_three_slice_workers_of_a_marker_tile_rarely_straddle_a_round():
    """VERDICT r3 item 5 asked for a work map in which the S = 3 workers of a marker tile share one round of an XCD's 32 workgroups.
    The existing map (worker = 3 * tile + slice, rounds of 32 consecutive workers) already does that for 30 of every 32 tiles: only
    the tiles whose three workers sit across a multiple of 32 straddle, 2 in 32 -- and none at all with S = 4.  What S = 3 costs per
    slice (9.35 against 8.88 ms) is the 3-way instead of 4-way sharing of a genotype panel, not the map."""
    ntm = 2605                                   # marker tiles of 384 at 1,000,000 markers
    groups = (ntm + 7) >> 3                      # marker tiles per XCD
    for S, expect in ((3, 2.0 / 32), (4, 0.0)):
        straddle = sum(1 for t in range(groups) if (S * t) // 32 != (S * t + S - 1) // 32)
        assert abs(straddle / groups - expect) <= 1.0 / groups + 1e-12, (S, straddle, groups)
    # L2 fill per stage step of an XCD's 32 workers: a genotype stage (48 KiB) per marker tile in flight + a W-digit stage (32 KiB) per slice
    fill = lambda S: (32.0 / S) * 48 + S * 32
    assert fill(4) == 512 and abs(fill(3) - 608) < 1


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


# ---- k_gemm_f64_dma (csrc/eagle_kernels.hip): LDS swizzles, DMA source offsets, k ownership -----------------------------------
# ds_read_b128 serves a wave in four groups of 16 lanes, ds_read_b64 in two groups of 32 (MI355X_MICROARCH.md, LDS); inside a group
# two lanes conflict when they touch the same 4-byte bank (64 banks) at different addresses.
B128_GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
               list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)), list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]
B64_GROUPS = [list(range(0, 32)), list(range(32, 64))]


def _conflict_free(addr_of_lane, groups, nbytes):
    for grp in groups:
        owner = {}
        for lane in grp:
            a = addr_of_lane(lane)
            for b in range(a // 4, (a + nbytes) // 4):
                bank = b % 64
                if owner.setdefault(bank, b) != b:
                    return False
    return True


def test_gemm_f64_dma_lds_maps():
    A_BYTES = 256 * 128
    # (1) the DMA writes wave-linear LDS bytes; the swizzled SOURCE offsets put logical chunk c of row r at physical chunk c ^ ((r >> 1) & 7)
    lda8 = 10240 * 8
    for w in range(8):
        for i in range(4):
            grp = w * 4 + i
            for lane in range(64):
                voffE = (lane >> 3) * lda8 + (((lane & 7) ^ (lane >> 4)) << 4)
                voff = voffE ^ 64 if (i & 1) else voffE
                src = grp * 8 * lda8 + voff                      # byte offset from the tile's first row, K block 0
                lds = grp * 1024 + lane * 16
                r, p = lds // 128, (lds % 128) // 16
                assert src == r * lda8 + ((p ^ ((r >> 1) & 7)) << 4)
    # B: row k = 2w + i of the stage, 128-byte block nt at physical block nt ^ ((k >> 1) & 1)
    for w in range(8):
        for i in range(2):
            k = 2 * w + i
            for lane in range(64):
                voffB = (((lane >> 3) ^ (w & 1)) << 7) + ((lane & 7) << 4)
                q, within = lane >> 3, lane & 7
                assert voffB == ((q ^ ((k >> 1) & 1)) << 7) + within * 16
    # (2) fragment reads: conflict free, and every (row, k) / (k, column) of the K block is read by exactly the lane group that owns it
    for wr in range(4):
        for m in range(4):
            for h in range(2):
                def addrA(lane):
                    i16, g = lane & 15, lane >> 4
                    offA = (wr * 64 + i16) * 128 + ((g ^ (i16 >> 1)) << 4)
                    return (offA ^ (h << 6)) + m * 2048
                assert _conflict_free(addrA, B128_GROUPS, 16)
                for lane in range(64):
                    i16, g = lane & 15, lane >> 4
                    a = addrA(lane)
                    r, p = a // 128, (a % 128) // 16
                    assert r == wr * 64 + m * 16 + i16
                    assert p ^ ((r >> 1) & 7) == g + 4 * h          # logical chunk g (k = 2g, 2g+1) or 4+g (k = 8+2g, 9+2g)
    for wc in range(2):
        for n in range(4):
            for h in range(2):
                for e in range(2):
                    def addrB(lane):
                        i16, g = lane & 15, lane >> 4
                        return A_BYTES + (2 * g) * 1024 + i16 * 8 + h * 8192 + e * 1024 + (((wc * 4 + n) ^ (g & 1)) << 7)
                    assert _conflict_free(addrB, B64_GROUPS, 8)
                    for lane in range(64):
                        i16, g = lane & 15, lane >> 4
                        a = addrB(lane) - A_BYTES
                        k, blk, col = a // 1024, (a % 1024) // 128, (a % 128) // 8
                        assert k == 2 * g + e + 8 * h                # the k the A fragment element (h, e) of lane group g holds
                        assert blk ^ ((k >> 1) & 1) == wc * 4 + n and col == i16
    # (3) over a K block the four MFMA steps (h, e) of the four lane groups cover k = 0..15 exactly once
    assert sorted(2 * g + e + 8 * h for g in range(4) for h in range(2) for e in range(2)) == list(range(16))


def test_gemm_f64_dma_tile_lists_cover_every_128_tile_once():
    """Tile lists of k_gemm_f64_dma (256-row tiles at 128-row granularity): every 128 x 128 tile of the wanted set is computed by
    exactly one 256 x 128 tile, for whole matrices, row ranges with an odd number of 128-row tiles, and the two triangular kinds."""
    def tiles(nt, kind, rt0, rt1):
        out = []
        for i in range(rt0, rt1, 2):
            for j in range(nt):
                if kind == 0 or (kind == 1 and j >= i) or (kind == 2 and j < i) or (kind == 3 and j <= i + 1):
                    out.append((i, j))
        return out
    for nt in (2, 4, 6, 80):
        for rt0, rt1 in ((0, nt), (1, nt), (0, 1), (nt - 3 if nt >= 4 else 0, nt)):
            cover = {}
            for i, j in tiles(nt, 0, rt0, rt1):
                for ii in (i, i + 1):
                    if ii < rt1:                                    # rows at or beyond row_end are neither read nor stored
                        cover[(ii, j)] = cover.get((ii, j), 0) + 1
            assert set(cover) == {(i, j) for i in range(rt0, rt1) for j in range(nt)} and set(cover.values()) == {1}
        up = {(ii, j) for i, j in tiles(nt, 1, 0, nt) for ii in (i, i + 1)}
        lo = {(ii, j) for i, j in tiles(nt, 2, 0, nt) for ii in (i, i + 1)}
        assert not (up & lo) and up | lo == {(i, j) for i in range(nt) for j in range(nt)}
        assert {(i, j) for i in range(nt) for j in range(nt) if i <= j} <= up       # every 128-tile on or above the diagonal
        below = {(ii, j) for i, j in tiles(nt, 3, 0, nt) for ii in (i, i + 1)}
        assert {(i, j) for i in range(nt) for j in range(nt) if i >= j} <= below    # ... on or below it (stored transposed)


def test_vara_f64d_lds_maps():
    """k_vara_f64d: the staging threads' fp64 A tile, the Wu DMA and the fragment reads agree on one layout, the reads are free of
    bank conflicts and a K block's sixteen k are owned once each."""
    A_BYTES = 128 * 128
    # (1) A: thread t converts bytes k = 8 ah + 2j, 2j + 1 of row ar = t >> 1 and writes logical chunk 4 ah + j at chunk ^ ((ar >> 1) & 7)
    written = {}
    for t in range(256):
        ar, ah = t >> 1, t & 1
        for j in range(4):
            a = ar * 128 + (((4 * ah + j) ^ ((ar >> 1) & 7)) << 4)
            r, p = a // 128, (a % 128) // 16
            assert r == ar and p ^ ((r >> 1) & 7) == 4 * ah + j
            assert a not in written
            written[a] = (ar, 8 * ah + 2 * j)
    assert len(written) == 128 * 8
    # (2) Wu by DMA: wave w, instruction i writes row k = 4w + i of the stage (wave-linear 1 KiB); source block q of lane at q ^ ((k >> 1) & 1)
    for w in range(4):
        for i in range(4):
            k = 4 * w + i
            for lane in range(64):
                voffB0 = ((lane >> 3) << 7) + ((lane & 7) << 4)
                voff = voffB0 ^ 128 if (i & 2) else voffB0
                phys_blk, within = lane >> 3, lane & 7
                assert voff == ((phys_blk ^ ((k >> 1) & 1)) << 7) + within * 16
    # (3) fragment reads
    for wr in range(2):
        for m in range(4):
            for h in range(2):
                def addrA(lane):
                    i16, g = lane & 15, lane >> 4
                    offA = (wr * 64 + i16) * 128 + ((g ^ (i16 >> 1)) << 4)
                    return (offA ^ (h << 6)) + m * 2048
                assert _conflict_free(addrA, B128_GROUPS, 16)
                for lane in range(64):
                    i16, g = lane & 15, lane >> 4
                    r, k0 = written[addrA(lane)]
                    assert r == wr * 64 + m * 16 + i16 and k0 == 2 * g + 8 * h
    for wc in range(2):
        for n in range(4):
            for h in range(2):
                for e in range(2):
                    def addrB(lane):
                        i16, g = lane & 15, lane >> 4
                        return A_BYTES + (2 * g) * 1024 + i16 * 8 + h * 8192 + e * 1024 + (((wc * 4 + n) ^ (g & 1)) << 7)
                    assert _conflict_free(addrB, B64_GROUPS, 8)
                    for lane in range(64):
                        i16, g = lane & 15, lane >> 4
                        a = addrB(lane) - A_BYTES
                        k, blk, col = a // 1024, (a % 1024) // 128, (a % 128) // 8
                        assert k == 2 * g + e + 8 * h
                        assert blk ^ ((k >> 1) & 1) == wc * 4 + n and col == i16
    assert A_BYTES + 16 * 1024 == 32768                                    # one stage; two stages + Psum = 66 KiB: two workgroups per CU
