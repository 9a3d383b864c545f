"""CPU, world_size 2, gloo: the N>1 logic of the marker-sharded path (eagleeverything_amd.sharded.Collectives).

Per-rank compute is done by the oracle here (there is no GPU); what is under test is the shard arithmetic, the
exact integer all-reduce of partial MM^T, and the all-gather + first-index tie-break of the per-shard top score
(find_qtl.R:76-80 semantics across shards)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN, ROOT
from eagleeverything_amd.sharded import Collectives, pick_best, shard_range


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, case, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle_np
        g = dict(np.load(os.path.join(GOLDEN, case + ".npz")))
        M8 = g["M8"]
        n, L = M8.shape
        m0, m1 = shard_range(L, rank, world)
        coll = Collectives(dist)
        # MM^T: exact int32 partial of the shard's marker columns -> one all-reduce
        part = oracle_np.mmt_int64(M8[:, m0:m1]).astype(np.int32)
        c32 = torch.from_numpy(part.copy())
        coll.sum_partial_mmt(c32)
        assert np.array_equal(c32.numpy().astype(np.int64), g["MMt"])
        # scan: operands broadcast from rank 0, each rank scans its rows of Mt
        S = torch.from_numpy(g["S"].copy()) if rank == 0 else torch.empty_like(torch.from_numpy(g["S"].copy()))
        coll.broadcast_(S)
        assert np.array_equal(S.numpy(), g["S"])
        a, vara = oracle_np.a_and_vara(M8.T[m0:m1], g["S"], g["V"], g["ahat"])
        tsq, idx1, mx = oracle_np.tsq_argmax(a, vara)
        gidx0 = (idx1 - 1 + m0) if idx1 > 0 else -1
        sel, best = coll.best_marker(mx, gidx0)
        assert sel == int(g["argmax"]), (sel, int(g["argmax"]))
        np.testing.assert_allclose(best, float(g["tsqmax"]), rtol=1e-9)
        # the same sum when only the upper 256-tiles of a padded partial are live (what the kernels produce): the lower
        # tiles do not travel and stay untouched
        big = torch.zeros((512, 512), dtype=torch.int32)
        big[:n, :n] = torch.from_numpy(part.copy())
        big[256:, :256] = -7 - rank                                    # dead tile: must neither travel nor change
        coll.sum_partial_mmt(big)
        exp = np.zeros((512, 512), dtype=np.int64)
        exp[:n, :n] = g["MMt"]
        got = big.numpy().astype(np.int64)
        assert np.array_equal(got[:256, :], exp[:256, :]) and np.array_equal(got[256:, 256:], exp[256:, 256:])
        assert np.all(got[256:, :256] == -7 - rank)
        # shared W: every rank computes its row block of S V S, one all-gather completes the image
        W = g["S"] @ (g["V"] @ g["S"])
        rows = 64
        full = torch.zeros((world * rows, 32), dtype=torch.float64)
        mine = torch.from_numpy(W[rank * rows:(rank + 1) * rows, :32].copy())
        coll.all_gather_rows(full, mine)
        assert np.array_equal(full.numpy(), W[:world * rows, :32])
        # tie across shards: both shards report the same maximum -> the smaller global index wins
        sel2, _ = coll.best_marker(5.0, 100 + 50 * (world - 1 - rank))
        assert sel2 == 101
        # a shard whose tsq are all NaN (index -1) never wins
        sel3, v3 = coll.best_marker(np.nan if rank == 0 else 2.0, -1 if rank == 0 else 7)
        assert sel3 == 8 and v3 == 2.0
        open(os.path.join(out_dir, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", ["genoDemo_150x4998", "synth_203x1531"])
def test_sharded_two_ranks_gloo(case, tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), case, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / ("ok%d" % r)).exists() for r in range(world))


def test_shard_range_partitions():
    for L in (1, 7, 500000, 1000003):
        for world in (1, 2, 3, 8):
            rs = [shard_range(L, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == L
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in rs]
            assert max(sizes) - min(sizes) <= 1


def test_pick_best_rules():
    assert pick_best([1.0, 3.0, 3.0], [5, 9, 2]) == (3, 3.0)
    assert pick_best([np.nan, np.nan], [-1, -1])[0] == 0
    assert pick_best([np.inf, 2.0], [4, 1]) == (5, np.inf)
