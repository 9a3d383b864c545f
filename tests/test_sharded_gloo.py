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
        # reduce to ONE rank (what bench.py does: only the process that hands MM^T to the host algebra needs the sum): rank 0
        # holds the sum (what the other ranks' buffers hold afterwards is unspecified, as with any reduce)
        for shape in ("plain", "tiled"):
            if shape == "plain":
                t = torch.from_numpy(part.copy())
                mine = part.astype(np.int64)
                want = g["MMt"]
            else:
                t = torch.zeros((512, 512), dtype=torch.int32)
                t[:n, :n] = torch.from_numpy(part.copy())
                mine = t.numpy().astype(np.int64).copy()
                want = exp
            coll.sum_partial_mmt(t, dst=0)
            got = t.numpy().astype(np.int64)
            if rank == 0:
                assert np.array_equal(got[:256, :], want[:256, :]) if shape == "tiled" else np.array_equal(got, want)
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


def _gpu_worker(rank, world, port, out_dir):
    """Two ranks sharing cuda:0 (collectives through the host over gloo): the device-resident sharded path end to end."""
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from eagleeverything_amd.sharded import DeviceShard
        n, Lloc = 640, 3000                      # 5 x 128-row tiles of W: world 2 does not divide them -> replicated W
        n2 = 512                                 # 4 tiles: shared W path
        for nn in (n, n2):
            rng = np.random.default_rng(7)
            Mt8 = (rng.binomial(2, rng.uniform(0.02, 0.5, size=(world * Lloc, 1)), size=(world * Lloc, nn)) - 1).astype(np.int8)
            A = rng.standard_normal((nn, 24)) / 6.0
            S = np.eye(nn) + A @ A.T
            V = 0.6 * np.eye(nn) - 0.04 * (A[:, :3] @ A[:, :3].T)
            ahat = rng.standard_normal(nn)
            coll = Collectives(dist)
            sh = DeviceShard(nn, Lloc, first_marker=rank * Lloc, device=0)
            sh.Mt8.zero_()
            sh.Mt8[:Lloc, :nn] = torch.from_numpy(Mt8[rank * Lloc:(rank + 1) * Lloc]).cuda()
            c32 = sh.mmt_partial()
            coll.sum_partial_mmt(c32)
            MMt, mx = sh.mmt_finish(c32)
            assert np.array_equal(MMt.cpu().numpy(), (Mt8.astype(np.int64).T @ Mt8.astype(np.int64)).astype(np.float64))
            Sd = torch.from_numpy(S).cuda() if rank == 0 else torch.empty((nn, nn), dtype=torch.float64, device="cuda")
            Vd = torch.from_numpy(V).cuda() if rank == 0 else torch.empty((nn, nn), dtype=torch.float64, device="cuda")
            ad = torch.from_numpy(ahat).cuda() if rank == 0 else torch.empty(nn, dtype=torch.float64, device="cuda")
            for t in (Sd, Vd, ad):
                coll.broadcast_(t)
            sh.set_operands(Sd, Vd, ad)
            sh.mode = 1
            sh.scan(coll)
            tsqmax, gidx, _ = sh.best()
            sel, best = coll.best_marker(tsqmax, gidx, device=sh.dev)
            # reference: fp64 numpy on the whole marker set
            W = S @ V @ S
            a = Mt8.astype(np.float64) @ (S @ ahat)
            vara = np.einsum("ij,jk,ik->i", Mt8.astype(np.float64), W, Mt8.astype(np.float64))
            tsq = a * a / vara
            assert sel == int(np.nanargmax(tsq)) + 1, (nn, sel, int(np.nanargmax(tsq)) + 1)
            np.testing.assert_allclose(best, np.nanmax(tsq), rtol=9e-7)
            mine = slice(rank * Lloc, (rank + 1) * Lloc)
            np.testing.assert_allclose(sh.vara[:Lloc].cpu().numpy(), vara[mine], rtol=9e-7, atol=1e-10 * np.abs(vara).max())
            np.testing.assert_allclose(sh.a[:Lloc].cpu().numpy(), a[mine], rtol=1e-9, atol=1e-12 * np.abs(a).max())
        open(os.path.join(out_dir, "gpu_ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_sharded_two_ranks_on_one_gpu(tmp_path):
    """The rehearsal of bench.py --gpus 2 as a test: exact MM^T through the packed all-reduce, W shared by rows (n = 512) and
    replicated (n = 640), per-shard scans and the global arg-max against fp64 numpy."""
    world = 2
    mp.spawn(_gpu_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / ("gpu_ok%d" % r)).exists() for r in range(world))
