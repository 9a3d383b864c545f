"""GPU, RCCL (torch.distributed backend "nccl"), a process group of ONE rank: every collective call of the marker-sharded path
(eagleeverything_amd.sharded.Collectives) issued for real on device tensors -- int32 reduce / all-reduce of the packed upper
tiles of MM^T, all-gather of W's rows, broadcast, all-gather of the shards' top scores -- and the results compared with the
same shard run without a process group.  One GPU per box: this covers the calls (dtypes, shapes, stream ordering against
this library's kernels), not a transfer between devices; the world-2 logic is tests/test_sharded_gloo.py."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_collectives_of_the_sharded_path_on_rccl_single_rank():
    import torch
    import torch.distributed as dist
    from eagleeverything_amd.sharded import Collectives, DeviceShard
    n, L = 1500, 6000   # np = 1536 = 12 row tiles of 128, 6 x 6 tiles of 256
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        coll = Collectives(dist, force=True)
        assert coll.active and coll.world == 1 and not coll.via_host
        sh = DeviceShard(n, L)
        sh.fill_synthetic()
        gen = torch.Generator(device=sh.dev); gen.manual_seed(3)
        A = torch.randn((n, 32), generator=gen, device=sh.dev, dtype=torch.float64) / 32.0
        S = torch.eye(n, dtype=torch.float64, device=sh.dev) * 0.5 + A @ A.T
        V = 0.7 * torch.eye(n, dtype=torch.float64, device=sh.dev) - 0.02 * (A[:, :4] @ A[:, :4].T)
        ahat = torch.randn(n, generator=gen, device=sh.dev, dtype=torch.float64)
        # MM^T: reduce to rank 0 and all-reduce of the packed upper tiles leave the one rank's exact partial unchanged
        c32 = sh.mmt_partial()
        ref = c32.clone()
        coll.sum_partial_mmt(c32, dst=0)
        assert torch.equal(c32, ref)
        coll.sum_partial_mmt(c32)
        assert torch.equal(c32, ref)
        G = sh.Mt8[:L, :n].to(torch.float64)
        assert torch.equal((G.T @ G).to(torch.int32), torch.triu(c32[:n, :n]) + torch.triu(c32[:n, :n], 1).T)
        # scan operands: rows of W through the all-gather against the replicated computation
        sh.mode = 1
        sh.set_operands(S, V, ahat)
        sh.scan_operands()
        W_rep, v_rep = sh.Wu.clone(), sh.v.clone()
        sh.set_operands(S, V, ahat)
        sh.scan_operands(coll)
        # (the row-block form sums W's tiles in another order than the upper-tile form: equal to rounding, not bit for bit)
        assert torch.allclose(sh.Wu, W_rep, rtol=0, atol=1e-13 * float(W_rep.abs().max())) and torch.equal(sh.v, v_rep)
        t = torch.arange(5, dtype=torch.float64, device=sh.dev)
        coll.broadcast_(t)
        assert torch.equal(t.cpu(), torch.arange(5, dtype=torch.float64))
        # the scan and the exchange of the top scores
        sh.scan(coll)
        tsq, idx0, _ = sh.best()
        sel, best = coll.best_marker(tsq, idx0, device=sh.dev)
        assert sel == idx0 + 1 and best == tsq
        a, vara = sh.a[:L].cpu().numpy(), sh.vara[:L].cpu().numpy()
        with np.errstate(divide="ignore", invalid="ignore"):
            t2 = a * a / vara
        assert int(np.nanargmax(t2)) == idx0
    finally:
        dist.destroy_process_group()
