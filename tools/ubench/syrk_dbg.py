"""Debug aid: the 384 x 256 MM^T kernel (tune 0) against the 256 x 256 one (tune 10) on small shapes; which stages / blocks differ."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from eagleeverything_amd import _lib
from eagleeverything_amd.sharded import DeviceShard
lib = _lib.load()
n = int(os.environ.get("N", 1000))
for L in (256, 512, 768, 1024, 1280, 2560, 5120):
    sh = DeviceShard(n, L); sh.fill_synthetic(); sh.individual_major_fp4()
    outs = {}
    for v in (10, 0):
        lib.eagle_dev_set_tune(sh.ctx, v)
        c = torch.zeros((sh.np_, sh.np_), dtype=torch.int32, device=sh.dev)
        assert lib.eagle_dev_mmt_accumulate_f4(sh.ctx, sh.M4.data_ptr(), sh.np_, sh.Lp, sh.Lp // 2, c.data_ptr(), sh._stream()) == 0
        torch.cuda.synchronize(); outs[v] = c.cpu().numpy()
    G = sh.Mt8[:L, :n].cpu().numpy().astype(np.int64)           # markers x individuals
    per_stage = [(G[t * 256:(t + 1) * 256, 0] ** 2).sum() for t in range((L + 255) // 256)]
    d = outs[0] != outs[10]
    print("L", L, "stages", len(per_stage), "mismatches", int(d.sum()), "diag00 wide", outs[0][0, 0], "ref", outs[10][0, 0], "per-stage", per_stage[:12])
    del sh
