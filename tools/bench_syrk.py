#!/usr/bin/env python3
"""GPU A/B of the MM^T kernel forms (tune switch) on one resident shard, interleaved rounds in one process.
VARIANTS: 0 = shipped (k_syrk_f4w: 384 x 256 tiles, asm-pipelined k-step, from 3,072 padded individuals; 11 forces it), 10 = 256 x 256 tiles with the pipelined k-step
(k_syrk_f4p), 9 = 256 x 256 tiles scheduled by hipcc (k_syrk_f4)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from eagleeverything_amd import _lib
from eagleeverything_amd.sharded import DeviceShard

n, L = int(os.environ.get("N", 5000)), int(os.environ.get("LM", 500000))
variants = [int(v) for v in os.environ.get("VARIANTS", "0,10,9").split(",")]
lib = _lib.load()
sh = DeviceShard(n, L)
sh.fill_synthetic()
sh.individual_major_fp4()
sh.M8 = None
c32 = torch.empty((sh.np_, sh.np_), dtype=torch.int32, device=sh.dev)
ref = None
res = {v: [] for v in variants}
for rnd in range(6):
    for v in variants:
        lib.eagle_dev_set_tune(sh.ctx, v)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c32.zero_()
        torch.cuda.synchronize()
        e0.record()
        assert lib.eagle_dev_mmt_accumulate_f4(sh.ctx, sh.M4.data_ptr(), sh.np_, sh.Lp, sh.Lp // 2, c32.data_ptr(), sh._stream()) == 0
        e1.record()
        torch.cuda.synchronize()
        if ref is None: ref = c32.clone()
        else: assert torch.equal(ref, c32), "variant %d differs" % v
        if rnd: res[v].append(e0.elapsed_time(e1))
lib.eagle_dev_set_tune(sh.ctx, 0)
# exact check of a few entries against int64 sums of the int8 image
Mt = sh.Mt8[:sh.Lloc, :8].to(torch.float64)  # exact: sums far below 2^53
assert torch.equal((Mt.T @ Mt).to(torch.int32), ref[:8, :8]), "MM^T corner differs from the int64 sum"
ops = sh.np_ * (sh.np_ + 256.0) * sh.Lp  # executed MAC-flop (upper 256-tiles, 2 per MAC)
for v in variants:
    ms = np.array(res[v]); print("variant %d: median %.3f ms  min %.3f ms  %.2f POP/s" % (v, np.median(ms), ms.min(), ops / np.median(ms) / 1e12))
