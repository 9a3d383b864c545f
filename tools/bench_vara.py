#!/usr/bin/env python3
"""GPU A/B of k_vara_i8 variants (tune switch) on one resident shard, interleaved rounds in one process."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from eagleeverything_amd import _lib
from eagleeverything_amd.sharded import DeviceShard

n, L = int(os.environ.get("N", 5000)), int(os.environ.get("LM", 131072))
S = int(os.environ.get("SLICES", 5))
variants = [int(v) for v in os.environ.get("VARIANTS", "0,9,8").split(",")]  # 0 = shipped (384 x 256 tile, asm-pipelined k-step), 9 = the same tile scheduled by hipcc, 8 = the 256 x 256 tile form, 7 = no cut of the last round, 14 = the XCD's workers paced by a soft barrier per column-tile pair (k_vara_i8p<true>, round-3 experiment)
lib = _lib.load()
sh = DeviceShard(n, L)
sh.fill_synthetic()
sh.mode, sh.nslices = 1, S
gen = torch.Generator(device=sh.dev); gen.manual_seed(1)
A = torch.randn((n, 64), generator=gen, device=sh.dev, dtype=torch.float64) / 64.0
Sm = torch.eye(n, dtype=torch.float64, device=sh.dev) * 0.4 + A @ A.T
V = 0.5 * torch.eye(n, dtype=torch.float64, device=sh.dev) - 0.01 * (A[:, :8] @ A[:, :8].T)
sh.set_operands(Sm, V, torch.randn(n, generator=gen, device=sh.dev, dtype=torch.float64))
sh.scan_operands(); sh.vara_prepare()
ref = None
res = {v: [] for v in variants}
for rnd in range(6):
    for v in variants:
        lib.eagle_dev_set_tune(sh.ctx, v)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        sh.vara_prepare(with_a=False)  # re-zeroes q
        e0.record(); sh.vara_kernel(); e1.record()
        torch.cuda.synchronize()
        if ref is None: ref = sh.vara.clone()
        else: assert torch.equal(ref, sh.vara), "variant %d differs" % v
        if rnd: res[v].append(e0.elapsed_time(e1))
lib.eagle_dev_set_tune(sh.ctx, 0)
print("slices used", sh.vara_i8_info())
for v in variants:
    ms = np.array(res[v]); print("variant %d: median %.3f ms  min %.3f ms" % (v, np.median(ms), ms.min()))
