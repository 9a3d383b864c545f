#!/usr/bin/env python3
"""GPU A/B of the fp64 vara kernel (scan mode 0) on one resident shard: 0 = k_vara_f64d (round 3: fp64 A tile converted once per K block,
LDS-DMA for Wu, no VALU between the MFMAs), 28 = k_vara_f64 (round 2).  Different (fixed) summation orders inside a K block: agreement to
rounding; each variant reproducible bit for bit."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from eagleeverything_amd import _lib
from eagleeverything_amd.sharded import DeviceShard

n, L = int(os.environ.get("N", 10000)), int(os.environ.get("LM", 131072))
variants = [int(v) for v in os.environ.get("VARIANTS", "0,28").split(",")]
lib = _lib.load()
sh = DeviceShard(n, L)
sh.fill_synthetic()
sh.mode = 0
gen = torch.Generator(device=sh.dev); gen.manual_seed(1)
A = torch.randn((n, 64), generator=gen, device=sh.dev, dtype=torch.float64) / 64.0
Sm = torch.eye(n, dtype=torch.float64, device=sh.dev) * 0.4 + A @ A.T
V = 0.5 * torch.eye(n, dtype=torch.float64, device=sh.dev) - 0.01 * (A[:, :8] @ A[:, :8].T)
sh.set_operands(Sm, V, torch.randn(n, generator=gen, device=sh.dev, dtype=torch.float64))
sh.scan_operands()
refs, res = {}, {v: [] for v in variants}
for rnd in range(4):
    for v in variants:
        lib.eagle_dev_set_tune(sh.ctx, v)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); sh.vara_kernel(); e1.record()
        torch.cuda.synchronize()
        if v not in refs: refs[v] = sh.vara.clone()
        assert torch.equal(refs[v], sh.vara), "variant %d is not reproducible" % v
        if rnd: res[v].append(e0.elapsed_time(e1))
lib.eagle_dev_set_tune(sh.ctx, 0)
base = refs[variants[0]]
for v in variants[1:]:
    rel = float(((refs[v] - base).abs() / base.abs().clamp_min(1e-300))[:L].max())
    print("variant %d vs %d: max relative difference %.3g" % (v, variants[0], rel))
    assert rel < 1e-11
np_, Lp = sh.np_, sh.Lp
flops = sum(2.0 * Lp * 128 * min((ct + 1) * 128, np_) for ct in range(np_ // 128))
for v in variants:
    ms = np.array(res[v]); print("variant %d: median %.3f ms  min %.3f ms  -> %.1f TFLOP/s = %.3f of 78.6" % (v, np.median(ms), ms.min(), flops / np.median(ms) / 1e9, flops / np.median(ms) / 1e9 / 78.6))
