#!/usr/bin/env python3
"""GPU A/B of k_gemm_f64_list schedule variants (ctx tune switch) on W = S (V S): interleaved rounds in one process."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from eagleeverything_amd import _lib
from eagleeverything_amd.sharded import DeviceShard

n = int(os.environ.get("N", 10000))
# 0 = shipped (k_gemm_f64_dma, 256 x 128 tiles, LDS-DMA, round 3); 25 = eight waves per 128 x 128 tile (shipped in round 2); 26 = four waves
variants = [int(v) for v in os.environ.get("VARIANTS", "0,25,26").split(",")]
lib = _lib.load()
sh = DeviceShard(n, 256)
gen = torch.Generator(device=sh.dev); gen.manual_seed(1)
A = torch.randn((n, 64), generator=gen, device=sh.dev, dtype=torch.float64) / 64.0
Sm = torch.eye(n, dtype=torch.float64, device=sh.dev) * 0.4 + A @ A.T
V = 0.5 * torch.eye(n, dtype=torch.float64, device=sh.dev) - 0.01 * (A[:, :8] @ A[:, :8].T)
sh.set_operands(Sm, V, torch.randn(n, generator=gen, device=sh.dev, dtype=torch.float64))
ref = None
refs = {}
res = {v: [] for v in variants}
for rnd in range(5):
    for v in variants:
        lib.eagle_dev_set_tune(sh.ctx, v)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); sh.scan_operands(); e1.record()
        torch.cuda.synchronize()
        if ref is None: ref = sh.Wu.clone()
        if v not in refs: refs[v] = sh.Wu.clone()
        assert torch.equal(refs[v], sh.Wu), "variant %d is not reproducible" % v      # every variant: the same bits on every run
        # different kernels / tile orders sum in different (fixed) orders: agreement to rounding
        assert torch.allclose(ref, sh.Wu, rtol=1e-12, atol=1e-13 * float(ref.abs().max())), "variant %d differs" % v
        if rnd: res[v].append(e0.elapsed_time(e1))
lib.eagle_dev_set_tune(sh.ctx, 0)
flops = 3.0 * sh.np_ ** 3
for v in variants:
    ms = np.array(res[v]); print("variant %d: median %.3f ms  min %.3f ms  -> %.1f TFLOP/s (3 np^3)" % (v, np.median(ms), ms.min(), flops / np.median(ms) / 1e9))
