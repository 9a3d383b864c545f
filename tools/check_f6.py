#!/usr/bin/env python3
"""fp4 x fp6 vara engine (mode 2) against the int8 engine (mode 1) and the fp64 kernel (mode 0): values, then time at C2."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from eagleeverything_amd.sharded import DeviceShard


def operands(sh, n, seed=5):
    g = torch.Generator(device=sh.dev)
    g.manual_seed(seed)
    A = torch.randn((n, 48), generator=g, device=sh.dev, dtype=torch.float64) / 32.0
    S = 0.5 * torch.eye(n, dtype=torch.float64, device=sh.dev) + A @ A.T
    V = 0.6 * torch.eye(n, dtype=torch.float64, device=sh.dev) - 0.03 * (A[:, :6] @ A[:, :6].T)
    ahat = torch.randn(n, generator=g, device=sh.dev, dtype=torch.float64)
    return S, V, ahat


for n, L in ((700, 3000), (1003, 20011)):
    sh = DeviceShard(n, L)
    sh.fill_synthetic(seed=3)
    sh.set_operands(*operands(sh, n))
    out = {}
    for mode in (0, 1, 2):
        sh.mode = mode
        sh.scan()
        torch.cuda.synchronize()
        out[mode] = sh.vara[:L].clone()
        if mode:
            print("n=%d L=%d mode %d: slices %d bound %.3e" % ((n, L, mode) + sh.vara_i8_info()[:2]))
    for mode in (1, 2):
        rel = ((out[mode] - out[0]).abs() / out[0].abs()).max().item()
        print("   mode %d vs fp64 kernel: max rel %.3e" % (mode, rel))
    for S_ in (4, 6, 8, 10):
        sh.mode, sh.nslices, sh.ws = 2, S_, None
        sh.scan()
        torch.cuda.synchronize()
        err = (sh.vara[:L] - out[0]).abs().max().item()
        print("   mode 2 forced S=%d: max abs err %.3e, bound %.3e" % (S_, err, sh.vara_i8_info()[1]))
    del sh

n, L = 5000, 500000
sh = DeviceShard(n, L)
sh.fill_synthetic()
sh.set_operands(*operands(sh, n, seed=7))
for mode in (1, 2, 1, 2):
    sh.mode, sh.nslices = mode, 0
    sh.scan()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        sh.scan_operands(); sh.vara_prepare()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); sh.vara_kernel(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print("C2 mode %d: slices %d, vara kernel %.2f ms (min %.2f)" % (mode, sh.vara_i8_info()[0], np.median(ts), min(ts)))
    if mode == 1:
        ref = sh.vara[:L].clone()
    else:
        print("   mode 2 vs mode 1: max rel %.3e" % ((sh.vara[:L] - ref).abs() / ref.abs()).max().item())
