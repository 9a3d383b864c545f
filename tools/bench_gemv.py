#!/usr/bin/env python3
"""GPU micro-benchmark of the genotype pass (a = Mt v, and with the diagonal term d = sum m^2 w) on the int8 MFMA.
HBM roofline: L_pad * n_pad genotype bytes per launch.  (History: the fp64 VALU form it replaced ran 0.464 / 0.623 ms
for a / a+d at C2 in the same process; nontemporal loads cost the MFMA form 12 %.)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from eagleeverything_amd import _lib
from eagleeverything_amd.sharded import DeviceShard

n, L = int(os.environ.get("N", 5000)), int(os.environ.get("LM", 500000))
lib = _lib.load()
lib.eagle_dev_gemv2_i8.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_long, C.c_void_p, C.c_void_p, C.c_double,
                                   C.c_void_p, C.c_void_p, C.c_void_p]
lib.eagle_dev_gemv2_i8.restype = C.c_int
sh = DeviceShard(n, L)
sh.fill_synthetic()
g = torch.Generator(device=sh.dev)
g.manual_seed(1)
v = torch.zeros(sh.np_, dtype=torch.float64, device=sh.dev)
w = torch.zeros(sh.np_, dtype=torch.float64, device=sh.dev)
v[:n] = torch.randn(n, generator=g, device=sh.dev, dtype=torch.float64) * 3.7
w[:n] = torch.rand(n, generator=g, device=sh.dev, dtype=torch.float64) + 0.25
outs = {}
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for with_w in (False, True):
    res = {0: []}
    for rnd in range(6):
        for valu in (0,):
            a = torch.zeros(sh.Lp, dtype=torch.float64, device=sh.dev)
            d = torch.zeros(sh.Lp, dtype=torch.float64, device=sh.dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = lib.eagle_dev_gemv2_i8(sh.ctx, sh.Mt8.data_ptr(), sh.Lp, sh.np_, sh.np_, v.data_ptr(), w.data_ptr() if with_w else None,
                                        0.5, a.data_ptr(), d.data_ptr() if with_w else None, stream)
            e1.record()
            torch.cuda.synchronize()
            assert rc == 0
            if rnd:
                res[valu].append(e0.elapsed_time(e1))
            outs[(with_w, valu)] = (a, d)
    a0, d0 = outs[(with_w, 0)]
    ref_a = 0.5 * (sh.Mt8[:4096].double() @ v)
    print("with_w=%s  a vs torch fp64 on 4096 rows: max rel %.2e" % (with_w, float((a0[:4096] - ref_a).abs().max() / ref_a.abs().max())))
    if with_w:
        ref_d = (sh.Mt8[:4096].double() ** 2) @ w
        print("           d vs torch fp64: max rel %.2e" % float((d0[:4096] - ref_d).abs().max() / ref_d.abs().max()))
    bytes_ = float(sh.Lp) * sh.np_
    for valu in (0,):
        ms = np.median(res[valu])
        print("  %s: median %.3f ms  -> %.0f GB/s (%.1f %% of 8 TB/s)" % ("k_slice_vec + k_gemv_mfma", ms, bytes_ / ms / 1e6, bytes_ / ms / 1e6 / 80))
