#!/usr/bin/env python3
"""GPU micro-benchmark of the int8 tile-engine schedule variants on the MM^T SYRK (interleaved rounds, one process)."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from eagleeverything_amd import _lib
from eagleeverything_amd.sharded import DeviceShard

n, L = int(os.environ.get("N", 5000)), int(os.environ.get("LM", 262144))
variants = [int(v) for v in os.environ.get("VARIANTS", "0,3,4").split(",")]
lib = _lib.load()
sh = DeviceShard(n, L)
sh.fill_synthetic()
sh.individual_major()
c32 = torch.empty((sh.np_, sh.np_), dtype=torch.int32, device=sh.dev)
ldpad = int(os.environ.get("LDPAD", 0))
M8p = torch.zeros((sh.np_, sh.Lp + ldpad), dtype=torch.int8, device=sh.dev)
M8p[:, :sh.Lp] = sh.M8
import ctypes as C
lib.eagle_dev_pack_fp4.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_long, C.c_void_p, C.c_void_p]
lib.eagle_dev_mmt_accumulate_f4.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_long, C.c_void_p, C.c_void_p]
lib.eagle_dev_mmt_accumulate_f4.restype = C.c_int
lib.eagle_dev_mmt_accumulate_i8.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_long, C.c_void_p, C.c_void_p]
lib.eagle_dev_mmt_accumulate_i8.restype = C.c_int
M4 = torch.zeros((sh.np_, sh.Lp // 2), dtype=torch.uint8, device=sh.dev)
assert lib.eagle_dev_pack_fp4(sh.ctx, sh.M8.data_ptr(), sh.np_, sh.Lp, sh.Lp, M4.data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
use_f4 = False
def run_syrk(out):
    out.zero_()
    if use_f4:
        rc = lib.eagle_dev_mmt_accumulate_f4(sh.ctx, M4.data_ptr(), sh.np_, sh.Lp, sh.Lp // 2, out.data_ptr(),
                                             C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
        return
    rc = lib.eagle_dev_mmt_accumulate_i8(sh.ctx, M8p.data_ptr(), sh.np_, sh.Lp, sh.Lp + ldpad, out.data_ptr(),
                                      C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
ref = None
res = {v: [] for v in variants}
ops = (sh.np_ * (sh.np_ + 256.0)) * sh.Lp
for rnd in range(5):
    for v in variants:
        lib.eagle_dev_set_tune(sh.ctx, 0 if v == 8 else v % 100)
        use_f4 = v == 8
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c32.zero_()
        e0.record(); run_syrk(c32); e1.record()
        torch.cuda.synchronize()
        if rnd == 0:
            if ref is None:
                ref = c32.clone()
            else:
                if v % 100 not in (3, 4):
                    assert torch.equal(ref, c32), "variant %d differs" % v
        else:
            res[v].append(e0.elapsed_time(e1))
lib.eagle_dev_set_tune(sh.ctx, 0)
for v in variants:
    ms = np.array(res[v])
    print("variant %d: median %.3f ms  min %.3f ms  -> %.0f TOP/s (median)" % (v, np.median(ms), ms.min(), ops / np.median(ms) / 1e9))
