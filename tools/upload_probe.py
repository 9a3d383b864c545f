#!/usr/bin/env python3
"""Where the host side of the reference-shaped scan call spends its time (EAGLE_HIP_TIMING=1 prints the library's own breakdown of the
staged upload of V): n x n operands, a small marker file, a few calls.  Usage: tools/upload_probe.py [N] [LM] [calls]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("EAGLE_HIP_TIMING", "1")
import numpy as np
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 4
import tempfile, torch
from eagleeverything_amd import rcpp_api as api, synth
from eagleeverything_amd.sharded import DeviceShard
sh = DeviceShard(n, L); sh.fill_synthetic(seed=2)
tmpd = tempfile.mkdtemp(dir=os.environ.get("TMPDIR", "/tmp"))
geno = synth.write_geno_pair_sidecars(tmpd, sh)
import bench
c32 = sh.mmt_partial()
MMt, _ = sh.mmt_finish(c32, normalise=True)
gen = torch.Generator(device=sh.dev); gen.manual_seed(7)
y = torch.randn(n, generator=gen, device=sh.dev, dtype=torch.float64)
X = torch.ones((n, 1), dtype=torch.float64, device=sh.dev)
St, Vt, at, _, _ = bench.host_operands_torch(torch, MMt, X, y, 1.0, 0.5)     # the model algebra's S, V, a_hat (as bench.py makes them)
S, V, ahat = np.asfortranarray(St.cpu().numpy()), np.asfortranarray(Vt.cpu().numpy()), at.cpu().numpy()
del sh, c32, MMt, St, Vt; torch.cuda.empty_cache()
api.set_scan_mode(1)
for c in range(calls):
    t = time.perf_counter()
    api.calculate_a_and_vara_rcpp(geno["asciifileMt"], np.nan, S, V, 8.0, (L, n), ahat)
    print("call %d: %.1f ms  %s\n   W engine %s" % (c, (time.perf_counter() - t) * 1e3, api.last_scan_timing(), api.last_w_info()), flush=True)
# the raw rates of this host: 16-thread memcpy pageable -> pageable, torch pinned -> device
x = np.empty_like(V); t = time.perf_counter(); np.copyto(x, V); print("numpy copy 800 MB (1 thread): %.1f GB/s" % (V.nbytes / 1e9 / (time.perf_counter() - t)))
p = torch.empty(V.shape, dtype=torch.float64).pin_memory(); d = torch.empty(V.shape, dtype=torch.float64, device="cuda")
for _ in range(2):
    torch.cuda.synchronize(); t = time.perf_counter(); d.copy_(p, non_blocking=True); torch.cuda.synchronize()
    print("pinned -> device: %.1f GB/s" % (V.nbytes / 1e9 / (time.perf_counter() - t)))
for f in os.listdir(tmpd): os.unlink(os.path.join(tmpd, f))
os.rmdir(tmpd)
