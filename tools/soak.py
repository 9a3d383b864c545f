#!/usr/bin/env python3
"""Leak / stability soak of the reference-shaped entry points: many calls on the same files, device memory before and after.
Usage: tools/soak.py [reps] [n] [L]     n >= 3,841 puts W on the int8 engine (pipelined upload of V, helper-thread verification of the
cached S on the host); every 7th call passes ANOTHER S (detected at the end of the call, started over), every 5th another V."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from eagleeverything_amd import rcpp_api, synth

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
L = int(sys.argv[3]) if len(sys.argv) > 3 else 40000
Mt8 = synth.genotypes_marker_major(n, L, seed=1)
rng = np.random.default_rng(0)
if n < 3841:
    A = rng.standard_normal((n, 32)) / 8.0
    S = [np.asfortranarray(np.eye(n) + A @ A.T), None]
    V = [np.asfortranarray(0.5 * np.eye(n) - 0.01 * (A[:, :4] @ A[:, :4].T)), np.asfortranarray(0.6 * np.eye(n) - 0.01 * (A[:, 4:8] @ A[:, 4:8].T))]
    ahat = rng.standard_normal(n)
else:   # the model algebra's operands on the genotypes' own MM^T (as bench.py makes them): what the int8 W engine is built for
    import bench
    M = torch.from_numpy(Mt8).to("cuda").double()
    MMt = M.T @ M
    MMt = MMt / MMt.max() + 0.95 * torch.eye(n, dtype=torch.float64, device="cuda")
    gen = torch.Generator(device="cuda"); gen.manual_seed(7)
    y = torch.randn(n, generator=gen, device="cuda", dtype=torch.float64)
    X = torch.ones((n, 1), dtype=torch.float64, device="cuda")
    St, V0, at, _, _ = bench.host_operands_torch(torch, MMt, X, y, 1.0, 0.5)
    _, V1, _, _, _ = bench.host_operands_torch(torch, MMt, X, y, 1.0, 0.7)
    S = [np.asfortranarray(St.cpu().numpy()), None]
    V = [np.asfortranarray(V0.cpu().numpy()), np.asfortranarray(V1.cpu().numpy())]
    ahat = at.cpu().numpy()
    del M, MMt, St, V0, V1
    torch.cuda.empty_cache()
S[1] = S[0].copy(order="F"); S[1][n // 3, n // 2] += 1e-9; S[1][n // 2, n // 3] += 1e-9       # one entry pair differs in the last bits
with tempfile.TemporaryDirectory() as d:
    geno = synth.write_geno_pair(d, Mt8)
    first = {}
    free0 = None
    t0 = time.time()
    h0, m0 = rcpp_api.scan_operand_cache_stats()
    changes = 0
    last_s = None
    for i in range(reps):
        si, vi = (1 if i % 7 == 6 else 0), (1 if i % 5 == 4 else 0)
        changes += last_s is not None and si != last_s
        last_s = si
        r = rcpp_api.calculate_a_and_vara_rcpp(geno["asciifileMt"], np.nan, S[si], V[vi], 8.0, (L, n), ahat)
        if i % 4 == 0:
            rcpp_api.calculateMMt_rcpp(geno["asciifileM"], 8.0, 4, np.nan, (n, L))
        if i % 10 == 0:
            rcpp_api.extract_geno_rcpp(geno["asciifileM"], 8.0, i % L, (n, L))
        key = (si, vi)
        if key not in first:
            first[key] = (r["a"].copy(), r["vara"].copy())
        else:
            assert np.array_equal(first[key][0], r["a"]) and np.array_equal(first[key][1], r["vara"]), "call %d differs bitwise" % i
        if i == 8:
            torch.cuda.synchronize(); free0 = torch.cuda.mem_get_info()[0]
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    h1, m1 = rcpp_api.scan_operand_cache_stats()
    assert not np.array_equal(first[(0, 0)][0], first[(1, 0)][0]) or reps < 7, "the changed S did not reach the result"
    assert m1 - m0 == changes and h1 - h0 == reps - 1 - changes, ((h0, m0), (h1, m1), changes)   # (the first call fills the cache: neither)
    print("n = %d, L = %d: %d scans (+ MMt + extract) in %.1f s, every (S, V) combination bitwise identical to its first occurrence; S changed %d times: cache hits %d, "
          "misses %d; W engine of the last call int8 = %d (pipelined %d); free device memory after warm-up %.1f MiB, at the end %.1f MiB (delta %.1f MiB)" % (
        n, L, reps, time.time() - t0, changes, h1 - h0, m1 - m0, rcpp_api.last_w_info()["int8"], rcpp_api.last_w_info()["pipelined"],
        free0 / 2**20, free1 / 2**20, (free1 - free0) / 2**20))
