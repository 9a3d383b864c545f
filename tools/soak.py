#!/usr/bin/env python3
"""Leak / stability soak of the reference-shaped entry points: many calls on the same files, device memory before and after."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from eagleeverything_amd import rcpp_api, synth

n, L, reps = 1500, 40000, int(sys.argv[1]) if len(sys.argv) > 1 else 100
Mt8 = synth.genotypes_marker_major(n, L, seed=1)
rng = np.random.default_rng(0)
A = rng.standard_normal((n, 32)) / 8.0
S = np.asfortranarray(np.eye(n) + A @ A.T)
V = np.asfortranarray(0.5 * np.eye(n) - 0.01 * (A[:, :4] @ A[:, :4].T))
ahat = rng.standard_normal(n)
with tempfile.TemporaryDirectory() as d:
    geno = synth.write_geno_pair(d, Mt8)
    first = None
    free0 = None
    t0 = time.time()
    for i in range(reps):
        r = rcpp_api.calculate_a_and_vara_rcpp(geno["asciifileMt"], np.nan, S, V, 8.0, (L, n), ahat)
        if i % 4 == 0:
            rcpp_api.calculateMMt_rcpp(geno["asciifileM"], 8.0, 4, np.nan, (n, L))
        if i % 10 == 0:
            rcpp_api.extract_geno_rcpp(geno["asciifileM"], 8.0, i % L, (n, L))
        if first is None:
            first = (r["a"].copy(), r["vara"].copy())
        else:
            assert np.array_equal(first[0], r["a"]) and np.array_equal(first[1], r["vara"]), "call %d differs bitwise" % i
        if i == 4:
            torch.cuda.synchronize(); free0 = torch.cuda.mem_get_info()[0]
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    print("%d scans + MMt + extract: %.1f s, bitwise identical; free device memory after warm-up %.1f MiB, at the end %.1f MiB (delta %.1f MiB)" % (
        reps, time.time() - t0, free0 / 2**20, free1 / 2**20, (free1 - free0) / 2**20))
