#!/usr/bin/env python3
"""One AM() run (planted QTL, synthetic genotypes) with the model algebra on host LAPACK and on the GPU:
where the time of an iteration goes once the scan takes milliseconds."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from eagleeverything_amd import am, host_model, synth

n, L, maxit = int(sys.argv[1]) if len(sys.argv) > 1 else 3000, int(sys.argv[2]) if len(sys.argv) > 2 else 60000, 3
Mt8 = synth.genotypes_marker_major(n, L, seed=2)
rng = np.random.default_rng(3)
y = 1.0 * Mt8[L // 3] - 0.9 * Mt8[2 * L // 3] + 0.7 * rng.standard_normal(n)
X = np.ones((n, 1))
with tempfile.TemporaryDirectory() as d:
    geno = synth.write_geno_pair(d, Mt8)
    out = {}
    for kind in ("device", "host"):
        t = time.perf_counter()
        try:
            r = am.AM(y, X, geno, maxit=maxit, algebra=kind)
        finally:
            host_model.set_algebra("host")
        out[kind] = (time.perf_counter() - t, r["all_picks"], r["extBIC_trace"])
        print("%-6s algebra: %.2f s for %d iterations, picks %s" % (kind, out[kind][0], maxit, r["all_picks"]))
    assert out["host"][1] == out["device"][1]
    print("extBIC trace max rel diff %.2e" % np.max(np.abs(np.array(out["host"][2]) - np.array(out["device"][2])) / np.abs(np.array(out["host"][2]))))
