"""Where one AM() run with the device model algebra spends its time: per C-ABI call (PROFILE=1: plus cProfile of the host side)."""
import os, sys, tempfile, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from eagleeverything_amd import am, host_model, synth, rcpp_api
n, L, maxit = 5000, 100000, 3
Mt8 = synth.genotypes_marker_major(n, L, seed=2)
rng = np.random.default_rng(3)
y = 1.0 * Mt8[L // 3] - 0.9 * Mt8[2 * L // 3] + 0.7 * rng.standard_normal(n)
X = np.ones((n, 1))
acc = {}
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); dt = time.perf_counter() - t
        c = acc.setdefault(name, [0, 0.0]); c[0] += 1; c[1] += dt
        return r
    setattr(obj, name, g)
for nm in ("sym_eig", "chol2inv", "inverse", "matmul", "mmt_sqrt_and_sqrtinv", "calculate_a_and_vara_rcpp", "calculateMMt_rcpp", "extract_geno_rcpp", "calculate_reduced_a_rcpp"):
    wrap(rcpp_api, nm)
with tempfile.TemporaryDirectory() as d:
    geno = synth.write_geno_pair(d, Mt8)
    import cProfile, pstats
    pr = cProfile.Profile() if os.environ.get("PROFILE") else None
    t = time.perf_counter()
    if pr: pr.enable()
    r = am.AM(y, X, geno, maxit=maxit, algebra="device")
    if pr: pr.disable()
    tot = time.perf_counter() - t
    if pr: pstats.Stats(pr).sort_stats("tottime").print_stats(18)
print("total %.2f s, picks %s" % (tot, r["all_picks"]))
for k, (c, s) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print("%-28s %3d calls %8.3f s  (%.3f s each)" % (k, c, s, s / c))
print("unaccounted (numpy on host, copies) %.2f s" % (tot - sum(s for _, s in acc.values())))
