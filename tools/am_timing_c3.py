#!/usr/bin/env python3
"""What ONE forward-selection iteration of AM() (E/R/AM.R:400-475) costs at the headline size, phase by phase, with the model algebra on
host LAPACK, on the device (rocSOLVER / the library's fp64 GEMM through the C ABI) and with the scan in the eigenbasis
of MM^T (SpectralBackend): calcMMt, chol2inv, the two eigen-decompositions of emma.REMLE / emma.MLE, the grid + root searches, the
operands of find_qtl (H, P, MMt^{+-1/2}, a_hat, Var a_hat), the scan itself.  Genotypes: N x LM synthetic, files = sparse
placeholders + 2-bit sidecars (what the library reads); planted QTL, so the loop has something to pick.

Usage: tools/am_timing_c3.py [N] [LM] [kinds]     kinds: comma list of device,spectral,host (host at N = 10,000 takes minutes per iteration)
"""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
kinds = (sys.argv[3] if len(sys.argv) > 3 else "device,spectral,host").split(",")
maxit = {"device": int(os.environ.get("MAXIT", 2)), "spectral": int(os.environ.get("MAXIT", 2)), "host": int(os.environ.get("MAXIT_HOST", 1))}


def log(msg):
    print("[am_timing %7.1fs] %s" % (time.time() - T0, msg), file=sys.stderr, flush=True)


def main():
    import torch
    from eagleeverything_amd import am, host_model, r_api, rcpp_api, synth
    from eagleeverything_amd.sharded import DeviceShard
    # host BLAS / LAPACK on the cores the cgroup gives this process (a team of os.cpu_count() threads on a 16-core share crawls)
    cores = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            cores = max(1, int(float(q) / float(per) + 0.5))
    except (OSError, ValueError):
        pass
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=min(cores, 64))
    except Exception:
        pass
    sh = DeviceShard(n, L)
    sh.fill_synthetic(seed=2)
    q1, q2 = L // 3, 2 * L // 3
    rng = np.random.default_rng(3)
    g1, g2 = sh.Mt8[q1, :n].cpu().numpy().astype(np.float64), sh.Mt8[q2, :n].cpu().numpy().astype(np.float64)
    y = 1.0 * g1 - 0.9 * g2 + 0.7 * rng.standard_normal(n)
    X = np.ones((n, 1))
    tmpd = tempfile.mkdtemp(dir=os.environ.get("TMPDIR", "/tmp"))
    geno = synth.write_geno_pair_sidecars(tmpd, sh)
    del sh
    torch.cuda.empty_cache()
    log("genotypes %d x %d written as sidecars" % (n, L))
    out = {"n": n, "L": L, "planted": [q1 + 1, q2 + 1], "runs": {}}
    acc = {}

    def wrap(obj, name, label=None):
        f = getattr(obj, name)

        def g(*a, **k):
            t = time.perf_counter()
            r = f(*a, **k)
            c = acc.setdefault(label or name, [0, 0.0])
            c[0] += 1
            c[1] += time.perf_counter() - t
            return r
        setattr(obj, name, g)
        return f

    saved = []
    for obj, name, label in ((host_model, "_chol2inv", "chol2inv(MMt) [AM.R:422] + inside calculateP"),
                             (am, "emma_eigen_L_wo_Z", "eigen(K) for emma.MLE (once per run here; every iteration in the reference)"),
                             (am, "emma_eigen_R_wo_Z", "eigen(S (K + I) S) for emma.REMLE / emma.MLE"),
                             (am, "calcVC", "emma.REMLE grid + roots (given the decomposition)"),
                             (am, "calc_extBIC", "emma.MLE grid + roots + extBIC (given the decompositions)"),
                             (host_model, "calculateP", "find_qtl: calculateP"),
                             (host_model, "calculateMMt_sqrt_and_sqrtinv", "find_qtl: MMt^(1/2), MMt^(-1/2)"),
                             (host_model, "calculate_reduced_vara", "find_qtl: Var(a_hat)"),
                             (host_model, "calculate_reduced_a", "find_qtl: a_hat"),
                             (rcpp_api, "calculate_a_and_vara_rcpp", "find_qtl: the scan (eagle_calculate_a_and_vara, PCIe included)"),
                             (rcpp_api, "calculateMMt_rcpp", "calcMMt (eagle_calculateMMt, cold: sidecar -> HBM -> MM^T -> host)"),
                             (rcpp_api, "spectral_prepare", "spectral: Z = Mt U (once per run)"),
                             (rcpp_api, "spectral_scan", "spectral: the scan (one pass over Z)"),
                             (rcpp_api, "extract_geno_rcpp", "extract_geno")):
        saved.append((obj, name, wrap(obj, name, label)))
    for kind in kinds:
        acc.clear()
        rcpp_api.drop_cache()
        backend = am.SpectralBackend() if kind == "spectral" else am.HipBackend()
        t = time.perf_counter()
        try:
            r = am.AM(y, X, geno, maxit=maxit[kind], backend=backend, algebra="host" if kind == "host" else "device")
        finally:
            host_model.set_algebra("host")
        tot = time.perf_counter() - t
        phases = {k: {"calls": c, "seconds": round(s, 3)} for k, (c, s) in sorted(acc.items(), key=lambda kv: -kv[1][1])}
        out["runs"][kind] = {"iterations": maxit[kind], "total_s": round(tot, 2), "picks": r["all_picks"], "extBIC_trace": r["extBIC_trace"],
                             "phases": phases, "w_engine_of_the_last_scan": rcpp_api.last_w_info() if kind != "spectral" else None,
                             "algebra": "host LAPACK (%d threads)" % min(cores, 64) if kind == "host" else "device (C ABI section 1c)"}
        log("%s: %.1f s for %d iteration(s), picks %s" % (kind, tot, maxit[kind], r["all_picks"]))
        for k, v in phases.items():
            log("    %-86s %3d x  %8.3f s" % (k, v["calls"], v["seconds"]))
    ks = [k for k in kinds if k in out["runs"]]
    m = min(len(out["runs"][k]["picks"]) for k in ks)
    out["same_picks_in_every_run (over the iterations all of them made)"] = all(out["runs"][k]["picks"][:m] == out["runs"][ks[0]]["picks"][:m] for k in ks)
    for f in os.listdir(tmpd):
        os.unlink(os.path.join(tmpd, f))
    os.rmdir(tmpd)
    print(json.dumps(out))


if __name__ == "__main__":
    T0 = time.time()
    main()
