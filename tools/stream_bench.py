#!/usr/bin/env python3
"""Out-of-core (streamed) MM^T build and marker scan through the reference-shaped entry points: storage rate, loader and kernel
seconds and their overlap (SURVEY 8(d) "Streaming"), for the text files and for their 2-bit sidecars.  File- and PCIe-inclusive:
never bench.py's value.  Usage: tools/stream_bench.py [n] [L] [chunk budget GB]"""
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 300000
    budget = sys.argv[3] if len(sys.argv) > 3 else "1.0"
    import torch
    from eagleeverything_amd import rcpp_api, synth
    from eagleeverything_amd.sharded import DeviceShard
    sh = DeviceShard(n, L)
    sh.fill_synthetic()
    Mt8 = sh.Mt8[:L, :n].cpu().numpy()
    del sh
    torch.cuda.empty_cache()
    rng = np.random.default_rng(0)
    A = rng.standard_normal((n, 64)) / 8.0
    S = np.asfortranarray(np.eye(n) + A @ A.T)
    V = np.asfortranarray(0.5 * np.eye(n) - 0.01 * (A[:, :8] @ A[:, :8].T))
    ahat = rng.standard_normal(n)
    out = {"n": n, "L": L, "chunk_budget_GB": float(budget), "host_threads": os.cpu_count()}
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as d:
        geno = synth.write_geno_pair(d, Mt8)
        del Mt8
        # 2-bit sidecars next to the text files (what ReadMarker() leaves behind)
        mt2 = os.path.join(d, "Mt2.ascii")
        rcpp_api.createMt_ASCII_rcpp(geno["asciifileM"], mt2, "text", 8.0, (n, L))
        out["text_bytes"] = os.path.getsize(mt2)
        out["sidecar_bytes"] = os.path.getsize(mt2 + ".e2b")
        rcpp_api.drop_cache()
        resident = {}
        t = time.perf_counter(); r0 = rcpp_api.calculate_a_and_vara_rcpp(mt2, np.nan, S, V, 8.0, (L, n), ahat); resident["scan_cold_s"] = time.perf_counter() - t
        a0, v0 = np.asarray(r0["a"]).ravel(), np.asarray(r0["vara"]).ravel()
        t = time.perf_counter(); rcpp_api.calculate_a_and_vara_rcpp(mt2, np.nan, S, V, 8.0, (L, n), ahat); resident["scan_warm_s"] = time.perf_counter() - t
        out["resident"] = resident
        os.environ["EAGLE_HIP_MAX_RESIDENT_GB"] = budget
        for side in ("1", "0"):
            os.environ["EAGLE_HIP_SIDECAR"] = side
            rcpp_api.drop_cache()
            key = "sidecar_2bit" if side == "1" else "text"
            t = time.perf_counter(); r1 = rcpp_api.calculate_a_and_vara_rcpp(mt2, np.nan, S, V, 8.0, (L, n), ahat); wall = time.perf_counter() - t
            a1, v1 = np.asarray(r1["a"]).ravel(), np.asarray(r1["vara"]).ravel()
            st = rcpp_api.last_stream_stats()
            st["call_wall_s"] = wall
            st["markers_per_s"] = L / wall
            st["a_equal_resident"] = bool(np.array_equal(a0, a1))
            st["vara_max_rel_vs_resident"] = float(np.max(np.abs(v1 - v0) / np.abs(v0)))
            out["scan_streamed_" + key] = st
            if side == "0":  # M.ascii as written above has no sidecar: column windows of every text line
                rcpp_api.drop_cache()
                t = time.perf_counter(); rcpp_api.calculateMMt_rcpp(geno["asciifileM"], 8.0, os.cpu_count(), np.nan, (n, L)); wall = time.perf_counter() - t
                st = rcpp_api.last_stream_stats()
                st["call_wall_s"] = wall
                out["mmt_streamed_text"] = st
        os.environ.pop("EAGLE_HIP_SIDECAR"); os.environ.pop("EAGLE_HIP_MAX_RESIDENT_GB")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
