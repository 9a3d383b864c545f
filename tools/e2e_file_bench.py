#!/usr/bin/env python3
"""PCIe- and file-inclusive timing of the reference-shaped entry points (host files -> host results) on the GPU box.
Not the bench metric (bench.py times with genotypes resident).
Operands here are the SIMPLE seeded S, V (no model algebra): the library picks 5-6 digit slices for them instead of the 4 of the
bench's operands, so scan times are not comparable with bench.py's step (that was round 2's unexplained '100 ms gap'); bench.py's
e2e_reference_shaped leg and --form abi time the same path on the bench's own operands."""
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
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
    S = np.eye(n) + A @ A.T
    V = 0.5 * np.eye(n) - 0.01 * (A[:, :8] @ A[:, :8].T)
    ahat = rng.standard_normal(n)
    S, V = np.asfortranarray(S), np.asfortranarray(V)  # R hands column-major matrices: no host-side re-layout
    out = {"n": n, "L": L}
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as d:
        t = time.perf_counter()
        geno = synth.write_geno_pair(d, Mt8)
        out["write_files_s"] = time.perf_counter() - t
        for name, fn in (("mmt", lambda: rcpp_api.calculateMMt_rcpp(geno["asciifileM"], 8.0, 16, np.nan, (n, L))),
                         ("scan", lambda: rcpp_api.calculate_a_and_vara_rcpp(geno["asciifileMt"], np.nan, S, V, 8.0, (L, n), ahat))):
            t = time.perf_counter(); fn(); cold = time.perf_counter() - t
            t = time.perf_counter(); fn(); warm = time.perf_counter() - t
            out[name + "_cold_s"] = cold   # text file (page cache) -> pinned -> HBM -> decode -> compute -> D2H
            out[name + "_warm_s"] = warm   # genotypes already resident in HBM; operands H2D, results D2H
            rcpp_api.drop_cache()
            t = time.perf_counter(); fn(); out[name + "_reload_s"] = time.perf_counter() - t  # cold again, staging buffers already pinned
        # ingestion: Mt.ascii (+ its 2-bit sidecar) rebuilt from M.ascii on the device, then reloads with / without the sidecar
        rcpp_api.drop_cache()
        mt2 = os.path.join(d, "Mt2.ascii")
        t = time.perf_counter(); rcpp_api.createMt_ASCII_rcpp(geno["asciifileM"], mt2, "text", 8.0, (n, L)); out["createMt_s"] = time.perf_counter() - t
        with open(mt2, "rb") as fa, open(geno["asciifileMt"], "rb") as fb:
            out["createMt_identical"] = fa.read() == fb.read()
        scan2 = lambda: rcpp_api.calculate_a_and_vara_rcpp(mt2, np.nan, S, V, 8.0, (L, n), ahat)
        t = time.perf_counter(); scan2(); out["scan_after_createMt_s"] = time.perf_counter() - t  # image already resident
        for env in ("1", "0"):
            os.environ["EAGLE_HIP_SIDECAR"] = env
            rcpp_api.drop_cache()
            t = time.perf_counter(); scan2(); out["scan_reload_sidecar%s_s" % env] = time.perf_counter() - t
        os.environ.pop("EAGLE_HIP_SIDECAR")
        # out-of-core: genotypes never resident, streamed in ~19k-marker chunks whose reads overlap the kernels
        os.environ["EAGLE_HIP_MAX_RESIDENT_GB"] = "0.2"
        for env in ("1", "0"):
            os.environ["EAGLE_HIP_SIDECAR"] = env
            rcpp_api.drop_cache()
            scan2()
            t = time.perf_counter(); r = scan2(); out["scan_streamed_sidecar%s_s" % env] = time.perf_counter() - t
        os.environ.pop("EAGLE_HIP_SIDECAR"); os.environ.pop("EAGLE_HIP_MAX_RESIDENT_GB")
        out["sidecar_bytes"] = os.path.getsize(mt2 + ".e2b")
        if os.environ.get("E2E_SKIP_READMARKER"):  # (the headline size: 20 GB of text table more)
            out["scan_cold_markers_per_s"] = L / out["scan_cold_s"]
            out["scan_warm_markers_per_s"] = L / out["scan_warm_s"]
            out["file_bytes_each"] = os.path.getsize(geno["asciifileM"])
            print(json.dumps(out))
            return
        # ReadMarker() on a whitespace-separated text table of the same genotypes (2 bytes per genotype)
        G = (Mt8.T + 1).astype(np.uint8)                      # n x L codes 0/1/2
        txt = np.empty((n, 2 * L), dtype=np.uint8)
        txt[:, 0::2] = G + ord("0")
        txt[:, 1::2] = ord(" ")
        txt[:, -1] = ord("\n")
        raw = os.path.join(d, "geno.txt")
        txt.tofile(raw)
        del txt, G
        from eagleeverything_amd import r_api
        rd = os.path.join(d, "rm")
        os.mkdir(rd)
        rcpp_api.drop_cache()
        t = time.perf_counter(); g2 = r_api.ReadMarker(raw, type="text", AA=0, AB=1, BB=2, outdir=rd); out["ReadMarker_text_s"] = time.perf_counter() - t
        out["ReadMarker_input_bytes"] = os.path.getsize(raw)
        with open(g2["asciifileMt"], "rb") as fa, open(geno["asciifileMt"], "rb") as fb:
            out["ReadMarker_Mt_identical"] = fa.read() == fb.read()
        out["scan_cold_markers_per_s"] = L / out["scan_cold_s"]
        out["scan_warm_markers_per_s"] = L / out["scan_warm_s"]
        out["file_bytes_each"] = os.path.getsize(geno["asciifileM"])
    print(json.dumps(out))


if __name__ == "__main__":
    main()
