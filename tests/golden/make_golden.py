#!/usr/bin/env python3
"""Generate tests/golden/*.npz.  Run in the BUILD CONTAINER only (reads /root/reference as data).

Inputs are the reference's own demo data files (text genotype / phenotype tables, read as DATA):
  MyPackage/geno.txt + pheno.txt            150 x 100
  MyPackage/genoDemo.dat + phenoDemo.dat    150 x 4998
plus one seeded synthetic case.  Expected outputs come from the independent numpy restatement
(oracle/oracle_np.py, exact int64 for MM^T, OpenBLAS fp64 otherwise) because the reference itself
cannot run here and records no outputs (parity unpinned).  The MM^T known answers
of SURVEY.md section 4 are asserted while generating.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from eagleeverything_amd import host_model, synth  # noqa: E402
from oracle import oracle_np  # noqa: E402

REF = "/root/reference/MyPackage"
OUT = os.path.dirname(os.path.abspath(__file__))

KNOWN = {  # MMt[0,0], MMt[0,1], MMt[149,149], max, min, trace, sum
    "geno_150x100": (63, 40, 69, 89, 10, 9748, 803130),
    "genoDemo_150x4998": (3643, 2351, 3711, 3815, 1950, 551356, 50562338),
}


def load_geno(path):
    g = np.loadtxt(path, dtype=np.int64)
    assert set(np.unique(g)) <= {0, 1, 2}
    return (g - 1).astype(np.int8)  # ReadBlock.cpp:53-54


def load_pheno(path, cols):
    with open(path) as f:
        hdr = f.readline().split()
        rows = [ln.split() for ln in f if ln.strip()]
    idx = [hdr.index(c) for c in cols]
    return np.array([[float(r[i]) for i in idx] for r in rows])


def make_case(name, M8, y, X, varE, varG, sel=np.nan):
    n, L = M8.shape
    MMt = oracle_np.mmt_int64(M8)
    if name in KNOWN:
        k = KNOWN[name]
        got = (MMt[0, 0], MMt[0, 1], MMt[n - 1, n - 1], MMt.max(), MMt.min(), np.trace(MMt), MMt.sum())
        assert tuple(int(v) for v in got) == k, (name, got, k)
    MMtn = oracle_np.normalise(MMt)
    ops = host_model.scan_operands(MMtn, X, y, varE, varG)
    Mt8 = np.ascontiguousarray(M8.T)
    a, vara = oracle_np.a_and_vara(Mt8, ops["S"], ops["V"], ops["ahat"])
    tsq, idx, mx = oracle_np.tsq_argmax(a, vara)
    ar = oracle_np.reduced_a(Mt8, varG, ops["P"], y)
    # masked variants (selected loci given 0-based and non-NA => masking fires)
    selm = np.array([3.0, 17.0, float(L - 1)])
    MMt_m = oracle_np.mmt_int64(M8, selm)
    a_m, vara_m = oracle_np.a_and_vara(Mt8, ops["S"], ops["V"], ops["ahat"], selm)
    tsq_m, idx_m, mx_m = oracle_np.tsq_argmax(a_m, vara_m)
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"),
        M8=M8, y=y, X=X, varE=varE, varG=varG, MMt=MMt, MMt_norm_max=float(MMt.max()),
        S=ops["S"], V=ops["V"], ahat=ops["ahat"], P=ops["P"],
        a=a, vara=vara, tsq=tsq, argmax=idx, tsqmax=mx, ar=ar,
        sel_masked=selm, MMt_masked=MMt_m, a_masked=a_m, vara_masked=vara_m, argmax_masked=idx_m)
    print(name, "n,L=", (n, L), "argmax", idx, "tsqmax", mx, "argmax_masked", idx_m,
          "min eig(MMt_norm)=%.4f" % np.linalg.eigvalsh(MMtn).min())


def main():
    M8 = load_geno(os.path.join(REF, "geno.txt"))
    ph = load_pheno(os.path.join(REF, "pheno.txt"), ["y", "cov1", "cov2"])
    X = np.column_stack([np.ones(len(ph)), ph[:, 1], ph[:, 2]])
    make_case("geno_150x100", M8, ph[:, 0], X, 1.0, 0.5)

    M8 = load_geno(os.path.join(REF, "genoDemo.dat"))
    ph = load_pheno(os.path.join(REF, "phenoDemo.dat"), ["trait1", "pc1", "pc2"])
    X = np.column_stack([np.ones(len(ph)), ph[:, 1], ph[:, 2]])
    make_case("genoDemo_150x4998", M8, ph[:, 0], X, 2.0, 1.5)

    # seeded synthetic, ragged sizes (not multiples of any tile)
    Mt8 = synth.genotypes_marker_major(203, 1531, seed=11)
    y, _ = synth.trait(Mt8)
    X = np.ones((203, 1))
    make_case("synth_203x1531", np.ascontiguousarray(Mt8.T), y, X, 1.0, 0.7)

    # ingestion fixtures: the reference's own data pair, the same 150 x 100 genotypes as a PLINK ped file and as a
    # 0/1/2 text table (DATA files, copied byte for byte)
    import shutil
    shutil.copyfile(os.path.join(REF, "geno.ped"), os.path.join(OUT, "geno_150x100.ped"))
    shutil.copyfile(os.path.join(REF, "geno.txt"), os.path.join(OUT, "geno_150x100.txt"))


if __name__ == "__main__":
    main()
