"""Marker-file ingestion: getRowColumn, createM_ASCII_rcpp (text and PLINK), createMt_ASCII_rcpp.

CPU part: the oracle (oracle/eagle_oracle_ingest.c) against the reference's own data pair geno.ped <-> geno.txt
(tests/golden/geno_150x100.{ped,txt}: the same genotypes in both encodings) and against hand-built cases for every
branch of the reference's converters.  GPU part (-m gpu): the library through its C ABI against the oracle, byte for byte.
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN

PED = os.path.join(GOLDEN, "geno_150x100.ped")
TXT = os.path.join(GOLDEN, "geno_150x100.txt")


def _lines(path):
    with open(path) as f:
        return f.read().split("\n")[:-1]


def _write(path, rows):
    with open(path, "w") as f:
        for r in rows:
            f.write(r + "\n")
    return str(path)


def random_text_table(rng, n, L, codes=("AA", "AB", "BB"), missing="NA", p_missing=0.02):
    g = rng.integers(0, 3, size=(n, L))
    miss = rng.random((n, L)) < p_missing
    rows, expect = [], []
    for i in range(n):
        toks = [missing if miss[i, j] else codes[g[i, j]] for j in range(L)]
        sep = ["  ", "\t", " "][i % 3]
        rows.append(sep.join(toks) + ("  " if i % 5 == 0 else ""))
        expect.append("".join("1" if miss[i, j] else str(g[i, j]) for j in range(L)))
    return rows, expect


def random_ped(rng, n, L, p_missing=0.03):
    """Biallelic loci with random allele letters; rows carry 6 leading columns like a PLINK ped file."""
    letters = np.array(list("ACGT12"))
    rows = []
    al = np.array([rng.choice(letters, size=2, replace=False) for _ in range(L)])
    for i in range(n):
        toks = ["F%d" % i, "I%d" % i, "0", "0", str(1 + i % 2), "%.3f" % rng.normal()]
        for j in range(L):
            if rng.random() < p_missing:
                toks += [["0", "0"], ["-", "-"], ["0", al[j, 0]], [al[j, 1], "-"]][rng.integers(0, 4)]
            else:
                toks += [al[j, rng.integers(0, 2)], al[j, rng.integers(0, 2)]]
        rows.append(" ".join(toks))
    return rows


# ------------------------------------------------------------------------------------------------ oracle (CPU)
def test_oracle_getRowColumn(oracle, tmp_path):
    assert oracle.getRowColumn(PED) == [150, 206]
    assert oracle.getRowColumn(TXT) == [150, 100]
    p = tmp_path / "t.txt"
    p.write_text("a b  c\nd e f\nlast line without newline")
    assert oracle.getRowColumn(str(p)) == [3, 3]
    with pytest.raises(oracle.OracleError):
        oracle.getRowColumn(str(tmp_path / "absent"))


def test_oracle_text_matches_golden(oracle, golden, tmp_path):
    out = str(tmp_path / "M.ascii")
    ok, info = oracle.createM_ASCII_rcpp(TXT, out, "text", 0, 1, 2, 8, [150, 100])
    assert ok and info["kind"] == "ok"
    got = np.array([[int(c) for c in ln] for ln in _lines(out)], dtype=np.int8)
    assert np.array_equal(got - 1, golden("geno_150x100")["M8"])
    assert _lines(out) == ["".join(ln.split()) for ln in _lines(TXT)]


def test_oracle_plink_pinned_by_reference_pair(oracle, tmp_path):
    """geno.ped and geno.txt hold the same genotypes.  The PLINK converter codes the allele it meets FIRST at a locus as
    allele 0 (CreateASCIInospace_PLINK.cpp:95-106, :174-180), so its output equals geno.txt at every locus whose first
    individual starts with 'A' and is the 0 <-> 2 mirror image where it starts with 'B'."""
    out = str(tmp_path / "M.ascii")
    ok, info = oracle.createM_ASCII_rcpp(PED, out, "PLINK", "-9", "-9", "-9", 8, [150, 206])
    assert ok and not info["missing_seen"]
    got = np.array([[int(c) for c in ln] for ln in _lines(out)])
    exp = np.array([[int(t) for t in ln.split()] for ln in _lines(TXT)])
    first = np.array(_lines(PED)[0].split()[6::2])
    same, flip = (got == exp).all(0), (got + exp == 2).all(0)
    assert (same | flip).all()
    assert np.array_equal(~same, first == "B") and (~same).sum() == 10
    # M M^T is invariant under the mirror image of a column of g-1
    assert np.array_equal((got - 1) @ (got - 1).T, (exp - 1) @ (exp - 1).T)


def test_oracle_text_branches(oracle, tmp_path):
    rng = np.random.default_rng(5)
    rows, expect = random_text_table(rng, 37, 53)
    src, out = _write(tmp_path / "g.txt", rows), str(tmp_path / "M.ascii")
    ok, info = oracle.createM_ASCII_rcpp(src, out, "text", "AA", "AB", "BB", 8, [37, 53], missing="NA")
    assert ok and _lines(out) == expect
    # unknown token: reference returns false, names the token and the 1-based row (CreateASCIInospace.cpp:104-116)
    bad = list(rows)
    bad[11] = bad[11].replace("AB", "XY", 1)
    ok, info = oracle.createM_ASCII_rcpp(_write(tmp_path / "b.txt", bad), out, "text", "AA", "AB", "BB", 8, [37, 53])
    assert not ok and info["kind"] == "token" and info["row"] == 12 and info["token"] == "XY"
    assert _lines(out) == expect[:11]  # rows converted before the failure stay in the file
    # unequal number of columns (:122-131)
    short = list(rows)
    short[20] = " ".join(short[20].split()[:-2])
    ok, info = oracle.createM_ASCII_rcpp(_write(tmp_path / "s.txt", short), out, "text", "AA", "AB", "BB", 8, [37, 53])
    assert not ok and info["kind"] == "columns" and info["row"] == 21 and info["columns"] == 51
    # a missing code equal to a genotype code is taken as the genotype (comparison order BB, AB, AA, missing)
    ok, _ = oracle.createM_ASCII_rcpp(_write(tmp_path / "m.txt", ["0 1 2 9", "9 9 0 1"]), out, "text", 0, 1, 2, 8, [2, 4], missing=9)
    assert ok and _lines(out) == ["0121", "1101"]


def test_oracle_plink_branches(oracle, tmp_path):
    out = str(tmp_path / "M.ascii")
    hdr = "f i 0 0 1 0.5 "
    # first individual B A: allele0 = B; missing pairs become hets; '-' is missing too
    rows = [hdr + "B A  A A  0 0", hdr + "B B  A A  G G", hdr + "A A  A -  T T", hdr + "A B  C C  G T"]
    ok, info = oracle.createM_ASCII_rcpp(_write(tmp_path / "p.ped", rows), out, "PLINK", "-9", "-9", "-9", 8, [4, 12])
    assert ok and info["missing_seen"]
    # locus 2: table starts (A,A), C joins as allele1 -> CC = 2.  locus 3: first row missing -> table (I,I); row 2 "G G"
    # sets allele0 = G; row 3 "T T": T joins as allele1 -> 2; row 4 "G T" het.
    assert _lines(out) == ["101", "000", "212", "121"]
    # a third allele is an error at that locus / individual (:155-161); rows before it stay written
    rows3 = rows + [hdr + "A B  G G  G T"]
    ok, info = oracle.createM_ASCII_rcpp(_write(tmp_path / "q.ped", rows3), out, "PLINK", "-9", "-9", "-9", 8, [5, 12])
    assert not ok and info["kind"] == "alleles" and (info["row"], info["locus"]) == (5, 2)
    assert _lines(out) == ["101", "000", "212", "121"]
    # wrong number of columns (:65-74)
    rowsc = [rows[0], rows[1] + " A"]
    ok, info = oracle.createM_ASCII_rcpp(_write(tmp_path / "c.ped", rowsc), out, "PLINK", "-9", "-9", "-9", 8, [2, 12])
    assert not ok and info["kind"] == "columns" and info["row"] == 2 and info["columns"] == 13


def test_oracle_createMt(oracle, golden, tmp_path):
    m, mt = str(tmp_path / "M.ascii"), str(tmp_path / "Mt.ascii")
    oracle.createM_ASCII_rcpp(TXT, m, "text", 0, 1, 2, 8, [150, 100])
    oracle.createMt_ASCII_rcpp(m, mt, "text", 8, [150, 100])
    got = np.array([[int(c) for c in ln] for ln in _lines(mt)], dtype=np.int8)
    assert got.shape == (100, 150) and np.array_equal(got.T - 1, golden("geno_150x100")["M8"])


# ------------------------------------------------------------------------------------------------ library (GPU)
def _same_file(a, b):
    with open(a, "rb") as fa, open(b, "rb") as fb:
        return fa.read() == fb.read()


@pytest.mark.gpu
def test_gpu_ingest_reference_pair(oracle, golden, tmp_path):
    from eagleeverything_amd import r_api, rcpp_api
    assert rcpp_api.getRowColumn(PED) == [150, 206] and rcpp_api.getRowColumn(TXT) == [150, 100]
    g = golden("geno_150x100")
    for kind, src, kw in (("text", TXT, dict(AA=0, AB=1, BB=2)), ("PLINK", PED, {})):
        d_lib, d_or = tmp_path / ("lib_" + kind), tmp_path / ("or_" + kind)
        d_lib.mkdir(), d_or.mkdir()
        msgs = []
        geno = r_api.ReadMarker(src, type=kind, outdir=str(d_lib), message=msgs.append, quiet=False, **kw)
        assert geno is not None and geno["dim_of_ascii_M"] == [150, 100]
        if kind == "text":
            oracle.createM_ASCII_rcpp(src, str(d_or / "M.ascii"), "text", 0, 1, 2, 8, [150, 100])
        else:
            oracle.createM_ASCII_rcpp(src, str(d_or / "M.ascii"), "PLINK", "-9", "-9", "-9", 8, [150, 206])
        oracle.createMt_ASCII_rcpp(str(d_or / "M.ascii"), str(d_or / "Mt.ascii"), kind, 8, [150, 100])
        assert _same_file(geno["asciifileM"], d_or / "M.ascii") and _same_file(geno["asciifileMt"], d_or / "Mt.ascii")
        assert any("Summary of Marker File" in m for m in msgs)
        # the files' int8 images are resident under the output paths: MM^T and the scan run without parsing text
        MMt = rcpp_api.calculateMMt_rcpp(geno["asciifileM"], 8.0, 4, np.nan, (150, 100))
        assert np.array_equal(MMt, g["MMt"].astype(np.float64))  # PLINK mirror columns leave MM^T unchanged
        out = rcpp_api.calculate_a_and_vara_rcpp(geno["asciifileMt"], np.nan, g["S"], g["V"], 8.0, (100, 150), g["ahat"])
        ref = oracle.calculate_a_and_vara_rcpp(str(d_or / "Mt.ascii"), np.nan, g["S"], g["V"], 8.0, (100, 150), g["ahat"])
        np.testing.assert_allclose(out["a"], ref["a"], rtol=1e-9, atol=1e-12 * np.abs(ref["a"]).max())
        np.testing.assert_allclose(out["vara"], ref["vara"], rtol=9e-7, atol=1e-12 * np.abs(ref["vara"]).max())  # digit budget
        rcpp_api.drop_cache()


@pytest.mark.gpu
@pytest.mark.parametrize("n,L", [(37, 53), (301, 1999), (1030, 777)])
def test_gpu_ingest_random_tables(oracle, tmp_path, n, L):
    from eagleeverything_amd import rcpp_api
    rng = np.random.default_rng(n * 1000 + L)
    rows, expect = random_text_table(rng, n, L)
    src = _write(tmp_path / "g.txt", rows)
    assert rcpp_api.getRowColumn(src) == oracle.getRowColumn(src) == [n, L]
    assert rcpp_api.createM_ASCII_rcpp(src, str(tmp_path / "M.ascii"), "text", "AA", "AB", "BB", 8, [n, L], missing="NA")
    assert _lines(tmp_path / "M.ascii") == expect
    rcpp_api.createMt_ASCII_rcpp(str(tmp_path / "M.ascii"), str(tmp_path / "Mt.ascii"), "text", 8, [n, L])
    oracle.createMt_ASCII_rcpp(str(tmp_path / "M.ascii"), str(tmp_path / "Mt_or.ascii"), "text", 8, [n, L])
    assert _same_file(tmp_path / "Mt.ascii", tmp_path / "Mt_or.ascii")
    # PLINK with missing alleles and random allele letters
    ped = _write(tmp_path / "g.ped", random_ped(rng, n, L))
    msgs = []
    ok = rcpp_api.createM_ASCII_rcpp(ped, str(tmp_path / "P.ascii"), "PLINK", "-9", "-9", "-9", 8, [n, 2 * L + 6], message=msgs.append)
    ok_or, info = oracle.createM_ASCII_rcpp(ped, str(tmp_path / "P_or.ascii"), "PLINK", "-9", "-9", "-9", 8, [n, 2 * L + 6])
    assert ok and ok_or and _same_file(tmp_path / "P.ascii", tmp_path / "P_or.ascii")
    assert info["missing_seen"] == any("missing alleles" in m for m in msgs)
    rcpp_api.drop_cache()


@pytest.mark.gpu
def test_gpu_ingest_errors_match_oracle(oracle, tmp_path):
    from eagleeverything_amd import rcpp_api
    rng = np.random.default_rng(9)
    rows, expect = random_text_table(rng, 64, 40)
    lib_out, or_out = str(tmp_path / "M.ascii"), str(tmp_path / "M_or.ascii")
    bad = list(rows)
    bad[30] = bad[30].replace("BB", "Q", 1)
    bad[45] = " ".join(bad[45].split()[:-1])
    src = _write(tmp_path / "b.txt", bad)
    msgs = []
    assert not rcpp_api.createM_ASCII_rcpp(src, lib_out, "text", "AA", "AB", "BB", 8, [64, 40], message=msgs.append)
    ok, info = oracle.createM_ASCII_rcpp(src, or_out, "text", "AA", "AB", "BB", 8, [64, 40])
    assert not ok and info["kind"] == "token" and info["row"] == 31
    assert "row 31" in rcpp_api.last_error() and any("For example , Q in row 31" in m for m in msgs)
    assert _same_file(lib_out, or_out)
    short = list(rows)
    short[7] = short[7] + " AA"
    src = _write(tmp_path / "s.txt", short)
    msgs = []
    assert not rcpp_api.createM_ASCII_rcpp(src, lib_out, "text", "AA", "AB", "BB", 8, [64, 40], message=msgs.append)
    ok, info = oracle.createM_ASCII_rcpp(src, or_out, "text", "AA", "AB", "BB", 8, [64, 40])
    assert not ok and info["kind"] == "columns" and info["row"] == 8 and info["columns"] == 41
    assert any("row 8 which contains 41" in m for m in msgs) and _same_file(lib_out, or_out)
    # PLINK: third allele; the missing-allele warning is only printed when the missing pair comes first in file order
    ped = random_ped(rng, 50, 30, p_missing=0.0)
    t = ped[33].split()
    t[6 + 2 * 17] = "Z"
    ped[33] = " ".join(t)
    t = ped[40].split()
    t[6] = "0"
    ped[40] = " ".join(t)
    src = _write(tmp_path / "q.ped", ped)
    msgs = []
    assert not rcpp_api.createM_ASCII_rcpp(src, lib_out, "PLINK", "-9", "-9", "-9", 8, [50, 66], message=msgs.append)
    ok, info = oracle.createM_ASCII_rcpp(src, or_out, "PLINK", "-9", "-9", "-9", 8, [50, 66])
    assert not ok and info["kind"] == "alleles" and (info["row"], info["locus"]) == (34, 18) and not info["missing_seen"]
    assert any("snp locus 18 for individual 34" in m for m in msgs) and not any("missing alleles" in m for m in msgs)
    assert _same_file(lib_out, or_out)
    ped[12] = ped[12] + " A"
    src = _write(tmp_path / "c.ped", ped)
    assert not rcpp_api.createM_ASCII_rcpp(src, lib_out, "PLINK", "-9", "-9", "-9", 8, [50, 66])
    ok, info = oracle.createM_ASCII_rcpp(src, or_out, "PLINK", "-9", "-9", "-9", 8, [50, 66])
    assert not ok and info["kind"] == "columns" and info["row"] == 13 and _same_file(lib_out, or_out)
    with pytest.raises(rcpp_api.EagleError):
        rcpp_api.getRowColumn(str(tmp_path / "absent.txt"))
    rcpp_api.drop_cache()


@pytest.mark.gpu
def test_gpu_createMt_streamed_windows(oracle, tmp_path, monkeypatch):
    """M.ascii too large to keep resident (forced by EAGLE_HIP_MAX_RESIDENT_GB): Mt.ascii comes from column windows."""
    from eagleeverything_amd import rcpp_api
    rng = np.random.default_rng(3)
    n, L = 333, 4100
    g = rng.integers(0, 3, size=(n, L))
    m = _write(tmp_path / "M.ascii", ["".join(map(str, r)) for r in g])
    monkeypatch.setenv("EAGLE_HIP_MAX_RESIDENT_GB", "0.0005")
    rcpp_api.drop_cache()
    msgs = []
    rcpp_api.createMt_ASCII_rcpp(m, str(tmp_path / "Mt.ascii"), "text", 8, [n, L], quiet=False, message=msgs.append)
    assert any("block transpose" in s for s in msgs)
    assert _lines(tmp_path / "Mt.ascii") == ["".join(map(str, c)) for c in g.T]
    rcpp_api.drop_cache()


@pytest.mark.gpu
def test_gpu_sidecar_2bit(oracle, golden, tmp_path, monkeypatch):
    """ReadMarker leaves <file>.e2b (2-bit codes) beside both text files; a later process (dropped cache) loads genotypes
    from it instead of parsing text.  A sidecar is trusted only while the text file has the recorded size and mtime."""
    import ctypes as C
    import torch
    from eagleeverything_amd import _lib, r_api, rcpp_api
    g = golden("geno_150x100")
    geno = r_api.ReadMarker(TXT, type="text", AA=0, AB=1, BB=2, outdir=str(tmp_path))
    fM, fMt = geno["asciifileM"], geno["asciifileMt"]
    assert os.path.getsize(fM + ".e2b") == 64 + 150 * 32 and os.path.getsize(fMt + ".e2b") == 64 + 100 * 48
    exp = g["MMt"].astype(np.float64)
    for env in ("1", "0"):  # from the sidecar, then from the text
        monkeypatch.setenv("EAGLE_HIP_SIDECAR", env)
        rcpp_api.drop_cache()
        assert np.array_equal(rcpp_api.calculateMMt_rcpp(fM, 8.0, 4, np.nan, (150, 100)), exp)
        blk = rcpp_api.ReadBlock(fMt, 7, 150, 40)
        assert np.array_equal(blk, g["M8"].T[7:47].astype(np.float64))
    monkeypatch.setenv("EAGLE_HIP_SIDECAR", "1")
    # a column window that does not start on a byte boundary of the packed rows
    L = _lib.load()
    ctx = rcpp_api.context()
    buf = torch.zeros((64, 256), dtype=torch.int8, device="cuda")
    assert L.eagle_dev_load_ascii(ctx, os.fsencode(fM), 10, 50, 5, 77, buf.data_ptr(), 256, 8.0, 4) == 0
    torch.cuda.synchronize()
    assert np.array_equal(buf[:50, :77].cpu().numpy(), g["M8"][10:60, 5:82]) and not buf[50:].any() and not buf[:, 77:].any()
    # proof that the sidecar is what gets read: an invalid code in its payload is reported
    with open(fM + ".e2b", "r+b") as f:
        f.seek(64 + 3)
        f.write(b"\xff")
    rcpp_api.drop_cache()
    with pytest.raises(rcpp_api.EagleError, match="invalid genotype codes"):
        rcpp_api.calculateMMt_rcpp(fM, 8.0, 4, np.nan, (150, 100))
    # a rewritten text file (new mtime) makes the sidecar stale: the text is parsed again
    lines = _lines(fM)
    lines[0] = "2" * 100
    os.utime(fM + ".e2b")
    _write(fM, lines)
    os.utime(fM, ns=(os.stat(fM).st_atime_ns, os.stat(fM).st_mtime_ns + 5_000_000))
    M2 = g["M8"].astype(np.int64).copy()
    M2[0, :] = 1
    assert np.array_equal(rcpp_api.calculateMMt_rcpp(fM, 8.0, 4, np.nan, (150, 100)), (M2 @ M2.T).astype(np.float64))
    rcpp_api.drop_cache()
