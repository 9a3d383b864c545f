"""Seeded synthetic genotypes / traits of the shape BASELINE.json names.

Per marker j an allele frequency p_j ~ U(0.05, 0.5), then g_ij ~ Binomial(2, p_j) (HWE); the stream
is keyed by (seed, marker block) so any marker shard can be generated without the rest.
Values are returned as int8 in {-1,0,1} = g-1 (E/src/ReadBlock.cpp:53-54) or written as the
fixed-width '0','1','2' text the reference's readers consume (M.ascii: n lines x L chars,
Mt.ascii: L lines x n chars; E/R/ReadMarker.R:306-307).
"""
import os

import numpy as np

BLOCK = 4096  # markers per independently seeded block


def genotypes_marker_major(n, L, seed=20240601, first_marker=0):
    """(L x n) int8 in {-1,0,1}; rows [first_marker, first_marker+L) of the global marker stream."""
    out = np.empty((L, n), dtype=np.int8)
    m = first_marker
    end = first_marker + L
    while m < end:
        blk = m // BLOCK
        b0 = blk * BLOCK
        rng = np.random.Generator(np.random.Philox(key=seed, counter=[blk, 0, 0, 0]))
        p = rng.uniform(0.05, 0.5, size=BLOCK)
        lo = m - b0
        hi = min(end, b0 + BLOCK) - b0
        # draw the whole block row-wise so a sub-range reproduces the same values
        g = rng.binomial(2, p[:, None], size=(BLOCK, n)).astype(np.int8)
        out[m - first_marker:m - first_marker + (hi - lo)] = g[lo:hi] - 1
        m = b0 + hi
    return out


def trait(Mt8, nqtl=10, beta=0.5, seed=7):
    L, n = Mt8.shape
    idx = np.linspace(0, L - 1, nqtl + 2, dtype=np.int64)[1:-1]
    rng = np.random.default_rng(seed)
    y = beta * Mt8[idx].astype(np.float64).sum(axis=0) + rng.standard_normal(n)
    return y, idx


def write_ascii(path, G8):
    """Write int8 {-1,0,1} rows as fixed-width '0','1','2' lines (CreateASCIInospace.cpp:108-118 layout)."""
    G8 = np.asarray(G8, dtype=np.int8)
    rows, cols = G8.shape
    buf = np.empty((rows, cols + 1), dtype=np.uint8)
    buf[:, :cols] = (G8 + 1 + ord("0")).astype(np.uint8)
    buf[:, cols] = ord("\n")
    with open(path, "wb") as f:
        f.write(buf.tobytes())
    return path


def write_geno_pair(dirname, Mt8, stem=""):
    """Write M.ascii (n x L) and Mt.ascii (L x n); returns the reference's `geno` list as a dict."""
    L, n = Mt8.shape
    fM = os.path.join(dirname, stem + "M.ascii")
    fMt = os.path.join(dirname, stem + "Mt.ascii")
    write_ascii(fM, np.ascontiguousarray(Mt8.T))
    write_ascii(fMt, Mt8)
    return {"asciifileM": fM, "asciifileMt": fMt, "dim_of_ascii_M": (n, L)}


def write_sidecar_from_device(lib, ctx, image, rows, cols, path_text, block_rows=32768):
    """Benchmark-size genotype files without writing 8 bits per genotype of text: `<path_text>.e2b`, the 2-bit sidecar the
    converters leave beside every text file (csrc/eagle_ingest.cpp; layout E2bHeader in csrc/eagle_ctx.h), packed from the
    resident int8 image (a torch tensor [>= rows][ld], values -1/0/1) with the library's own kernel, next to a SPARSE placeholder
    of the text file's exact size.  A sidecar is trusted while it records the text file's size and mtime, and a load that
    finds a valid one never reads the text (eagle_dev_load_ascii), so the placeholder's holes are never touched."""
    import ctypes as C
    import struct

    import torch
    with open(path_text, "wb") as f:
        f.truncate(int(rows) * (int(cols) + 1))
        f.seek(int(cols))
        f.write(b"\n")   # the end of line 1: what the loader probes for the line width (a file of holes has no line end to find)
    st = os.stat(path_text)
    row_bytes = (int(cols) + 3) // 4
    hdr = struct.pack("<8sIIQQQQqQ", b"EAGLE2B\0", 1, 0, int(rows), int(cols), row_bytes, st.st_size, st.st_mtime_ns, 0)
    assert len(hdr) == 64
    pack = lib.eagle_dev_pack2b
    pack.restype = C.c_int
    pack.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_long, C.c_void_p, C.c_long, C.c_void_p]
    ld = image.stride(0)
    buf = torch.empty((min(block_rows, int(rows)), row_bytes), dtype=torch.uint8, device=image.device)
    stream = C.c_void_p(torch.cuda.current_stream(image.device).cuda_stream)
    with open(path_text + ".e2b", "wb") as f:
        f.write(hdr)
        for r0 in range(0, int(rows), block_rows):
            nr = min(block_rows, int(rows) - r0)
            rc = pack(ctx, image[r0:].data_ptr(), nr, int(cols), ld, buf.data_ptr(), row_bytes, stream)
            if rc != 0:
                raise RuntimeError("eagle_dev_pack2b failed: %d" % rc)
            f.write(buf[:nr].cpu().numpy().tobytes())
        f.flush()
        os.fsync(f.fileno())
    return os.path.getsize(path_text + ".e2b")


def write_geno_pair_sidecars(dirname, sh, stem=""):
    """M.ascii / Mt.ascii of a DeviceShard's genotypes as sparse placeholders + 2-bit sidecars (see write_sidecar_from_device);
    returns the reference's `geno` list as a dict."""
    fM = os.path.join(dirname, stem + "M.ascii")
    fMt = os.path.join(dirname, stem + "Mt.ascii")
    write_sidecar_from_device(sh.L, sh.ctx, sh.Mt8, sh.Lloc, sh.n, fMt)
    M8 = sh.individual_major()
    write_sidecar_from_device(sh.L, sh.ctx, M8, sh.n, sh.Lloc, fM)
    return {"asciifileM": fM, "asciifileMt": fMt, "dim_of_ascii_M": (sh.n, sh.Lloc)}
