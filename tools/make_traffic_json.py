#!/usr/bin/env python3
"""profiles/<prof dir>/summary.txt (tools/profile_gpu.sh + summarise_prof.py) -> profiles/r04_traffic.json: per-launch HBM-side
bytes, L2 hit rate and MFMA-busy fraction of the dominant kernels, tagged with the hash of the kernel sources they were
measured on (bench.py attaches roofline.traffic only while that hash and the configuration still match).
Usage: tools/make_traffic_json.py profiles/<dir> n markers slices"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_sha16  # noqa: E402

d, n, markers, slices = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
d2 = sys.argv[5] if len(sys.argv) > 5 else None   # profile of a run WITH bench.py's secondary entries: the opt-in kernels are taken from it
vals = {}


def read(dirname, into):
    for line in open(os.path.join(dirname, "summary.txt")):
        m = re.match(r"(\S+)\s+(\S+)\s+n=(\d+)\s+mean=(\S+)", line)
        if m:
            into[(m.group(1), m.group(2))] = float(m.group(4))
        m = re.match(r"(\S+)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.e+-]+)(\s+[\d.]+\s+([\d.]+))?\s*$", line)
        if m and (m.group(1), "avg_ms") not in into:
            into[(m.group(1), "avg_ms")] = float(m.group(4))
            if m.group(7):
                into[(m.group(1), "max_ms")] = float(m.group(7))


read(d, vals)
vals2 = {}
if d2:
    read(d2, vals2)

def entry(k, algorithmic, secondary=False):
    global vals
    keep = vals
    if secondary:
        if not d2:
            return None
        vals = vals2
    try:
        return _entry(k, algorithmic, d2 if secondary else d)
    finally:
        vals = keep


def _entry(k, algorithmic, src):
    e = {"config": {"n": n, "markers": markers, "slices": slices}, "source": src.rstrip("/")}
    f, w = vals.get((k, "FETCH_SIZE")), vals.get((k, "WRITE_SIZE"))
    if f is not None and w is not None:
        e.update({"FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "hbm_side_bytes": (2 * f + w) * 1024,
                  "note": "(2 x FETCH_SIZE + WRITE_SIZE) x 1024: gfx950 FETCH_SIZE counts 64 B per 128-B request (MI355X_MICROARCH.md, HBM); "
                          "separate --pmc passes; Infinity Cache hits are included in FETCH_SIZE"})
    if (k, "TCC_REQ_sum") in vals:
        e["TCC_hit_rate"] = vals[(k, "TCC_HIT_sum")] / vals[(k, "TCC_REQ_sum")]
    if (k, "GRBM_GUI_ACTIVE") in vals and (k, "SQ_VALU_MFMA_BUSY_CYCLES") in vals:
        e["GRBM_GUI_ACTIVE_sum_over_8_XCD"] = vals[(k, "GRBM_GUI_ACTIVE")]
        e["mfma_busy_frac_of_gpu_cycles"] = vals[(k, "SQ_VALU_MFMA_BUSY_CYCLES")] / (vals[(k, "GRBM_GUI_ACTIVE")] * 128)
    if (k, "SQ_LDS_BANK_CONFLICT") in vals:
        e["SQ_LDS_BANK_CONFLICT"] = vals[(k, "SQ_LDS_BANK_CONFLICT")]
        e["SQ_LDS_IDX_ACTIVE"] = vals.get((k, "SQ_LDS_IDX_ACTIVE"))
    if (k, "avg_ms") in vals:
        e["avg_ms"] = vals[(k, "avg_ms")]
    if (k, "max_ms") in vals:
        e["max_ms"] = vals[(k, "max_ms")]
    e["algorithmic_bytes"] = algorithmic
    return e


npad = (n + 255) // 256 * 256
lpad = (markers + 255) // 256 * 256
out = {"kernel_sha16": kernel_sha16(),
       "k_vara_i8": entry("k_vara_i8p", float(lpad) * npad + slices * npad * npad / 2.0),
       # per launch shape: <true,true> = the genotype pass of the digit-slice scan (a, the diagonal term and m^T rho in one read)
       "k_gemv_mfma": entry("k_gemv_mfma<true,true>", float(lpad) * npad),
       "k_gemv_mfma<false,false>": entry("k_gemv_mfma<false,false>", float(lpad) * npad),
       # the int8 digit-slice products of W = S V S: per launch the digit panels of both operands (pairs x 2 x 256-row panels, shared) -- algorithmic = the slices read once
       "k_w8_gemm_p": entry("k_w8_gemm_p", 2.0 * 6 * npad * npad),
       "k_w8_combine1": entry("k_w8_combine1", (16.0 + 8.0 + 4.0 * 5) * npad * npad),
       "k_syrk_f4": entry("k_syrk_f4w", float(lpad) * npad / 2.0),
       "k_gemm_f64_dma": entry("k_gemm_f64_dma", 3.0 * 8 * npad * npad),
       "k_transpose_pack_fp4": entry("k_transpose_pack_fp4", 1.5 * float(lpad) * npad),
       "k_vara_f64d": entry("k_vara_f64d", float(lpad) * npad + 8.0 * npad * npad / 2.0, True),
       "k_gram_rowabs_i8": entry("k_gram_rowabs_i8", float(npad) * npad),
       "k_spectral_scan": entry("k_spectral_scan", 8.0 * float(lpad) * npad, True),
       "k_zbuild_i8": entry("k_zbuild_i8", float(lpad) * npad + 6.0 * npad * npad + 8.0 * float(lpad) * npad, True)}
json.dump(out, open(os.path.join(ROOT, "profiles", "r04_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
