#!/usr/bin/env python3
"""Condense rocprofv3 csv output (tools/profile_gpu.sh) into the per-kernel summary kept under profiles/."""
import csv
import re
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
OURS = ("k_vara_i8p", "k_vara_i8w", "k_syrk_f4w", "k_syrk_f4p", "k_vara_i8", "k_vara_f64", "k_cert_lb", "k_cert_select", "k_cert_gather", "k_tiles_pack", "k_syrk_f4", "k_pack_fp4", "k_marker_shift", "k_rho_rows", "k_rho_colsum", "k_rho_cols", "k_rho_final", "k_syrk_i8", "k_gemm_f64", "k_gemv_mfma", "k_slice_vec", "k_sym_check", "k_vara_prep", "k_absmax_offdiag", "k_slice_w", "k_fold_upper", "k_colgemv", "k_tsq", "k_absmax",
        "k_transpose_pack_fp4", "k_transpose_i8", "k_spectral_scan", "k_spectral_finish", "k_zbuild_i8", "k_zbuild", "k_slice_u", "k_cert_bounds", "k_cert_lb_b", "k_cert_select_b", "k_mmt_finish", "k_mmt_normalise", "k_decode_ascii", "k_vara_i8_finish",
        "k_gram_rowabs_i8", "k_gram_hi_i8", "k_last_digit_sym", "k_spectral_decide", "k_ext_select", "k_ext_head", "k_ext_copy_slice", "k_ext_gather", "k_ext_apply",
        "k_w8_gemm_p", "k_w8_gemm", "k_w8_slice", "k_w8_rowstats", "k_w8_combine1", "k_w8_combine2", "k_w8_asymsq", "k_w8_reduce", "k_w8_rowgemv", "k_w8_mgemv",
        "k_w8_rowdot", "k_w8_sumdiag", "k_w8_rho_from_r", "k_w8_mean2", "k_w8_fill_ones", "k_w8_rows_f64")


def find(pattern):
    fs = glob.glob(os.path.join(out, pattern), recursive=True)
    return fs[0] if fs else None


def short(name):
    for k in OURS:
        if k in name:
            if "k_gemv_mfma" in name:   # per launch shape: the genotype pass of the digit-slice scan is <true, true> (a, diagonal term, m^T rho)
                m = re.search(r"k_gemv_mfma<([^>]*)>", name)
                tp = m.group(1).replace(" ", "") if m else ("true,true" if "Lb1ELb1E" in name else ("false,false" if "Lb0ELb0E" in name else ("true,false" if "Lb1ELb0E" in name else "?")))
                return "k_gemv_mfma<%s>" % tp
            if "k_w8_mgemv" in name:
                return "k_w8_mgemv_row" if "k_w8_mgemv_row" in name else ("k_w8_mgemv_part" if "k_w8_mgemv_part" in name else "k_w8_mgemv_sum")
            if "k_gemm_f64_dma_tail" in name:
                return "k_gemm_f64_dma_tail"
            if "k_gemm_f64_dma" in name:
                return "k_gemm_f64_dma"
            if "k_gemm_f64_list" in name:
                return "k_gemm_f64_list"
            if "k_gemm_f64_tail" in name:
                return "k_gemm_f64_tail"
            if "k_gemm_f64" in name:
                return "k_gemm_f64<i8A,rowdot>" if "Li1ELi1E" in name or "<1, 1>" in name else "k_gemm_f64<f64A,store>"
            if "k_vara_i8_finish" in name:
                return "k_vara_i8_finish"
            if "k_vara_i8p" in name and ("Lb0ELb1E" in name or "<false, true>" in name):
                return "k_vara_i8p<ext>"   # the launch of eagle_dev_vara_i8_extend (dropped on the device when nobody qualifies)
            if "k_vara_f64_sum" in name:
                return "k_vara_f64_sum"
            if "k_vara_f64d" in name:   # round 3's fp64 vara kernel (k_vara_f64 = round 2's, tune 28)
                return "k_vara_f64d<split>" if ("Lb1E" in name or "<true>" in name) else "k_vara_f64d"
            if "k_vara_f64" in name:
                return "k_vara_f64<split>" if ("Lb1E" in name or "<true>" in name) else "k_vara_f64"
            return k
    return None


print("# rocprofv3 summary of", os.path.basename(out))
f = find("stats/**/*kernel_stats.csv")
if f:
    print("\n## kernel stats (rocprofv3 --kernel-trace --stats), this repo's kernels")
    print("%-28s %8s %14s %14s %8s %12s %12s" % ("kernel", "calls", "total_ms", "avg_ms", "pct", "min_ms", "max_ms"))
    for r in csv.DictReader(open(f)):
        s = short(r["Name"])
        if s:
            print("%-28s %8s %14.3f %14.4f %8s %12.4f %12.4f" % (s, r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                                 float(r["AverageNs"]) / 1e6, r["Percentage"], float(r.get("MinNs", 0)) / 1e6,
                                                                 float(r.get("MaxNs", 0)) / 1e6))
for tag in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_tcc", "pmc_grbm"):
    f = find(tag + "/**/*counter_collection.csv")
    if not f:
        continue
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(f)):
        s = short(r["Kernel_Name"])
        if s:
            acc[s][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("\n## %s: mean counter value per dispatch" % tag)
    for k in sorted(acc):
        for c in sorted(acc[k]):
            v = acc[k][c]
            print("%-28s %-28s n=%-4d mean=%.6g" % (k, c, len(v), sum(v) / len(v)))
