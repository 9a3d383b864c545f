#!/bin/bash
# Runs on the GPU box: counters of the vara kernel variants of tools/bench_vara.py (VARIANTS, N, LM, SLICES from the environment).
# Usage: tools/profile_vara_ab.sh <tag>      output: gpurun_out/prof_<tag>/summary.txt
set -o pipefail
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
KF="--kernel-include-regex k_vara_i8"
rocprofv3 $KF --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -o pmc -- python3 $ROOT/tools/bench_vara.py > $OUT/pmc_sq.log 2>&1 || { tail -5 $OUT/pmc_sq.log; exit 1; }
rocprofv3 $KF --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_grbm -o pmc -- python3 $ROOT/tools/bench_vara.py > $OUT/pmc_grbm.log 2>&1 || { tail -5 $OUT/pmc_grbm.log; exit 1; }
rocprofv3 $KF --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python3 $ROOT/tools/bench_vara.py > $OUT/pmc_fetch.log 2>&1 || { tail -5 $OUT/pmc_fetch.log; exit 1; }
rocprofv3 $KF --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/pmc_tcc -o pmc -- python3 $ROOT/tools/bench_vara.py > $OUT/pmc_tcc.log 2>&1 || { tail -5 $OUT/pmc_tcc.log; exit 1; }
cd $ROOT && python3 - $OUT > $OUT/summary.txt <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for tag in ("pmc_sq", "pmc_grbm", "pmc_fetch", "pmc_tcc"):
    for f in glob.glob(os.path.join(out, tag, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "finish" in k or "bound" in k: continue
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(os.path.join(out, tag, "**", "*kernel_trace.csv"), recursive=True):
        if tag != "pmc_grbm": continue
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "finish" in k or "bound" in k: continue
            dur[k].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e6)
for k in sorted(acc):
    c = {n: sum(v) / len(v) for n, v in acc[k].items()}
    d = sorted(dur[k])[len(dur[k]) // 2] if dur[k] else float("nan")
    print(k)
    print("  median duration under GRBM pass %.3f ms, dispatches %d" % (d, len(dur[k])))
    for n in sorted(c): print("  %-28s %.4g" % (n, c[n]))
    if "GRBM_GUI_ACTIVE" in c and d == d: print("  clock ~ %.2f GHz (GRBM_GUI_ACTIVE is summed over the 8 XCDs)" % (c["GRBM_GUI_ACTIVE"] / 8 / d / 1e6))
    if "SQ_BUSY_CYCLES" in c and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        print("  MFMA busy / SQ busy cycles (counter ratio) %.3f" % (c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["SQ_BUSY_CYCLES"]))
    if "GRBM_GUI_ACTIVE" in c and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        print("  MFMA busy fraction of GPU cycles %.3f (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 128))" % (c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] * 128)))
    if "SQ_WAIT_INST_ANY" in c and "SQ_WAVE_CYCLES" in c: print("  wait-inst / wave cycles %.3f" % (c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]))
    if "TCC_REQ_sum" in c: print("  L2 hit rate %.3f" % (c["TCC_HIT_sum"] / c["TCC_REQ_sum"]))
    if "FETCH_SIZE" in c: print("  fabric-side read bytes 2 x FETCH_SIZE x 1024 = %.1f GB per launch" % (2 * c["FETCH_SIZE"] * 1024 / 1e9))
PY
cat $OUT/summary.txt
find $OUT -name "*.db" -delete
