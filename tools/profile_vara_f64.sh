#!/bin/bash
# Runs on the GPU box: A/B of the fp64 vara kernel (tools/bench_vara_f64.py) + LDS counters of the default variant.
# Usage: tools/profile_vara_f64.sh <tag>     output: gpurun_out/prof_<tag>/
set -o pipefail
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/tools/bench_vara_f64.py > $OUT/ab.txt 2>&1 || { tail -5 $OUT/ab.txt; exit 1; }
VARIANTS=0 rocprofv3 --kernel-include-regex k_vara_f64 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_sq -o pmc -- python3 $ROOT/tools/bench_vara_f64.py > $OUT/pmc_sq.log 2>&1 || { tail -5 $OUT/pmc_sq.log; exit 1; }
cd $ROOT && python3 - $OUT >> $OUT/ab.txt <<'PY'
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "pmc_sq", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_vara_f64d" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
c = {k: sum(v) / len(v) for k, v in acc.items()}
for k in sorted(c): print("  %-28s %.4g" % (k, c[k]))
if "SQ_LDS_IDX_ACTIVE" in c: print("  bank-conflict cycles / LDS active cycles %.3f" % (c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]))
PY
cat $OUT/ab.txt
find $OUT -name "*.db" -delete
