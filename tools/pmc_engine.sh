#!/bin/bash
# L2 hit/miss and fetch size of the SYRK micro-benchmark (tools/bench_i8_engine.py) under rocprofv3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_engine
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export VARIANTS=${VARIANTS:-0}
rocprofv3 --kernel-include-regex "k_syrk_i8" --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/tcc -o pmc -- python3 $ROOT/tools/bench_i8_engine.py > $OUT/tcc.log 2>&1 || tail -5 $OUT/tcc.log
rocprofv3 --kernel-include-regex "k_syrk_i8" --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o pmc -- python3 $ROOT/tools/bench_i8_engine.py > $OUT/fetch.log 2>&1 || tail -5 $OUT/fetch.log
cd $ROOT
python3 - <<PY
import csv, glob, collections
for tag in ("tcc","fetch"):
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True)
    if not f: print(tag, "no output"); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])): acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items(): print(tag, k, "n=%d mean=%.4g" % (len(v), sum(v)/len(v)))
PY
