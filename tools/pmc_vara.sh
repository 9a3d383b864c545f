#!/bin/bash
# L2 hit/miss + fetch size of k_vara_i8 inside bench.py under rocprofv3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_vara
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --cpu-sample 0 --steps 2 --warmup 1 --simple-operands --mode i8 $@"
rocprofv3 --kernel-include-regex "k_vara_i8" --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/tcc -o pmc -- python3 $ARGS > $OUT/tcc.log 2>&1 || tail -5 $OUT/tcc.log
rocprofv3 --kernel-include-regex "k_vara_i8" --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o pmc -- python3 $ARGS > $OUT/fetch.log 2>&1 || tail -5 $OUT/fetch.log
cd $ROOT
python3 - <<PY
import csv, glob, collections
for tag in ("tcc","fetch"):
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True)
    if not f: print(tag, "no output"); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "finish" in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items(): print(tag, k, "n=%d mean=%.4g" % (len(v), sum(v)/len(v)))
PY
