#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel stats + separate PMC passes of tools/c4_shard.py at the n = 50,000 shape
# (the fractions of that shape rested on HIP events only).  Usage: tools/profile_c4.sh <tag> [n] [L]
set -o pipefail
TAG=$1; N=${2:-50000}; L=${3:-65536}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/tools/c4_shard.py $N $L 8"
KF="--kernel-include-regex k_vara_i8|k_w8_gemm|k_gemm_f64|k_syrk_f4|k_gram|k_w8_slice|k_w8_combine"
echo "== stats pass (W on the int8 engine)"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- $CMD > $OUT/stats.log 2>&1 || { tail -5 $OUT/stats.log; exit 1; }
echo "== stats pass (W on the fp64 GEMM)"
W_MODE=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_w64 -o stats -- $CMD > $OUT/stats_w64.log 2>&1 || { tail -5 $OUT/stats_w64.log; exit 1; }
for P in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY" "tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "grbm GRBM_GUI_ACTIVE"; do
  set -- $P; T=$1; shift
  echo "== pmc $T"
  rocprofv3 $KF --kernel-trace --pmc $@ --output-format csv -d $OUT/pmc_$T -o pmc -- $CMD > $OUT/pmc_$T.log 2>&1 || { tail -5 $OUT/pmc_$T.log; }
done
cd $ROOT && python3 tools/summarise_prof.py $OUT > $OUT/summary.txt 2>&1
f=$(find $OUT/stats_w64 -name "*kernel_stats.csv" | head -1)
echo "## W on the fp64 GEMM (W_MODE=0): kernel stats" >> $OUT/summary.txt
python3 - "$f" >> $OUT/summary.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r["Name"] for k in ("k_gemm_f64", "k_vara_i8p", "k_syrk_f4w")):
        print("%-60s calls %s avg %.3f ms min %.3f max %.3f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6))
PY
cat $OUT/summary.txt | head -60
find $OUT -name "*_kernel_trace.csv" -size +20M -delete
find $OUT -name "*.db" -delete
du -sh $OUT
