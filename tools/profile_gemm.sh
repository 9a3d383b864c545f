#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel stats + separate PMC passes of tools/bench_gemm.py (W = S V S at N individuals).
# Usage: tools/profile_gemm.sh <tag> [N] [VARIANTS]      outputs under gpurun_out/prof_<tag>/
set -o pipefail
TAG=$1; export N=${2:-10000}; export VARIANTS=${3:-0}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
KF="--kernel-include-regex k_gemm_f64|k_sym_check|k_fold_upper|k_colgemv"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $ROOT/tools/bench_gemm.py > $OUT/stats.log 2>&1 || { tail -5 $OUT/stats.log; exit 1; }
rocprofv3 $KF --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python3 $ROOT/tools/bench_gemm.py > $OUT/pmc_fetch.log 2>&1 || { tail -5 $OUT/pmc_fetch.log; exit 1; }
rocprofv3 $KF --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- python3 $ROOT/tools/bench_gemm.py > $OUT/pmc_write.log 2>&1 || { tail -5 $OUT/pmc_write.log; exit 1; }
rocprofv3 $KF --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -o pmc -- python3 $ROOT/tools/bench_gemm.py > $OUT/pmc_sq.log 2>&1 || { tail -5 $OUT/pmc_sq.log; }
rocprofv3 $KF --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/pmc_tcc -o pmc -- python3 $ROOT/tools/bench_gemm.py > $OUT/pmc_tcc.log 2>&1 || { tail -5 $OUT/pmc_tcc.log; }
rocprofv3 $KF --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_grbm -o pmc -- python3 $ROOT/tools/bench_gemm.py > $OUT/pmc_grbm.log 2>&1 || { tail -5 $OUT/pmc_grbm.log; }
cd $ROOT && python3 tools/summarise_prof.py $OUT > $OUT/summary.txt 2>&1; cat $OUT/summary.txt
find $OUT -name "*_kernel_trace.csv" -size +20M -delete
find $OUT -name "*.db" -delete
du -sh $OUT
