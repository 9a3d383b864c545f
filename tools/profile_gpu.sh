#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel stats + separate PMC passes of bench.py (headline shape, 10k x 1M).
# PMC passes reload the operands the stats pass computed: torch.linalg.eigh (rocSOLVER) segfaults under rocprofv3
# counter collection (log: profiles/r02_eigh_under_pmc.log, made by tools/eigh_under_pmc.py), and the workload must be the
# same in every pass.
# Usage: tools/profile_gpu.sh <tag> [bench args...]      outputs under gpurun_out/prof_<tag>/
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
OPS=/tmp/eagle_bench_operands.pt
# SECONDARY=1: the stats pass and the FETCH_SIZE / SQ / GRBM passes also run bench.py's secondary entries (fp64-mode scan, spectral
# scan, Z builds), so that k_vara_f64, k_spectral_scan and k_zbuild_i8 appear in the summary
if [ -n "$SECONDARY" ]; then NOSEC="--no-e2e"; else NOSEC="--no-secondary"; fi
ARGS="$ROOT/bench.py --cpu-sample 0 $NOSEC --steps 3 --warmup 1 --load-operands $OPS $@"
KF="--kernel-include-regex k_vara_i8|k_vara_f64|k_syrk_f4|k_gemm_f64|k_gemv|k_mmt_finish|k_slice_w|k_pack_fp4|k_transpose_pack_fp4|k_cert|k_spectral|k_zbuild|k_gram_rowabs|k_gram_hi|k_last_digit|k_w8_"
echo "== stats pass";
# the stats pass profiles the bench command itself (model-algebra operands; the CPU sample and the secondary entries are skipped)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $ROOT/bench.py --cpu-sample 0 $NOSEC --save-operands $OPS "$@" > $OUT/stats.log 2>&1 || { tail -5 $OUT/stats.log; exit 1; }
echo "== pmc FETCH_SIZE"
rocprofv3 $KF --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1 || { tail -5 $OUT/pmc_fetch.log; exit 1; }
echo "== pmc WRITE_SIZE"
rocprofv3 $KF --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- python3 $ARGS > $OUT/pmc_write.log 2>&1 || { tail -5 $OUT/pmc_write.log; exit 1; }
echo "== pmc SQ"
rocprofv3 $KF --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -o pmc -- python3 $ARGS > $OUT/pmc_sq.log 2>&1 || { tail -5 $OUT/pmc_sq.log; }
echo "== pmc TCC"
rocprofv3 $KF --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/pmc_tcc -o pmc -- python3 $ARGS > $OUT/pmc_tcc.log 2>&1 || { tail -5 $OUT/pmc_tcc.log; }
echo "== pmc GRBM"
rocprofv3 $KF --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_grbm -o pmc -- python3 $ARGS > $OUT/pmc_grbm.log 2>&1 || { tail -5 $OUT/pmc_grbm.log; }
cd $ROOT && python3 tools/summarise_prof.py $OUT > $OUT/summary.txt 2>&1; cat $OUT/summary.txt
# keep the merge-back small: drop the per-dispatch traces of torch's setup kernels
find $OUT -name "*_kernel_trace.csv" -size +20M -delete
find $OUT -name "*.db" -delete
du -sh $OUT
