#!/bin/bash
# The four structured-population cases of profiles/r0N_structure_diag.txt (tools/diag_structure.py), one after the other.
# Usage (through gpurun): tools/structure_panel.sh > gpurun_out/structure_panel.txt
set -o pipefail
for c in "3 0.1 0.05 4096 262144" "8 0.3 0.005 4096 262144" "2 0.5 0.01 10000 500000" "20 0.05 0.002 2000 200000"; do
    set -- $c
    echo "## K=$1 FST=$2 PMIN=$3 N=$4 LM=$5"
    K=$1 FST=$2 PMIN=$3 N=$4 LM=$5 timeout -k 10 400 python3 tools/diag_structure.py 2>&1 | grep -v "Warning\|amdgpu.ids\|warnings.warn" || exit 1
done
