#!/bin/bash
# Regenerates the measurements kept under profiles/r04/ (run on a GPU box from the repo root through gpurun; outputs land in
# gpurun_out/r04/).
#   bash tools/profile_r04.sh rhs_pmc    block-of-right-hand-sides solve at n = 65536: FETCH_SIZE / WRITE_SIZE / MFMA-busy passes
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04
mkdir -p $O
if [ "$1" = rhs_pmc ]; then
    for c in FETCH_SIZE WRITE_SIZE; do
        timeout -k 10 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_rhs_$c -o p -- python3 tools/rhs_speed.py --reps 1 ${2:-32768} > $O/pmc_rhs_$c.log 2> $O/pmc_rhs_$c.err || exit 1
    done
    python3 tools/pmc_sum.py $O/pmc_rhs_FETCH_SIZE/p_counter_collection.csv $O/pmc_rhs_WRITE_SIZE/p_counter_collection.csv > $O/pmc_rhs_fetch_write.txt
    C="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE"
    timeout -k 10 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc_rhs_mfma -o p -- python3 tools/rhs_speed.py --reps 1 ${2:-32768} > $O/pmc_rhs_mfma.log 2> $O/pmc_rhs_mfma.err || exit 1
    python3 tools/pmc_mfma_busy.py $O/pmc_rhs_mfma/p_counter_collection.csv > $O/pmc_rhs_mfma_busy.txt
fi
