#!/bin/bash
# Regenerates the measurements kept under profiles/r04/ (run on a GPU box from the repo root through gpurun; outputs land in
# gpurun_out/r04/, tools/collect_r04.sh copies the summaries worth keeping to profiles/r04/).
#   bash tools/profile_r04.sh main       bench lines + rocprofv3 kernel stats of the BASELINE configs (config 05 with its block-of-64 solve)
#   bash tools/profile_r04.sh rhs        block-of-right-hand-sides solve: sizes, the round-3 recursion beside it, kernel stats
#   bash tools/profile_r04.sh rhs_pmc N  FETCH_SIZE / WRITE_SIZE / MFMA-busy passes of the block solve at N points (counters in their own runs)
#   bash tools/profile_r04.sh side       map rate, batch lines, factor / solve size sweeps
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04
mkdir -p $O
stats() {   # stats <tag> <command...>: the command under rocprofv3 + the per-kernel summary
    tag=$1; shift
    timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -o p -- "$@" > $O/${tag}_under_rocprof.json 2> $O/prof_$tag.err &&
    cp $O/prof_$tag/p_kernel_stats.csv $O/${tag}_kernel_stats.csv
}
if [ "$1" = main ]; then
    timeout -k 10 600 python3 bench.py > $O/bench_n131072.json 2> $O/bench_n131072.err &&
    stats bench_n131072 python3 bench.py --cpu-sample 0 &&
    timeout -k 10 300 python3 bench.py --n-pts 8192 --steps 10 --warmup 3 > $O/bench_n16384.json 2> $O/bench_n16384.err &&
    stats bench_n16384 python3 bench.py --n-pts 8192 --steps 2 --warmup 1 --cpu-sample 0 &&
    timeout -k 10 600 python3 bench.py --d 3 --n-pts 16384 --cpu-sample 2048 > $O/bench_tokamak_d3_n98304.json 2> $O/bench_tok.err &&
    stats bench_tokamak_d3_n98304 python3 bench.py --d 3 --n-pts 16384 --steps 1 --warmup 0 --cpu-sample 0 &&
    timeout -k 10 600 python3 bench.py --d 2 --n-pts 32768 --family C --cpu-sample 2048 > $O/bench_henon_d2_n131072.json 2> $O/bench_henon.err &&
    stats bench_henon_d2_n131072 python3 bench.py --d 2 --n-pts 32768 --family C --steps 1 --warmup 0 --cpu-sample 0 &&
    cp $O/prof_bench_n131072/p_agent_info.csv $O/agent_info.csv
elif [ "$1" = rhs ]; then
    timeout -k 10 300 python3 tools/rhs_speed.py --reps 3 1024 2048 4096 8192 16384 32768 2>&1 | grep nrhs > $O/rhs_sizes.txt &&
    timeout -k 10 300 python3 tools/rhs_speed.py --reps 3 --d 3 16384 2>&1 | grep nrhs >> $O/rhs_sizes.txt &&
    timeout -k 10 300 python3 tools/rhs_speed.py --reps 3 --nrhs 8 8192 32768 2>&1 | grep nrhs >> $O/rhs_sizes.txt &&
    timeout -k 10 300 python3 tools/rhs_speed.py --reps 3 --nrhs 16 8192 32768 2>&1 | grep nrhs >> $O/rhs_sizes.txt &&
    timeout -k 10 300 python3 tools/rhs_speed.py --reps 3 --nrhs 32 8192 32768 2>&1 | grep nrhs >> $O/rhs_sizes.txt &&
    timeout -k 10 300 python3 tools/rhs_speed.py --reps 3 --nrhs 32 --d 3 16384 2>&1 | grep nrhs >> $O/rhs_sizes.txt &&
    timeout -k 10 300 python3 tools/rhs_speed.py --reps 3 --nrhs 256 8192 2>&1 | grep nrhs >> $O/rhs_sizes.txt &&
    SGPR_TRSM=rec timeout -k 10 300 python3 tools/rhs_speed.py --reps 2 8192 2>&1 | grep nrhs | sed 's/^/SGPR_TRSM=rec (round 3: recursion over the GEMM kernel): /' >> $O/rhs_sizes.txt &&
    SGPR_TRSM=rec timeout -k 10 300 python3 tools/rhs_speed.py --reps 2 --d 3 16384 2>&1 | grep nrhs | sed 's/^/SGPR_TRSM=rec (round 3: recursion over the GEMM kernel): /' >> $O/rhs_sizes.txt &&
    stats rhs_n98304 python3 tools/rhs_speed.py --reps 3 --d 3 16384
elif [ "$1" = rhs_pmc ]; then
    N=${2:-32768}
    for c in FETCH_SIZE WRITE_SIZE; do
        timeout -k 10 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_rhs_$c -o p -- python3 tools/rhs_speed.py --reps 1 ${3:-} $N > $O/pmc_rhs_$c.log 2> $O/pmc_rhs_$c.err || exit 1
    done
    python3 tools/pmc_sum.py $O/pmc_rhs_FETCH_SIZE/p_counter_collection.csv $O/pmc_rhs_WRITE_SIZE/p_counter_collection.csv > $O/pmc_rhs_fetch_write.txt
    C="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE"
    timeout -k 10 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc_rhs_mfma -o p -- python3 tools/rhs_speed.py --reps 1 ${3:-} $N > $O/pmc_rhs_mfma.log 2> $O/pmc_rhs_mfma.err || exit 1
    python3 tools/pmc_mfma_busy.py $O/pmc_rhs_mfma/p_counter_collection.csv > $O/pmc_rhs_mfma_busy.txt
elif [ "$1" = side ]; then
    timeout -k 10 400 python3 tools/map_rate.py > $O/map_rate.md 2> $O/map_rate.err
    for b in 80,1024 160,1024 512,64 1024,64 2048,64; do
        timeout -k 10 200 python3 bench.py --batch $b --steps 7 --warmup 2 > $O/bench_batch_${b/,/x}.json 2> $O/bench_batch_${b/,/x}.err || exit 1
    done
    timeout -k 10 300 python3 tools/potrf_modes.py 1024 2048 4096 6656 7168 8192 10240 12288 14336 16384 2>&1 | grep mode > $O/potrf_sizes.txt
    SGPR_POTRF_Q=0 timeout -k 10 300 python3 tools/potrf_modes.py 6656 7168 8192 10240 12288 14336 2>&1 | grep mode | sed 's/^/SGPR_POTRF_Q=0: /' >> $O/potrf_sizes.txt
    timeout -k 10 300 python3 tools/solve_speed.py 4096 8192 16384 32768 65536 2>&1 | grep solve > $O/solve_sizes.txt
fi
