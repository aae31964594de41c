#!/bin/bash
# Regenerates the measurements kept under profiles/r03/ (run on a GPU box from the repo root through gpurun; outputs land
# in gpurun_out/r03/, tools/collect_r03.sh copies the summaries worth keeping to profiles/r03/).
#   bash tools/profile_r03.sh main    bench lines + rocprofv3 kernel stats of the BASELINE configs, size sweeps
#   bash tools/profile_r03.sh mfma    MFMA-busy PMC pass at n = 131072 and n = 16384 (counters in their own runs)
#   bash tools/profile_r03.sh kmax    the k-per-launch cap experiment (SGPR_GEMM_KMAX)
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03
mkdir -p $O
stats() {   # stats <tag> <bench args...>: the bench line under rocprofv3 + the per-kernel summary
    tag=$1; shift
    timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -o p -- python3 bench.py "$@" > $O/bench_${tag}_under_rocprof.json 2> $O/prof_$tag.err &&
    cp $O/prof_$tag/p_kernel_stats.csv $O/bench_${tag}_kernel_stats.csv
}
if [ "$1" = main ]; then
    timeout -k 10 600 python3 bench.py > $O/bench_n131072.json 2> $O/bench_n131072.err &&
    stats n131072 --cpu-sample 0 &&
    SGPR_GEMM_KMAX=0 timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 > $O/bench_n131072_kmax0.json 2> $O/bench_n131072_kmax0.err &&
    timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 > $O/bench_n131072_kmax8192.json 2> $O/bench_n131072_kmax8192.err &&
    timeout -k 10 300 python3 bench.py --n-pts 8192 --steps 10 --warmup 3 > $O/bench_n16384.json 2> $O/bench_n16384.err &&
    stats n16384 --n-pts 8192 --steps 2 --warmup 1 --cpu-sample 0 &&
    timeout -k 10 600 python3 bench.py --d 2 --n-pts 32768 --family C --cpu-sample 2048 > $O/bench_henon_d2_n131072.json 2> $O/bench_henon.err &&
    stats henon_d2_n131072 --d 2 --n-pts 32768 --family C --steps 1 --warmup 0 --cpu-sample 0 &&
    timeout -k 10 600 python3 bench.py --d 3 --n-pts 16384 --cpu-sample 2048 > $O/bench_tokamak_d3_n98304.json 2> $O/bench_tok.err &&
    stats tokamak_d3_n98304 --d 3 --n-pts 16384 --steps 1 --warmup 0 --cpu-sample 0 &&
    timeout -k 10 300 python3 tools/potrf_modes.py 2048 4096 8192 16384 24576 32768 2>&1 | grep mode > $O/potrf_sizes.txt &&
    timeout -k 10 300 python3 tools/solve_speed.py 4096 8192 16384 32768 65536 2>&1 | grep solve > $O/solve_sizes.txt &&
    cp $O/prof_n131072/p_agent_info.csv $O/agent_info.csv
elif [ "$1" = rehearsal ]; then
    # NOT measurements: the N > 1 leg end to end on ONE card (all ranks on cuda:0, collectives over gloo)
    for g in 2 4; do
        SGPR_BENCH_ONE_CARD=1 timeout -k 10 400 python3 bench.py --gpus $g --n-pts 16384 --nb 2048 --steps 1 --warmup 1 --cpu-sample 0 > $O/rehearsal_one_card_gpus$g.json 2> $O/rehearsal_one_card_gpus$g.err || exit 1
    done
    for g in 3 4; do
        SGPR_BENCH_ONE_CARD=1 timeout -k 10 400 python3 bench.py --gpus $g --d 2 --family C --n-pts 6144 --nb 1024 --steps 1 --warmup 1 --cpu-sample 0 > $O/rehearsal_one_card_d2_gpus$g.json 2> $O/rehearsal_one_card_d2_gpus$g.err || exit 1
    done
elif [ "$1" = batch ]; then
    timeout -k 10 900 python3 tools/batch_rate.py 2>/dev/null > $O/batch_rate.md
elif [ "$1" = mfma ]; then
    C="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE"
    timeout -k 10 900 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc_mfma_n131072 -o p -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-launch-events > $O/pmc_mfma_n131072.json 2> $O/pmc_mfma_n131072.err &&
    python3 tools/pmc_mfma_busy.py $O/pmc_mfma_n131072/p_counter_collection.csv > $O/pmc_mfma_busy_n131072.txt &&
    timeout -k 10 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc_mfma_n16384 -o p -- python3 bench.py --n-pts 8192 --steps 1 --warmup 0 --cpu-sample 0 --no-launch-events > $O/pmc_mfma_n16384.json 2> $O/pmc_mfma_n16384.err &&
    python3 tools/pmc_mfma_busy.py $O/pmc_mfma_n16384/p_counter_collection.csv > $O/pmc_mfma_busy_n16384.txt
elif [ "$1" = pmc ]; then
    # HBM traffic of the headline configuration: separate --pmc FETCH_SIZE / WRITE_SIZE passes (counters in their own runs)
    timeout -k 10 300 python3 tools/gemm_launches.py 65536 --json $O/launches_n131072.json > $O/gemm_launches_n131072.txt 2>&1 || exit 1
    for c in FETCH_SIZE WRITE_SIZE; do
        timeout -k 10 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o p -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-launch-events > $O/pmc_$c.json 2> $O/pmc_$c.err || exit 1
    done
    python3 tools/pmc_sum.py $O/pmc_FETCH_SIZE/p_counter_collection.csv $O/pmc_WRITE_SIZE/p_counter_collection.csv > $O/pmc_fetch_write_summary_n131072.txt &&
    python3 tools/pmc_traffic_json.py $O/pmc_FETCH_SIZE/p_counter_collection.csv $O/pmc_WRITE_SIZE/p_counter_collection.csv $O/launches_n131072.json 65536 > $O/pmc_traffic_n131072.json
elif [ "$1" = kmax ]; then
    : > $O/kmax.txt
    for km in 0 8192 16384; do
        SGPR_GEMM_KMAX=$km timeout -k 10 300 python3 tools/probe_kmax.py 65536 65536 2 >> $O/kmax.txt 2>> $O/kmax.err || exit 1
        SGPR_GEMM_KMAX=$km timeout -k 10 300 python3 tools/probe_kmax.py 65536 32768 2 >> $O/kmax.txt 2>> $O/kmax.err || exit 1
    done
    for km in 0 16384; do
        SGPR_GEMM_KMAX=$km timeout -k 10 400 python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 > $O/bench_kmax$km.json 2> $O/bench_kmax$km.err || exit 1
    done
    for km in 0 16384; do
        SGPR_GEMM_KMAX=$km timeout -k 10 400 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_tcc_kmax$km -o p -- python3 tools/probe_kmax.py 65536 65536 1 > $O/pmc_tcc_kmax$km.log 2>&1 || exit 1
        python3 tools/pmc_sum.py $O/pmc_tcc_kmax$km/p_counter_collection.csv > $O/pmc_tcc_kmax$km.txt
    done
fi
