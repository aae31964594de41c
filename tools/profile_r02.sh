#!/bin/bash
# Regenerates the measurements kept under profiles/r02/ (run on a GPU box from the repo root through
# gpurun; outputs land in gpurun_out/r02/, the summaries worth keeping are copied to profiles/r02/ by hand).
#   bash tools/profile_r02.sh main     bench lines + rocprofv3 kernel stats of the BASELINE configs, size sweeps
#   bash tools/profile_r02.sh pmc      HBM traffic: separate --pmc FETCH_SIZE / WRITE_SIZE passes + L2 hit counters
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02
mkdir -p $O
stats() {   # stats <tag> <bench args...>: the bench line under rocprofv3 + the per-kernel summary
    tag=$1; shift
    timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -o p -- python3 bench.py "$@" > $O/bench_${tag}_under_rocprof.json 2> $O/prof_$tag.err &&
    cp $O/prof_$tag/p_kernel_stats.csv $O/bench_${tag}_kernel_stats.csv
}
if [ "$1" = main ]; then
    timeout -k 10 600 python3 bench.py > $O/bench_n131072.json 2> $O/bench_n131072.err &&
    stats n131072 --cpu-sample 0 &&
    timeout -k 10 300 python3 bench.py --n-pts 8192 --steps 10 --warmup 3 > $O/bench_n16384.json 2> $O/bench_n16384.err &&
    stats n16384 --n-pts 8192 --steps 2 --warmup 1 --cpu-sample 0 &&
    timeout -k 10 600 python3 bench.py --d 2 --n-pts 32768 --family C --cpu-sample 2048 > $O/bench_henon_d2_n131072.json 2> $O/bench_henon.err &&
    stats henon_d2_n131072 --d 2 --n-pts 32768 --family C --steps 1 --warmup 0 --cpu-sample 0 &&
    timeout -k 10 600 python3 bench.py --d 3 --n-pts 16384 --cpu-sample 2048 > $O/bench_tokamak_d3_n98304.json 2> $O/bench_tok.err &&
    stats tokamak_d3_n98304 --d 3 --n-pts 16384 --steps 1 --warmup 0 --cpu-sample 0 &&
    timeout -k 10 300 python3 tools/gemm_launches.py 65536 --json $O/launches_n131072.json > $O/gemm_launches_n131072.txt 2>&1 &&
    timeout -k 10 300 python3 tools/potrf_modes.py 2048 4096 8192 16384 24576 32768 2>&1 | grep mode > $O/potrf_sizes.txt &&
    timeout -k 10 300 python3 tools/solve_speed.py 4096 8192 16384 32768 65536 2>&1 | grep solve > $O/solve_sizes.txt &&
    timeout -k 10 300 python3 tools/batch_rate.py 2>/dev/null > $O/batch_rate.md &&
    cp $O/prof_n131072/p_agent_info.csv $O/agent_info.csv
elif [ "$1" = pmc ]; then
    for c in FETCH_SIZE WRITE_SIZE; do
        timeout -k 10 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o p -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-launch-events > $O/pmc_$c.json 2> $O/pmc_$c.err || exit 1
    done
    python3 tools/pmc_sum.py $O/pmc_FETCH_SIZE/p_counter_collection.csv $O/pmc_WRITE_SIZE/p_counter_collection.csv > $O/pmc_fetch_write_summary_n131072.txt &&
    python3 tools/pmc_traffic_json.py $O/pmc_FETCH_SIZE/p_counter_collection.csv $O/pmc_WRITE_SIZE/p_counter_collection.csv profiles/r02/launches_n131072.json 65536 > $O/pmc_traffic_n131072.json &&
    timeout -k 10 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_tcc -o p -- python3 tools/probe_gemm.py 32768 32768 16384 1 32768 32768 1024 1 > $O/pmc_tcc.log 2>&1 &&
    python3 tools/pmc_sum.py $O/pmc_tcc/p_counter_collection.csv > $O/pmc_tcc_summary.txt
elif [ "$1" = pmc_events ]; then
    # ONE run of a --pmc pass WITH the per-launch HIP events (round 1's crash): events now come from a fixed pool
    timeout -k 10 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_events -o p -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 > $O/pmc_events.json 2> $O/pmc_events.err
    echo "exit code $?" >> $O/pmc_events.err
    tail -5 $O/pmc_events.err
fi
