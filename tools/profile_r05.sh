#!/bin/bash
# Regenerates the measurements kept under profiles/r05/ (run on a GPU box from the repo root through gpurun; outputs land in
# gpurun_out/r05/, tools/collect_r05.sh copies the summaries worth keeping to profiles/r05/ and REFUSES a counter pass whose
# kernel hash is not the tree's).
#   bash tools/profile_r05.sh main      bench lines + rocprofv3 kernel stats of the four single-GPU BASELINE configs
#   bash tools/profile_r05.sh pmc       HBM traffic of the headline configuration: separate --pmc FETCH_SIZE / WRITE_SIZE passes
#   bash tools/profile_r05.sh pmc_tok   the same for config 05_tokamak (d = 3, n = 98304) incl. its block solve
#   bash tools/profile_r05.sh side      factor / solve size sweeps (queue vs look-ahead A/B), batch lines, map rate, block solve sizes,
#                                       latency census of the leaf's building blocks, chain stamps
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05
mkdir -p $O
stats() {   # stats <tag> <command...>: the command under rocprofv3 + the per-kernel summary
    tag=$1; shift
    timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -o p -- "$@" > $O/${tag}_under_rocprof.json 2> $O/prof_$tag.err &&
    cp $O/prof_$tag/p_kernel_stats.csv $O/${tag}_kernel_stats.csv
}
if [ "$1" = main ]; then
    timeout -k 10 600 python3 bench.py --steps 3 --warmup 1 > $O/bench_n131072.json 2> $O/bench_n131072.err &&
    stats bench_n131072 python3 bench.py --cpu-sample 0 --cond-iters 0 &&
    timeout -k 10 300 python3 bench.py --n-pts 8192 --steps 10 --warmup 3 > $O/bench_n16384.json 2> $O/bench_n16384.err &&
    stats bench_n16384 python3 bench.py --n-pts 8192 --steps 3 --warmup 1 --cpu-sample 0 --cond-iters 0 &&
    timeout -k 10 600 python3 bench.py --d 3 --n-pts 16384 --steps 3 --warmup 1 --cpu-sample 2048 > $O/bench_tokamak_d3_n98304.json 2> $O/bench_tok.err &&
    stats bench_tokamak_d3_n98304 python3 bench.py --d 3 --n-pts 16384 --steps 2 --warmup 1 --cpu-sample 0 --cond-iters 0 &&
    timeout -k 10 600 python3 bench.py --d 2 --n-pts 32768 --family C --steps 3 --warmup 1 --cpu-sample 2048 > $O/bench_henon_d2_n131072.json 2> $O/bench_henon.err &&
    stats bench_henon_d2_n131072 python3 bench.py --d 2 --n-pts 32768 --family C --steps 2 --warmup 1 --cpu-sample 0 --cond-iters 0 &&
    cp $O/prof_bench_n131072/p_agent_info.csv $O/agent_info.csv
elif [ "$1" = pmc ]; then
    timeout -k 10 300 python3 tools/gemm_launches.py 65536 --json $O/launches_n131072.json > $O/gemm_launches_n131072.txt 2>&1 || exit 1
    for c in FETCH_SIZE WRITE_SIZE; do
        timeout -k 10 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o p -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 --cond-iters 0 --no-launch-events > $O/pmc_$c.json 2> $O/pmc_$c.err || exit 1
    done
    python3 tools/pmc_sum.py $O/pmc_FETCH_SIZE/p_counter_collection.csv $O/pmc_WRITE_SIZE/p_counter_collection.csv > $O/pmc_fetch_write_summary_n131072.txt &&
    python3 tools/pmc_traffic_json.py $O/pmc_FETCH_SIZE/p_counter_collection.csv $O/pmc_WRITE_SIZE/p_counter_collection.csv $O/launches_n131072.json 65536 > $O/pmc_traffic_n131072.json
elif [ "$1" = side ]; then
    timeout -k 10 300 python3 tools/potrf_modes.py 512 1024 2048 3072 4096 5120 6144 6656 7168 8192 10240 12288 14336 16384 2>&1 | grep mode > $O/potrf_sizes.txt
    SGPR_POTRF_Q=0 timeout -k 10 300 python3 tools/potrf_modes.py 6656 7168 8192 10240 12288 14336 2>&1 | grep mode | sed 's/^/SGPR_POTRF_Q=0: /' >> $O/potrf_sizes.txt
    { echo "# same box, same session: factor stage (ms) of the task-queue driver and of the look-ahead driver (SGPR_POTRF_Q=0), and the panel"; echo "# kernel's round-5 pieces switched off one by one (tools/potrf_modes.py name=value ...)";
      timeout -k 10 300 python3 tools/potrf_modes.py 8192 2>&1 | grep mode | sed 's/^/queue (default): /';
      SGPR_POTRF_Q=0 timeout -k 10 300 python3 tools/potrf_modes.py 8192 2>&1 | grep mode | sed 's/^/look-ahead: /';
      timeout -k 10 300 python3 tools/potrf_modes.py 1024 2048 4096 6144 2>&1 | grep mode | sed 's/^/round 5 (default): /';
      timeout -k 10 300 python3 tools/potrf_modes.py panel_tiles=0 1024 2048 4096 6144 2>&1 | grep mode;
      timeout -k 10 300 python3 tools/potrf_modes.py panel_tiles=0 panel_helpers=0 1024 2048 4096 6144 2>&1 | grep mode;
      timeout -k 10 300 python3 tools/potrf_modes.py panel_tiles=0 panel_helpers=0 panel_below_early=0 1024 2048 4096 6144 2>&1 | grep mode; } > $O/potrf_q_vs_la.log
    timeout -k 10 300 python3 tools/solve_speed.py 4096 8192 16384 32768 65536 2>&1 | grep solve > $O/solve_sizes.txt
    timeout -k 10 300 python3 tools/rhs_speed.py --reps 3 8192 16384 32768 2>&1 | grep nrhs > $O/rhs_sizes.txt
    timeout -k 10 300 python3 tools/rhs_speed.py --reps 3 --d 3 16384 2>&1 | grep nrhs >> $O/rhs_sizes.txt
    timeout -k 10 300 python3 tools/rhs_speed.py --reps 3 --nrhs 32 --d 3 16384 2>&1 | grep nrhs >> $O/rhs_sizes.txt
    for b in 80,1024 160,1024 512,64 1024,64 2048,64; do
        timeout -k 10 200 python3 bench.py --batch $b --steps 7 --warmup 2 > $O/bench_batch_${b/,/x}.json 2> $O/bench_batch_${b/,/x}.err || exit 1
    done
    timeout -k 10 400 python3 tools/map_rate.py > $O/map_rate.md 2> $O/map_rate.err
    python3 tools/probe_leaf.py 2>&1 | grep -v amdgpu.ids > $O/probe_leaf.txt
    python3 tools/probe_lat.py 2>&1 | grep -v amdgpu.ids > $O/probe_lat.txt
    python3 tools/probe_gemm_k.py 2>&1 | grep -v amdgpu.ids > $O/gemm_k.txt
fi
