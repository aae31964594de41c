#!/bin/bash
# copies the summaries of a tools/profile_r02.sh session from gpurun_out/r02/ into profiles/r02/ (repo root)
set -e
cd "$(dirname "$0")/.."
S=gpurun_out/r02; D=profiles/r02
for f in bench_n131072 bench_n16384 bench_henon_d2_n131072 bench_tokamak_d3_n98304; do
    cp $S/${f}_kernel_stats.csv $D/
    for g in $f ${f}_under_rocprof; do grep '^{' $S/$g.json | tail -1 > $D/$g.json; done
done
cp $S/launches_n131072.json $S/potrf_sizes.txt $S/solve_sizes.txt $S/batch_rate.md $S/agent_info.csv $S/pmc_fetch_write_summary_n131072.txt $D/
grep -v amdgpu.ids $S/gemm_launches_n131072.txt > $D/gemm_launches_n131072.txt
cp $S/pmc_tcc_summary.txt $D/ && grep TFLOP $S/pmc_tcc.log >> $D/pmc_tcc_summary.txt
python tools/pmc_traffic_json.py $S/pmc_FETCH_SIZE/p_counter_collection.csv $S/pmc_WRITE_SIZE/p_counter_collection.csv $D/launches_n131072.json 65536 > $D/pmc_traffic_n131072.json
