#!/bin/bash
# copies the summaries of the tools/profile_r04.sh sessions from gpurun_out/r04/ into profiles/r04/ (repo root)
cd "$(dirname "$0")/.."
S=gpurun_out/r04; D=profiles/r04
mkdir -p $D
for f in bench_n131072 bench_n16384 bench_henon_d2_n131072 bench_tokamak_d3_n98304; do
    [ -f $S/${f}_kernel_stats.csv ] && cp $S/${f}_kernel_stats.csv $D/
    for g in $f ${f}_under_rocprof; do [ -f $S/$g.json ] && grep '^{' $S/$g.json | tail -1 > $D/$g.json; done
done
[ -f $S/rhs_n98304_kernel_stats.csv ] && cp $S/rhs_n98304_kernel_stats.csv $D/
[ -f $S/rhs_n98304_under_rocprof.json ] && grep nrhs $S/rhs_n98304_under_rocprof.json > $D/rhs_n98304_under_rocprof.txt
for f in rhs_sizes.txt potrf_sizes.txt solve_sizes.txt agent_info.csv map_rate.md pmc_rhs_fetch_write.txt pmc_rhs_mfma_busy.txt potrf_q_vs_la.log potrf_park.log; do [ -f $S/$f ] && grep -v amdgpu.ids $S/$f > $D/$f; done
for g in $S/bench_batch_*.json; do [ -f $g ] && grep '^{' $g | tail -1 > $D/$(basename $g); done
for f in pmc_rhs_FETCH_SIZE pmc_rhs_mfma; do [ -f $S/$f.log ] && grep nrhs $S/$f.log > $D/${f}_run.txt; done
ls $D
