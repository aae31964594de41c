#!/bin/bash
# copies the summaries of the tools/profile_r05.sh sessions from gpurun_out/r05/ into profiles/r05/ (repo root).
# A counter pass (pmc_traffic_*.json) is only accepted when its code_hash is the hash of the kernel sources in THIS tree
# (bench.py: kernel_code_hash): bench.py reports roofline.traffic from it and from nothing else.
cd "$(dirname "$0")/.."
S=gpurun_out/r05; D=profiles/r05
mkdir -p $D
rc=0
HASH=$(python3 -c "import bench; print(bench.kernel_code_hash())")
for f in $S/pmc_traffic_*.json; do
    [ -f $f ] || continue
    h=$(python3 -c "import json,sys; print(json.load(open('$f')).get('code_hash'))")
    if [ "$h" != "$HASH" ]; then
        echo "collect_r05: $f was taken with kernels $h, the tree is $HASH -- NOT copied: re-run 'bash tools/profile_r05.sh pmc' on this tree" >&2
        rc=1
    else
        cp $f $D/
    fi
done
for f in bench_n131072 bench_n16384 bench_henon_d2_n131072 bench_tokamak_d3_n98304; do
    [ -f $S/${f}_kernel_stats.csv ] && cp $S/${f}_kernel_stats.csv $D/
    for g in $f ${f}_under_rocprof; do [ -f $S/$g.json ] && grep '^{' $S/$g.json | tail -1 > $D/$g.json; done
done
for f in panel_stress.txt rhs_sizes.txt potrf_sizes.txt solve_sizes.txt agent_info.csv map_rate.md potrf_q_vs_la.log probe_leaf.txt probe_lat.txt gemm_k.txt \
         pmc_fetch_write_summary_n131072.txt launches_n131072.json; do [ -f $S/$f ] && grep -v amdgpu.ids $S/$f > $D/$f; done
[ -f $S/gemm_launches_n131072.txt ] && grep -v amdgpu.ids $S/gemm_launches_n131072.txt > $D/gemm_launches_n131072.txt
for g in $S/bench_batch_*.json; do [ -f $g ] && grep '^{' $g | tail -1 > $D/$(basename $g); done
[ -f $D/bench_n131072_kernel_stats.csv ] && python3 tools/check_frac.py $D > $D/frac_from_kernel_stats.md 2>/dev/null
ls $D
exit $rc
