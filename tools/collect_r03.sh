#!/bin/bash
# copies the summaries of the tools/profile_r03.sh sessions from gpurun_out/r03/ into profiles/r03/ (repo root)
set -e
cd "$(dirname "$0")/.."
S=gpurun_out/r03; D=profiles/r03
mkdir -p $D
for f in bench_n131072 bench_n16384 bench_henon_d2_n131072 bench_tokamak_d3_n98304; do
    cp $S/${f}_kernel_stats.csv $D/
    for g in $f ${f}_under_rocprof; do grep '^{' $S/$g.json | tail -1 > $D/$g.json; done
done
for g in bench_n131072_kmax0 bench_n131072_kmax8192; do grep '^{' $S/$g.json | tail -1 > $D/$g.json; done
cp $S/potrf_sizes.txt $S/solve_sizes.txt $S/agent_info.csv $D/
cp $S/flow_devs.txt $D/flow_deviations.txt
python tools/check_frac.py $D > $D/frac_from_kernel_stats.md
[ -f $S/batch_rate.md ] && cp $S/batch_rate.md $D/
for f in launches_n131072.json pmc_fetch_write_summary_n131072.txt pmc_traffic_n131072.json; do [ -f $S/$f ] && cp $S/$f $D/; done
[ -f $S/gemm_launches_n131072.txt ] && grep -v amdgpu.ids $S/gemm_launches_n131072.txt > $D/gemm_launches_n131072.txt
for g in $S/rehearsal_one_card_*.json; do [ -f $g ] && grep '^{' $g | tail -1 > $D/$(basename $g); done
for n in n131072 n16384; do
    [ -f $S/pmc_mfma_busy_$n.txt ] && cp $S/pmc_mfma_busy_$n.txt $D/
done
{
    echo "# k-per-launch cap (SGPR_GEMM_KMAX), one MI355X per block of lines"
    echo
    echo "One lower-triangular C -= A A^T, m = 65536 (tools/probe_kmax.py; two timed repeats each):"
    echo '```'
    cat $S/kmax.txt
    echo '```'
    echo "L2 (TCC) hits / misses of the k = 65536 product, KMAX = 0 and 16384 (rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum):"
    echo '```'
    grep gemm_nt $S/pmc_tcc_kmax0.txt | sed 's/^/KMAX=0      /'
    grep gemm_nt $S/pmc_tcc_kmax16384.txt | sed 's/^/KMAX=16384  /'
    echo '```'
    echo "Whole bench step at n = 131072 (python3 bench.py --steps 2 --warmup 1 --cpu-sample 0), back to back on one box:"
    echo '```'
    for g in bench_kmax0 bench_kmax16384; do python3 - $S/$g.json $g <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("%-24s value %.2f TFLOP/s  %.1f ms/step  chol %.2f TFLOP/s  MFMA kernel all launches %.2f, alone %.2f" % (
    sys.argv[2], d["value"], d["ms_per_step"], d["chol_tflops"], d["roofline"]["achieved"], d["roofline"].get("achieved_alone") or 0))
PY
    done
    echo "(another box, another session:)"
    for g in bench_n131072_kmax0 bench_n131072_kmax8192; do python3 - $S/$g.json $g <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("%-24s value %.2f TFLOP/s  %.1f ms/step  chol %.2f TFLOP/s  MFMA kernel all launches %.2f, alone %.2f" % (
    sys.argv[2], d["value"], d["ms_per_step"], d["chol_tflops"], d["roofline"]["achieved"], d["roofline"].get("achieved_alone") or 0))
PY
    done
    echo '```'
} > $D/kmax.md
