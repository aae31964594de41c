"""roofline.achieved of a bench line recomputed from the rocprofv3 kernel stats of the SAME run:
    python tools/check_frac.py profiles/r03 [tags...]
reads bench_<tag>_under_rocprof.json and bench_<tag>_kernel_stats.csv; the two figures must agree to ~1 % (the bench line's
is flop / sum of HIP-event durations over one extra step, the CSV's is flop / sum of profiler durations over all steps)."""
import csv
import json
import os
import sys

d0 = sys.argv[1]
tags = sys.argv[2:] or ["n131072", "n16384", "henon_d2_n131072", "tokamak_d3_n98304"]
print("| config | launches per step | steps in the trace | achieved, bench line (TFLOP/s) | achieved, kernel stats | ratio | avg launch ms: line / stats |")
print("|---|---|---|---|---|---|---|")
for tag in tags:
    d = json.loads([l for l in open(os.path.join(d0, "bench_%s_under_rocprof.json" % tag)) if l.startswith("{")][-1])
    r = d["roofline"]
    rows = list(csv.DictReader(open(os.path.join(d0, "bench_%s_kernel_stats.csv" % tag))))
    if r["kernel"].startswith("chol_queue_kernel"):
        # the persistent worker kernel: its first instance spans the factor stage, the instances behind it find nothing to do
        #  -- total duration over the steps of the trace against n^3/3 per step
        g = [x for x in rows if "chol_queue_kernel" in x["Name"]][0]
        steps = d["steps"] + d["warmup"] + 1            # timed + warm-up + the untimed per-launch-event step
        calls, avg_ns = int(g["Calls"]), float(g["AverageNs"])
        ach = r["flop_per_launch"] * steps / (avg_ns * calls * 1e-9) / 1e12
        print("| %s | 1 (+%d idle instances) | %.2f | %.2f (frac %.3f) | %.2f | %.4f | %.4f / %.4f |" % (
            tag, round(calls / steps) - 1, steps, r["achieved"], r["frac"], ach, ach / r["achieved"], r["avg_launch_ms"], avg_ns * calls / steps * 1e-6))
        continue
    g = [x for x in rows if "gemm_nt_kernel<256, 128>" in x["Name"]][0]
    calls, avg_ns = int(g["Calls"]), float(g["AverageNs"])
    steps = calls / r["launches"]
    ach = r["flop_per_launch"] * r["launches"] * steps / (avg_ns * calls * 1e-9) / 1e12
    print("| %s | %d | %.2f | %.2f (frac %.3f) | %.2f | %.4f | %.4f / %.4f |" % (tag, r["launches"], steps, r["achieved"], r["frac"], ach,
                                                                            ach / r["achieved"], r["avg_launch_ms"], avg_ns * 1e-6))
