"""Build profiles/<round>/pmc_traffic_n<order>.json from the two rocprofv3 --pmc passes and the per-launch records of the
MFMA kernel:

  python tools/pmc_traffic_json.py FETCH_counter_collection.csv WRITE_counter_collection.csv launches.json N_PTS [FAMILY [D]] > out.json

launches.json comes from `python tools/gemm_launches.py N --json launches.json`.  FAMILY (default A) and D (canonical pairs
per point, default 1) name the configuration the passes were taken on and pick the Gram kernel's row.
FETCH_SIZE / WRITE_SIZE are reported in KiB.  FETCH_SIZE is doubled (gfx950 reports half of a wide coalesced read); the
doubling is re-calibrated here on trsv_strips_kernel, which reads L exactly once per launch."""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from collections import defaultdict

fetch_csv, write_csv, launches_json, n_pts = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
family = sys.argv[5] if len(sys.argv) > 5 else "A"
dpairs = int(sys.argv[6]) if len(sys.argv) > 6 else 1


def sums(path, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"].replace("void ", "").replace("sgpr::(anonymous namespace)::", "").split("(")[0]
            tot[k] += float(r["Counter_Value"])
            cnt[k] += 1
    return tot, cnt


ft, fc = sums(fetch_csv, "FETCH_SIZE")
wt, wc = sums(write_csv, "WRITE_SIZE")
la = json.load(open(launches_json))
n = 2 * dpairs * n_pts
out = {"config": {"n_pts": n_pts, "order_n": n, "family": family, "triangle": "full", "pairs_per_point": dpairs},
       "code_hash": __import__("bench").kernel_code_hash(),
       "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) of bench.py --steps 1 "
                 "--warmup 0 --cpu-sample 0 --no-launch-events (bench.py itself skips its event pass under the profiler); FETCH_SIZE doubled per the gfx950 note, calibrated on "
                 "trsv_strips_kernel (calibration_trsv_strips below); traffic = (2 FETCH + WRITE) KiB * 1024 / launches"}
gram_keys = [k for k in ft if k.startswith("gram_nd_kernel" if dpairs > 1 else "gram_pairs_kernel")]
gram_key = max(gram_keys, key=lambda k: wt.get(k, 0.0)) if gram_keys else "gram_pairs_kernel"
gram_name = "gram_nd_kernel" if dpairs > 1 else "gram_pairs_kernel"
for key, name in (("gemm_nt_kernel<256, 128>", "gemm_nt_kernel<256, 128>"), (gram_key, gram_name)):
    f, w, c = ft.get(key, 0.0), wt.get(key, 0.0), max(fc.get(key, 0), 1)
    out[name] = {"launches": fc.get(key, 0), "fetch_kb_total": f, "write_kb_total": w,
                 "traffic_bytes_per_launch": (2.0 * f + w) * 1024.0 / c}
g = out["gemm_nt_kernel<256, 128>"]
g["algorithmic_bytes_per_launch"] = la["big_compulsory_bytes"] / max(la["big_launches"], 1)
g["algorithmic_flop_per_launch"] = la["big_flop"] / max(la["big_launches"], 1)
g["launches_in_event_run"] = la["big_launches"]
out[gram_name]["algorithmic_bytes_per_launch"] = 8.0 * n * n
# calibration of the FETCH_SIZE doubling on a kernel whose reads are known exactly: the strip solves read L
# once each (8 n^2 / 2 bytes per launch, two launches per step) with 16-byte loads
gvs = [k for k in ft if k.startswith("trsv_strips_kernel")]            # <true> forward, <false> backward
if gvs:
    f_tot, launches = sum(ft[k] for k in gvs), sum(fc[k] for k in gvs)
    out["calibration_trsv_strips"] = {"launches": launches, "fetch_kb_total_reported": f_tot,
                                      "bytes_read_algorithmic_total": 8.0 * n * n / 2 * launches,
                                      "reported_over_algorithmic": f_tot * 1024.0 / (8.0 * n * n / 2 * launches)}
print(json.dumps(out, indent=1))
