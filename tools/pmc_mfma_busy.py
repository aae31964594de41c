"""Per-kernel MFMA-busy summary of one `rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES
SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE` pass.
    python tools/pmc_mfma_busy.py <p_counter_collection.csv> [n_cu = 256]

Columns: launches; the counters summed over the launches; and
  mfma_busy  = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * n_cu * 4)
               the MfmaUtil formula of rocprofiler's derived counters (busy cycles summed over the SIMDs of all CUs, against
               the cycles the GPU was active: rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs; 4 SIMDs per CU)
  f64_tflop  = SQ_INSTS_VALU_MFMA_MOPS_F64 * 512 flop (one MOP = 512 flop: a v_mfma_f64_16x16x4 is 2048 flop = 4 MOPs),
               the flop the matrix cores EXECUTED (tile round-up and the skipped upper triangle included / excluded as run)
  mfma_cyc_per_flop = busy cycles per executed flop (1/32 per SIMD at the fp64 dense peak: 78.6 TFLOP/s / 1024 SIMDs / 2.4 GHz)"""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
n_cu = int(sys.argv[2]) if len(sys.argv) > 2 else 256
tot = defaultdict(lambda: defaultdict(float))
disp = defaultdict(set)
with open(path) as f:
    for r in csv.DictReader(f):
        k = r["Kernel_Name"].replace("void ", "").replace("sgpr::(anonymous namespace)::", "").replace("sgpr::", "").split("(")[0]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r.get("Dispatch_Id", r.get("Correlation_Id", "")))
print("# %s" % path)
print("%-44s %8s %14s %14s %14s %14s %9s %10s" % ("kernel", "launches", "MFMA_BUSY_CYC", "SQ_BUSY_CYC", "MOPS_F64", "GUI_ACTIVE",
                                                 "mfma_busy", "f64_tflop"))
for k in sorted(tot, key=lambda k: -tot[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)):
    c = tot[k]
    mb, sb, mo, ga = (c.get(n, 0.0) for n in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F64",
                                              "GRBM_GUI_ACTIVE"))
    busy = mb / (ga / 8.0 * n_cu * 4.0) if ga > 0 else float("nan")
    print("%-44s %8d %14.6g %14.6g %14.6g %14.6g %9.4f %10.4g" % (k[:44], len(disp[k]), mb, sb, mo, ga, busy, mo * 512 / 1e12))
