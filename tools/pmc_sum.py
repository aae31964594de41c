"""Sum a rocprofv3 --pmc counter_collection.csv per kernel: launches, total and per-launch value.
usage: python tools_pmc_sum.py <counter_collection.csv> [more.csv ...]"""
import csv
import sys
from collections import defaultdict

for path in sys.argv[1:]:
    tot = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    with open(path) as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"].replace("void ", "").replace("sgpr::(anonymous namespace)::", "").split("(")[0]
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]] += 1
    print("#", path)
    for k in sorted(tot, key=lambda k: -max(tot[k].values())):
        for c in tot[k]:
            print("%-40s %-12s launches %6d  total %.6g  per-launch %.6g" % (k[:40], c, cnt[k][c], tot[k][c],
                                                                            tot[k][c] / cnt[k][c]))
