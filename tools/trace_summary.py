"""Per-queue summary of a rocprofv3 --kernel-trace CSV for the LAST factorisation in it:
python tools/trace_summary.py path/to/kernel_trace.csv"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))


def short(n):
    n = n.replace('sgpr::(anonymous namespace)::', '').replace('void ', '')
    return n.split('(')[0][:45]


g = [i for i, r in enumerate(rows) if 'gram_' in r['Kernel_Name']]
R = rows[g[-1] + 2:]
dur = lambda r: int(r['End_Timestamp']) - int(r['Start_Timestamp'])
t0 = int(R[0]['Start_Timestamp'])
t1 = max(int(r['End_Timestamp']) for r in R)
print("kernels", len(R), "span ms", (t1 - t0) / 1e6)
byq = collections.defaultdict(list)
for r in R:
    byq[r['Queue_Id']].append(r)
for q, L in byq.items():
    L.sort(key=lambda r: int(r['Start_Timestamp']))
    print("queue", q, "n", len(L), "busy ms", sum(map(dur, L)) / 1e6)
    agg = collections.defaultdict(lambda: [0, 0])
    for r in L:
        k = short(r['Kernel_Name'])
        agg[k][0] += 1
        agg[k][1] += dur(r)
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("   %-40s n=%4d  %.3f ms  avg %.1f us" % (k, v[0], v[1] / 1e6, v[1] / v[0] / 1e3))
    gaps = [int(L[i + 1]['Start_Timestamp']) - int(L[i]['End_Timestamp']) for i in range(len(L) - 1)]
    if gaps:
        print("   gaps total ms", sum(x for x in gaps if x > 0) / 1e6, "max us", max(gaps) / 1e3)
if len(sys.argv) > 2:
    for r in R[:int(sys.argv[2])]:
        print(r['Queue_Id'], short(r['Kernel_Name']), r['Grid_Size_X'], (int(r['Start_Timestamp']) - t0) // 1000, dur(r) // 1000)
