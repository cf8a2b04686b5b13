"""Gaps between consecutive kernels of the stepping graph, from a rocprofv3 --kernel-trace CSV (…_kernel_trace.csv).
usage: trace_gaps.py <kernel_trace.csv>   -> per kernel name: mean duration, mean gap to the previous kernel's end"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur, gap, cnt = collections.defaultdict(float), collections.defaultdict(float), collections.Counter()
prev_end = None
for r in rows:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if prev_end is not None and s - prev_end < 50000:      # ignore host stalls
        gap[n] += s - prev_end; dur[n] += e - s; cnt[n] += 1
    prev_end = e
tot = 0.0
for n, c in cnt.most_common(12):
    print(f"{n:62s} calls {c:7d} dur {dur[n]/c/1e3:7.2f} us  gap-before {gap[n]/c/1e3:6.2f} us")
    tot += (dur[n] + gap[n]) / c if c > 1000 else 0
print("sum(dur+gap) of the frequent kernels:", round(tot / 1e3, 2), "us")
