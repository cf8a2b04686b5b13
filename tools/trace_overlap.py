#!/usr/bin/env python3
"""Do the kernels of the game groups (`bench.py --split N`: one graph per group, each on its own stream) overlap on the GPU?
Reads a rocprofv3 --kernel-trace CSV: per kernel name the mean duration; per queue the busy time; the time with >= 2 kernels in flight.
    rocprofv3 --kernel-trace --output-format csv -d D -- python3 bench.py --split 2 ...;  python3 tools/trace_overlap.py D"""
import csv, glob, json, re, sys, collections

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), (re.search(r"k_\w+(<[^>]*>)?", r["Kernel_Name"]) or re.search(r"\w+", r["Kernel_Name"])).group(0), r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
# the steady part: the last 60 % of the launches
rows = rows[int(len(rows) * 0.4):]
per = collections.defaultdict(list)
queues = collections.defaultdict(int)
for s, e, n, q, st in rows:
    per[n].append(e - s)
    queues[(q, st)] += e - s
ev = sorted([(s, 1) for s, e, *_ in rows] + [(e, -1) for s, e, *_ in rows])
depth, last, busy1, busy2 = 0, ev[0][0], 0, 0
for t, d in ev:
    if depth >= 1:
        busy1 += t - last
    if depth >= 2:
        busy2 += t - last
    depth, last = depth + d, t
span = rows[-1][1] - rows[0][0]
print(json.dumps({"launches": len(rows), "span_ms": span / 1e6, "time_with_a_kernel_running_ms": busy1 / 1e6, "time_with_two_or_more_running_ms": busy2 / 1e6,
                  "busy_per_queue_stream_ms": {f"{q}/{st}": v / 1e6 for (q, st), v in queues.items()},
                  "mean_us": {n: round(sum(v) / len(v) / 1e3, 2) for n, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:12]},
                  "calls": {n: len(v) for n, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:12]}}, indent=1))
