#!/bin/bash
# round 4: the default bench line + rocprofv3 per-kernel stats of the same command (short window)
#   gpurun --timeout 900 -- 'bash tools/r04_bench_stats.sh [tag] [bench args]'   -> gpurun_out/r04_<tag>/*
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-bench}; shift
O=gpurun_out/r04_$tag
mkdir -p $O
timeout -k 10 400 python3 bench.py --cpu-seconds 0 --steps 20 --warmup 5 "$@" > $O/bench.json 2> $O/bench.err || { echo "bench failed"; tail -5 $O/bench.err; exit 1; }
python3 -c "
import json,sys
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','sims_per_sec')})
for k in d['kernel_rooflines']: print(k['kernel'], round(k['avg_launch_us'],2), round(k['frac'],4))
"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 10 --warmup 5 --cpu-seconds 0 "$@" > $O/bench_under_rocprof.json 2> $O/stats.err || { echo "rocprof failed"; tail -5 $O/stats.err; exit 1; }
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
f=$(find $O/stats -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:14]: print(f"{r['Name'][:110]:110s} calls {r['Calls']:>7s} avg_us {float(r['AverageNs'])/1e3:8.2f} pct {r['Percentage']}")
PY
