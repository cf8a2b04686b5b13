cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "1 0" "2 8"; do
  set -- $cfg
  O=gpurun_out/st_async_$1_$2
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --async-moves 1 --per-launch $1 --young-us $2 --steps 6 --warmup 3 --cpu-seconds 0 > $O.json 2> $O.err
  find $O -name "*kernel_trace.csv" -delete
  echo "== per_launch $1 young $2"; python3 -c "
import json;d=json.load(open('$O.json'));print(d['ms_per_step'], d.get('tree_launches_per_move'))"
  python3 - <<PY
import csv,glob
f=glob.glob('$O/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print('  ', r['Name'][27:80].split('(')[0], r['Calls'], round(float(r['AverageNs'])/1e3,2), r['Percentage'])
PY
done
