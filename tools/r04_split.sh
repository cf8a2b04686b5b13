#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_split
mkdir -p $O
for s in 1 2 4 1 2; do
  timeout -k 10 300 python3 bench.py --cpu-seconds 0 --fp32-steps 0 --steps 20 --warmup 5 --split $s > $O/split$s.json 2> $O/split$s.err || { echo "split $s failed"; tail -3 $O/split$s.err; continue; }
  python3 -c "
import json
d=json.loads(open('$O/split$s.json').read().strip().splitlines()[-1])
print('split=$s', round(d['value'],1), 'games/s', round(d['ms_per_step'],2), 'ms/step', round(d['sims_per_sec']/1e6,2), 'Msims/s')"
done
