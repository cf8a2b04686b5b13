#!/bin/bash
# asynchronous moves: (simulations per launch, launch-age limit in us) sweep on the bench workload
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 300 python3 -m pytest tests/test_gpu_async.py -m gpu -x -q 2>&1 | tail -3 || exit 1
for cfg in "1 0" "2 8" "2 12" "2 16" "3 12" "3 16" "4 20"; do
  set -- $cfg
  timeout -k 10 240 python3 bench.py --async-moves 1 --per-launch $1 --young-us $2 --steps 16 --warmup 4 --cpu-seconds 0 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('per_launch $1 young $2:', round(d['value'],1), round(d['ms_per_step'],2), round(d['sims_per_sec']/1e6,2), d.get('tree_launches_per_move'), [(k['kernel'][:12],round(k['avg_launch_us'],1)) for k in d['kernel_rooflines'][:2]])"
done
