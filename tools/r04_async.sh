#!/bin/bash
# round 4: asynchronous moves after the Dirichlet rows left the mover's chain - tests, then lock-step vs async on ONE box
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_async
mkdir -p $O
timeout -k 10 500 python3 -m pytest tests/test_gpu_async.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
line() { python3 -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1])
print('$2', round(d['value'],1), 'games/s', round(d['ms_per_step'],2), 'ms/step', round(d['sims_per_sec']/1e6,2), 'Msims/s', 'launches/move', round(d.get('tree_launches_per_move',0),1))"; }
timeout -k 10 300 python3 bench.py --cpu-seconds 0 --fp32-steps 0 --steps 20 --warmup 5 > $O/lockstep.json 2> $O/lockstep.err && line $O/lockstep.json lockstep
for pl in 1 2; do
  timeout -k 10 300 python3 bench.py --cpu-seconds 0 --fp32-steps 0 --steps 20 --warmup 5 --async-moves 1 --per-launch $pl > $O/async_pl$pl.json 2> $O/async_pl$pl.err && line $O/async_pl$pl.json "async per_launch=$pl"
done
for y in 20 30 45; do
  timeout -k 10 300 python3 bench.py --cpu-seconds 0 --fp32-steps 0 --steps 20 --warmup 5 --async-moves 1 --per-launch 3 --young-us $y > $O/async_young$y.json 2> $O/async_young$y.err && line $O/async_young$y.json "async per_launch=3 young=$y"
done
