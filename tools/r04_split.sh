#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_split
mkdir -p $O
run() { tag=$1; shift; timeout -k 10 300 python3 bench.py --cpu-seconds 0 --fp32-steps 0 --steps 20 --warmup 5 "$@" > $O/$tag.json 2> $O/$tag.err || { echo "$tag failed"; tail -3 $O/$tag.err; return; }
  python3 -c "
import json
d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1])
print('$tag', round(d['value'],1), 'games/s', round(d['ms_per_step'],2), 'ms/step', round(d['sims_per_sec']/1e6,2), 'Msims/s')"; }
run split1
for s in 4 8; do for f in 32 64 96 128; do run split${s}_fit$f --split $s --split-fit $f; done; done
run split16_fit32 --split 16 --split-fit 32
run split16_fit64 --split 16 --split-fit 64
run split1b
