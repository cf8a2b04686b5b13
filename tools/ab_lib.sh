#!/bin/bash
# Same-box A/B of two builds of libazk.so: the default bench, alternating twice, with the in-tree library and with another one of the same ABI
# (given relative to the repo root; a *.so in-tree is git-ignored and travels with the gpurun snapshot).
#   gpurun --timeout 900 -- 'bash tools/ab_lib.sh alpha-zero_amd/azk/libazk_variant.so [label]'     -> gpurun_out/ab_<label>.txt
cd "$GRAFT_REPO_ROOT" || exit 1
L=alpha-zero_amd/azk
V=${1:?library}
T=${2:-variant}
[ -f "$V" ] || { echo "no $V"; exit 1; }
O=gpurun_out/ab_$T.txt
: > $O
line() { python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k={x['kernel'][:12]:round(x['avg_launch_us'],1) for x in d['kernel_rooflines']}
print(sys.argv[2], round(d['value'],1),'games/s', round(d['ms_per_step'],2),'ms/move', k)" "$1" "$2" | tee -a $O; }
cp $L/libazk.so /tmp/libazk_base.so
cp "$V" /tmp/libazk_var.so
for rep in 1 2; do
  python3 bench.py --steps 30 --warmup 8 --cpu-seconds 0 --fp32-steps 0 > gpurun_out/ab_base_$rep.json 2> gpurun_out/ab_base_$rep.err || exit 1
  line gpurun_out/ab_base_$rep.json "in-tree library "
  cp /tmp/libazk_var.so $L/libazk.so
  python3 bench.py --steps 30 --warmup 8 --cpu-seconds 0 --fp32-steps 0 > gpurun_out/ab_var_$rep.json 2> gpurun_out/ab_var_$rep.err || { cp /tmp/libazk_base.so $L/libazk.so; exit 1; }
  line gpurun_out/ab_var_$rep.json "$T "
  cp /tmp/libazk_base.so $L/libazk.so
done
