#!/bin/bash
# Same-box A/B of the round's tree rework: the default bench with the closing library and with the library built from the commit before
# the rework (563af85: `git archive 563af85 alpha-zero_amd/csrc include | tar -x -C /tmp/old && make -C /tmp/old/alpha-zero_amd/csrc`,
# copied in-tree as alpha-zero_amd/azk/libazk_before_tree_rework.so - git-ignored, travels with the gpurun snapshot).  Same Python, same ABI.
#   gpurun --timeout 900 -- 'bash tools/r04_tree_ab.sh'     -> gpurun_out/r04_tree_ab.txt
cd "$GRAFT_REPO_ROOT" || exit 1
L=alpha-zero_amd/azk
[ -f $L/libazk_before_tree_rework.so ] || { echo "no old library"; exit 1; }
O=gpurun_out/r04_tree_ab.txt
: > $O
line() { python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k={x['kernel'][:12]:round(x['avg_launch_us'],1) for x in d['kernel_rooflines']}
print(sys.argv[2], round(d['value'],1),'games/s', round(d['ms_per_step'],2),'ms/move', k)" "$1" "$2" | tee -a $O; }
cp $L/libazk.so /tmp/libazk_new.so
for rep in 1 2; do
  python3 bench.py --steps 30 --warmup 8 --cpu-seconds 0 --fp32-steps 0 > gpurun_out/ab_new_$rep.json 2> gpurun_out/ab_new_$rep.err || exit 1
  line gpurun_out/ab_new_$rep.json "closing build      "
  cp $L/libazk_before_tree_rework.so $L/libazk.so
  python3 bench.py --steps 30 --warmup 8 --cpu-seconds 0 --fp32-steps 0 > gpurun_out/ab_old_$rep.json 2> gpurun_out/ab_old_$rep.err || { cp /tmp/libazk_new.so $L/libazk.so; exit 1; }
  line gpurun_out/ab_old_$rep.json "before tree rework "
  cp /tmp/libazk_new.so $L/libazk.so
done
