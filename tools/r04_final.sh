#!/bin/bash
# The round's closing evidence on ONE GPU box: GPU suite, the bench lines, rocprofv3 per-kernel stats.
#   gpurun --timeout 1200 -- 'bash tools/r04_final.sh'     -> gpurun_out/r04_final/*     (PMC passes: tools/profile_pmc.sh, its own call)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_final
mkdir -p $O
run() { name=$1; shift; timeout -k 10 300 "$@" > $O/$name.json 2> $O/$name.err || { echo "$name failed"; tail -5 $O/$name.err; exit 1; }; echo "$name: $(python3 -c "
import json
d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
print(round(d['value'],1), d['unit'], round(d['ms_per_step'],2), 'ms/step', round(d['sims_per_sec']/1e6,2), 'M sims/s', ('fp32_line %.1f games/s %.2f ms' % (d['fp32_line']['games_per_sec'], d['fp32_line']['ms_per_step'])) if d.get('fp32_line') else '')")"; }
timeout -k 10 700 python3 -m pytest tests -m gpu -q > $O/gputests.log 2>&1 || { tail -20 $O/gputests.log; exit 1; }
tail -1 $O/gputests.log
run bench_default python3 bench.py
run bench_tail_wide_registers python3 bench.py --tail-wide registers --steps 20 --warmup 5 --cpu-seconds 0 --fp32-steps 0
run bench_fp32 python3 bench.py --nn-dtype fp32 --steps 40 --warmup 10 --cpu-seconds 0
run bench_fp32_tail_wide_registers python3 bench.py --nn-dtype fp32 --tail-wide registers --steps 20 --warmup 5 --cpu-seconds 0
run bench_long_160_steps python3 bench.py --steps 160 --warmup 60 --cpu-seconds 0 --fp32-steps 0
for k in 2 4; do run bench_virtual_loss_k$k python3 bench.py --virtual-loss $k --steps 20 --warmup 5 --cpu-seconds 0; done
run bench_config5_n1 python3 bench.py --train-step --steps 16 --warmup 20 --cpu-seconds 0
echo "side lines done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_default -- python3 bench.py --steps 10 --warmup 5 --cpu-seconds 0 --fp32-steps 0 > $O/bench_under_rocprof_default.json 2> $O/stats_default.err || exit 1
echo "stats default done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_fp32 -- python3 bench.py --nn-dtype fp32 --steps 6 --warmup 2 --preroll-full 8 --cpu-seconds 0 > $O/bench_under_rocprof_fp32.json 2> $O/stats_fp32.err || exit 1
echo "stats fp32 done"
find $O -name "*kernel_trace.csv" -delete
find $O -name "*agent_info.csv" -delete
du -sh $O
