#!/bin/bash
# The round's closing evidence on the GPU box: GPU suite, the two bench lines, rocprofv3 per-kernel stats of each.
#   gpurun --timeout 1200 -- 'bash tools/final_round.sh'     -> gpurun_out/r03_final/*
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_final
mkdir -p $O
timeout -k 10 500 python3 -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1 || { tail -20 $O/gputests.log; exit 1; }
tail -2 $O/gputests.log
timeout -k 10 240 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -5 $O/bench_default.err; exit 1; }
cat $O/bench_default.json | cut -c1-400
timeout -k 10 240 python3 bench.py --nn-dtype fp32 --steps 40 --warmup 10 --cpu-seconds 0 > $O/bench_fp32.json 2> $O/bench_fp32.err || { tail -5 $O/bench_fp32.err; exit 1; }
cat $O/bench_fp32.json | cut -c1-300
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_default -- python3 bench.py --steps 10 --warmup 5 --cpu-seconds 0 > $O/bench_under_rocprof_default.json 2> $O/stats_default.err || exit 1
echo "stats default done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_fp32 -- python3 bench.py --nn-dtype fp32 --steps 6 --warmup 2 --preroll-full 8 --cpu-seconds 0 > $O/bench_under_rocprof_fp32.json 2> $O/stats_fp32.err || exit 1
echo "stats fp32 done"
find $O -name "*kernel_trace.csv" -delete
find $O -name "*agent_info.csv" -delete
du -sh $O
