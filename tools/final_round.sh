#!/bin/bash
# The round's closing evidence on the GPU box: GPU suite, network parity, the bench lines, rocprofv3 per-kernel stats.
#   gpurun --timeout 1200 -- 'bash tools/final_round.sh'     -> gpurun_out/r03_final/*
# (the default bench line quotes profiles/r03_nn_parity.json and profiles/r03_bench_fp32.json: both are refreshed on the box first)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_final
mkdir -p $O
run() { name=$1; shift; timeout -k 10 300 "$@" > $O/$name.json 2> $O/$name.err || { echo "$name failed"; tail -5 $O/$name.err; exit 1; }; }
timeout -k 10 500 python3 -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1 || { tail -20 $O/gputests.log; exit 1; }
tail -2 $O/gputests.log
run nn_parity python3 tools/measure_nn_parity.py 800
cp $O/nn_parity.json profiles/r03_nn_parity.json
echo "parity done"
run bench_fp32 python3 bench.py --nn-dtype fp32 --steps 40 --warmup 10 --cpu-seconds 0
cp $O/bench_fp32.json profiles/r03_bench_fp32.json
cut -c1-200 $O/bench_fp32.json
run bench_default python3 bench.py
cut -c1-200 $O/bench_default.json
for k in 2 4; do run bench_virtual_loss_k$k python3 bench.py --virtual-loss $k --steps 20 --warmup 5 --cpu-seconds 0; done
run bench_async_per_launch1 python3 bench.py --async-moves 1 --per-launch 1 --steps 20 --warmup 5 --cpu-seconds 0
run bench_config5_n1 python3 bench.py --train-step --steps 16 --warmup 20 --cpu-seconds 0
run bench_embed_conv python3 bench.py --embed conv --steps 20 --warmup 5 --cpu-seconds 0
run bench_long_160_steps python3 bench.py --steps 160 --warmup 60 --cpu-seconds 0
echo "side lines done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_default -- python3 bench.py --steps 10 --warmup 5 --cpu-seconds 0 > $O/bench_under_rocprof_default.json 2> $O/stats_default.err || exit 1
echo "stats default done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_fp32 -- python3 bench.py --nn-dtype fp32 --steps 6 --warmup 2 --preroll-full 8 --cpu-seconds 0 > $O/bench_under_rocprof_fp32.json 2> $O/stats_fp32.err || exit 1
echo "stats fp32 done"
find $O -name "*kernel_trace.csv" -delete
find $O -name "*agent_info.csv" -delete
du -sh $O
