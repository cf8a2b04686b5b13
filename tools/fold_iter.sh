#!/bin/bash
# one iteration on k_embed_fold: its tests, its launch time per batch size, the bench line
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 300 python3 -m pytest tests/test_gpu_fold.py -m gpu -x -q 2>&1 | tail -4 || exit 1
timeout -k 10 200 python3 tools/run_embed_fold.py 914 1400 > gpurun_out/r03_embed_fold_times.json 2>gpurun_out/ef.err && tr -d "\n " < gpurun_out/r03_embed_fold_times.json && echo
timeout -k 10 240 python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 > gpurun_out/r03_bench_fold.json 2> gpurun_out/r03_bench_fold.err || { tail -5 gpurun_out/r03_bench_fold.err; exit 1; }
python3 -c "
import json;d=json.load(open('gpurun_out/r03_bench_fold.json'));print(d['value'],d['ms_per_step'],d['sims_per_sec'],[ (k['kernel'][:14],round(k['avg_launch_us'],1),round(k['frac'],3)) for k in d['kernel_rooflines']])"
