#!/bin/bash
# one iteration of the k_tree work: the legal-move / search parity tests, then k_tree's mean time and the sub-phase stamps of
# azk_valid_moves (DBG instantiation) under the micro-driver.  usage: gpurun -- bash tools/r04_tree_iter.sh [tag]
tag=${1:-x}
mkdir -p gpurun_out
python -m pytest tests/test_gpu_engine.py tests/test_gpu_facade.py -x -q > gpurun_out/tree_iter_${tag}_tests.txt 2>&1
rc=$?
tail -n 3 gpurun_out/tree_iter_${tag}_tests.txt
[ $rc -ne 0 ] && exit $rc
python3 tools/run_tree.py > gpurun_out/tree_iter_${tag}_plain.txt 2>&1 && AZK_TREE_ABLATE=32 python3 tools/run_tree.py > gpurun_out/tree_iter_${tag}_a32.txt 2>&1
tail -n 1 gpurun_out/tree_iter_${tag}_plain.txt | cut -c1-60
tail -n 1 gpurun_out/tree_iter_${tag}_a32.txt
