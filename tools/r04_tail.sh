#!/bin/bash
# round 4, tail experiment: parity tests of the LDS-staged wide links, then timings (whole chain + each wide link, both forms)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_tail
mkdir -p $O
timeout -k 10 500 python3 -m pytest tests/test_gpu_tail_lds.py tests/test_gpu_nn.py tests/test_gpu_exact.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -15 $O/tests.log
timeout -k 10 200 python3 tools/run_tail2.py 2048 917 200 > $O/time_917.txt 2>&1 && grep "round 1" $O/time_917.txt
timeout -k 10 200 python3 tools/run_tail2.py 2048 1100 200 > $O/time_1100.txt 2>&1 && grep "round 1" $O/time_1100.txt
