#!/bin/bash
# One-off diagnosis of the rocprofv3 --pmc SIGSEGV (profiles/README.md, round 4): two single passes, each run ONCE, with the
# process's /proc/self/maps saved (AZK_DUMP_MAPS) so that the stack trace's addresses can be attributed to a library and offset.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_pmc_diag
mkdir -p $O
for v in graph nograph; do
  extra=""; [ $v = nograph ] && extra="--no-graph"
  AZK_DUMP_MAPS=$O/maps_$v.txt timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "k_tree" --output-format csv -d $O/pmc_$v -- python3 bench.py --steps 1 --warmup 1 --preroll-cheap 32 --preroll-full 2 --cpu-seconds 0 --fp32-steps 0 $extra > $O/$v.out 2> $O/$v.err
  echo "variant $v rc=$? bench_line=$(grep -c '"metric"' $O/$v.out)"
  find $O/pmc_$v -name "*.csv" -size +200k -delete
done
