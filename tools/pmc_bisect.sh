#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
i=0
for extra in "" "--no-graph" "--cache-entries 0" "--timer-stride 100000"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "k_tree" --output-format csv -d gpurun_out/pmc_bis$i -- python3 bench.py --steps 1 --warmup 0 --preroll-cheap 0 --preroll-full 1 --cpu-seconds 0 $extra > /dev/null 2> gpurun_out/pmc_bis$i.err
  echo "variant [$extra] rc=$? files: $(ls gpurun_out/pmc_bis$i/*/ 2>/dev/null | tr '\n' ' ')"
  grep -m2 "SIGSEGV\|Abort" gpurun_out/pmc_bis$i.err
  find gpurun_out/pmc_bis$i -name "*.csv" -size +1M -delete
done
