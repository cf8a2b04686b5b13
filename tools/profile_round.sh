#!/bin/bash
# The round's rocprofv3 evidence, on the GPU box: per-kernel stats of the default and the fp32 bench, then PMC passes (separate runs,
# --kernel-trace only beside --pmc) for HBM traffic, L2 hits and the SQ instruction mix of the step's kernels.
#   gpurun -- 'bash tools/profile_round.sh'        -> gpurun_out/r03_prof/*
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_prof
mkdir -p $O
B="python3 bench.py --steps 3 --warmup 2 --preroll-full 6 --cpu-seconds 0"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_default -- python3 bench.py --steps 10 --warmup 5 --cpu-seconds 0 > $O/bench_under_rocprof_default.json 2> $O/stats_default.err || exit 1
echo "stats default done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_fp32 -- python3 bench.py --nn-dtype fp32 --steps 6 --warmup 2 --preroll-full 8 --cpu-seconds 0 > $O/bench_under_rocprof_fp32.json 2> $O/stats_fp32.err || exit 1
echo "stats fp32 done"
find $O -name "*kernel_trace.csv" -delete
RX='k_tree|k_embed_pool|k_embed_fold|k_tail_gemm'
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $pass --kernel-include-regex "$RX" --output-format csv -d $O/pmc_$tag -- $B > /dev/null 2> $O/pmc_$tag.err || { echo "pass $tag failed"; tail -3 $O/pmc_$tag.err; continue; }
  find $O/pmc_$tag -name "*kernel_trace.csv" -delete
  echo "pmc $tag done"
done
python3 tools/pmc_summary.py $O/pmc_* > $O/pmc_summary.json
du -sh $O
