#!/bin/bash
# PMC passes of the round (separate runs, --kernel-trace only beside --pmc): HBM traffic, L2 hits, SQ instruction mix of the step's kernels.
#   gpurun -- 'bash tools/profile_pmc.sh [bench args]'        -> gpurun_out/r03_prof/pmc_summary_<first counter>.json
# rocprofv3 sometimes aborts while it starts up under --pmc on this pool (seen with and without our code): every pass is tried up to 4 times.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_prof
mkdir -p $O
( while sleep 45; do echo "[heartbeat $(date +%T)]"; done ) &
HB=$!
B="python3 bench.py --steps 1 --warmup 1 --preroll-cheap 32 --preroll-full 2 --cpu-seconds 0 $*"
RX='k_tree|k_embed_pool|k_embed_fold'
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA"; do
  tag=$(echo $pass | cut -d' ' -f1)
  ok=0
  for try in 1 2 3 4; do
    rm -rf $O/pmc_$tag
    if timeout -k 10 400 rocprofv3 --kernel-trace --pmc $pass --kernel-include-regex "$RX" --output-format csv -d $O/pmc_$tag -- $B > /dev/null 2> $O/pmc_$tag.err; then ok=1; break; fi
    echo "pass $tag try $try failed: $(grep -m1 'SIGSEGV\|Abort\|rror' $O/pmc_$tag.err | cut -c1-120)"
  done
  [ $ok = 1 ] || continue
  python3 tools/pmc_summary.py $O/pmc_$tag > $O/pmc_summary_$tag.json
  rm -rf $O/pmc_$tag                                                   # (tens of MB each: only the summaries travel back)
  echo "pmc $tag done $(date +%T)"
done
kill $HB
du -sh $O
