#!/bin/bash
# PMC passes (separate runs, --kernel-trace only beside --pmc): HBM traffic, L2 hits, SQ instruction mix of the step's kernels.
#   gpurun --timeout 1200 -- 'bash tools/profile_pmc.sh [bench args]'     -> gpurun_out/r04_pmc/{pmc_summary_<tag>.json, pmc_<tag>.err, pmc_status.txt}
# Every pass runs ONCE.  Its exit status and its whole stderr are kept (pmc_status.txt, pmc_<tag>.err): a pass that fails is
# recorded and skipped, never retried - round 3's retry loop hid a start-up abort of rocprofv3 and overwrote its evidence.
# What the preserved failures of round 4 showed (profiles/README.md): SIGSEGV in librocprofiler-sdk.so, called from an HSA-runtime
# worker thread (a dispatch-completion callback), several seconds INTO the run, no frame of ours on that stack; 4 of 5 passes died with
# five kernel families in --kernel-include-regex, 0 of 2 with one.  So a pass profiles ONE kernel family: fewer counter records per second.
# The profiled program is python3 itself (no wrapper process between rocprofv3 and the GPU user) and it spawns no child:
# --cpu-seconds 0 (no CPU-baseline workers), --fp32-steps 0, libazk.so prebuilt.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_pmc
mkdir -p $O
: > $O/pmc_status.txt
( while sleep 45; do echo "[heartbeat $(date +%T)]"; done ) &
HB=$!
B="python3 bench.py --steps 1 --warmup 1 --preroll-cheap 32 --preroll-full 2 --cpu-seconds 0 --fp32-steps 0 $*"
for RX in ${PMC_FAMILIES:-k_tree k_embed_fold k_tail_lds k_tail_gemm}; do        # (PMC_FAMILIES="k_tree": one family again, e.g. after a rename)
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA"; do
  tag=${RX}_$(echo $pass | cut -d' ' -f1)
  t0=$(date +%s)
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $pass --kernel-include-regex "$RX" --output-format csv -d $O/pmc_$tag -- $B > $O/pmc_$tag.out 2> $O/pmc_$tag.err
  rc=$?
  echo "pass $tag rc=$rc seconds=$(( $(date +%s) - t0 )) bench_line=$(grep -c '"metric"' $O/pmc_$tag.out) stderr_lines=$(wc -l < $O/pmc_$tag.err)" | tee -a $O/pmc_status.txt
  if [ $rc = 0 ]; then python3 tools/pmc_summary.py $O/pmc_$tag > $O/pmc_summary_$tag.json; fi
  rm -rf $O/pmc_$tag                                                   # (tens of MB each: only the summaries travel back)
done
done
python3 - $O <<'PY'
import glob, json, os, sys
out = {}
for f in sorted(glob.glob(os.path.join(sys.argv[1], "pmc_summary_*.json"))):
    for k, v in json.load(open(f)).items():
        out.setdefault(k, {}).update(v)
json.dump(out, open(os.path.join(sys.argv[1], "pmc_merged.json"), "w"), indent=1)
PY
kill $HB
du -sh $O
