cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for n in 2 4; do
timeout -k 10 240 python3 bench.py --split $n --steps 12 --warmup 6 --cpu-seconds 0 > gpurun_out/r03_split$n.json 2> gpurun_out/r03_split$n.err || { tail -5 gpurun_out/r03_split$n.err; exit 1; }
python3 -c "
import json;d=json.load(open('gpurun_out/r03_split$n.json'));print($n,d['value'],d['ms_per_step'],[ (k['kernel'][:14],round(k['avg_launch_us'],1)) for k in d['kernel_rooflines']])"
done
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr_split2 -- python3 bench.py --split 2 --steps 2 --warmup 1 --preroll-full 4 --cpu-seconds 0 > gpurun_out/tr_split2.json 2> gpurun_out/tr_split2.err && python3 tools/trace_overlap.py gpurun_out/tr_split2 > gpurun_out/r03_split2_overlap.json; rm -rf gpurun_out/tr_split2; cat gpurun_out/r03_split2_overlap.json
