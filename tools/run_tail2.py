"""Micro-driver: the tail chain (azk_nn_tail_gemm x 5), with the two wide links LDS-staged (azk_nn_tail_gemm_lds) or in registers,
against the library-GEMM tail; then each wide link alone in both forms.  usage: run_tail2.py [rows] [live] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "alpha-zero_amd"))
import torch
import azk
from pvnet import NetConfig, PolicyValueNet

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
live = int(sys.argv[2]) if len(sys.argv) > 2 else 917
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
cfg = NetConfig(15, 15, 2, 225, 5, 512, 8, 1)
net = PolicyValueNet(cfg, seed=0, device="cuda", dtype=torch.bfloat16, path="clsfold")
z = (torch.randn(n, 8, 512, device="cuda") * 0.1).to(torch.bfloat16)
net.live_count = torch.tensor([live], dtype=torch.int32, device="cuda")
net.out_buffers = (torch.zeros(n, 225, device="cuda"), torch.zeros(n, device="cuda"))


def timeit(fn, inner=8):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(inner):
            fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps / inner * 1e3


for rnd in range(2):                      # interleaved rounds in one process
    for name, chain, lds in (("chain, wide links LDS-staged", True, True), ("chain, all links in registers", True, False), ("library", False, False)):
        net.use_chain_tail, net.use_lds_tail = chain, lds
        print(f"round {rnd} {name}: rows {n} live {live}: {timeit(lambda: net.tail_fast(z)):.2f} us per tail (graph replay, 8 tails per graph)")
f = net._fold
D = 512
x1 = (torch.randn(n, D, device="cuda")).to(torch.bfloat16)
hh = (torch.randn(n, 4 * D, device="cuda") * 0.5).to(torch.bfloat16)
st = torch.stack([x1.float().view(n, 8, 64).sum(2), (x1.float() ** 2).view(n, 8, 64).sum(2)], dim=2).contiguous()
o3 = torch.empty(n, 4 * D, device="cuda", dtype=torch.bfloat16)
o4 = torch.empty(n, D, device="cuda", dtype=torch.bfloat16)
st2 = torch.empty(n, 8, 2, device="cuda")
cnt = net.live_count
for rnd in range(2):
    for lds in (True, False):
        t3 = timeit(lambda: azk.nn_tail_gemm(x1, f["W0GP"], 4 * D, D, azk.TAIL_GELU, bias=f["b0G_f"], out=o3, a_stats=st, count=cnt, col_sums=f["W0GP_csum"] if lds else None, lds=lds))
        t4 = timeit(lambda: azk.nn_tail_gemm(hh, f["W3P"], D, 4 * D, azk.TAIL_RESID, bias=f["b3_f"], resid=x1, out=o4, stats_out=st2, count=cnt, lds=lds))
        print(f"round {rnd} lds={lds}: link 3 (512 -> 2048, GELU) {t3:.2f} us   link 4 (2048 -> 512, residual) {t4:.2f} us   (back-to-back launches of the same link)")

# the fp32-accurate tail (fp16 hi / lo planes): k_gemm_h x 5, with the two wide links LDS-staged or in registers
netx = PolicyValueNet(cfg, seed=0, device="cuda", dtype=torch.float32, path="clsfold")
zx = torch.randn(n, 8, azk.EMBED_FOLD_ROW, device="cuda") * 0.05
netx.live_count = net.live_count
netx.out_buffers = net.out_buffers
for rnd in range(2):
    for lds in (True, False):
        netx.use_lds_tail = lds
        print(f"round {rnd} fp32-accurate tail, wide links {'LDS-staged' if lds else 'in registers'}: rows {n} live {live}: {timeit(lambda: netx.tail_exact_h(zx)):.2f} us per tail")
