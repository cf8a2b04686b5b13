"""What a dependent launch costs inside a captured graph on this GPU (the floor of the tail's small links): chains of (a) a one-block
LayerNorm over 4 rows, (b) the tail's small links at 16 / 917 live rows, replayed back to back.  usage: run_floor.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "alpha-zero_amd"))
import torch
import azk
from pvnet import NetConfig, PolicyValueNet

cfg = NetConfig(15, 15, 2, 225, 5, 512, 8, 1)
net = PolicyValueNet(cfg, seed=0, device="cuda", dtype=torch.bfloat16, path="clsfold")
f = net._fold
D, H, n = 512, 8, 2048


def timeit(fn, inner=16, reps=100):
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


x4 = torch.randn(4, D, device="cuda").to(torch.bfloat16)
y4 = torch.empty_like(x4)
w, b = f["ln2_w"], f["ln2_b"]
print(f"one-block LayerNorm over 4 rows, chained: {timeit(lambda: azk.lib().azk_nn_layernorm_rows(x4.data_ptr(), w.data_ptr(), b.data_ptr(), 1e-5, y4.data_ptr(), None, 4, D, None, torch.cuda.current_stream().cuda_stream)):.2f} us per launch")
u = torch.randn(n, D, device="cuda").to(torch.bfloat16)
x1 = torch.empty(n, D, device="cuda", dtype=torch.bfloat16)
st1 = torch.empty(n, 8, 2, device="cuda")
rows = torch.randn(n, H * azk.EMBED_FOLD_ROW, device="cuda").to(torch.bfloat16)
lb, vb = torch.empty(n, 225, device="cuda"), torch.empty(n, device="cuda")
for live in (16, 256, 917):
    cnt = torch.tensor([live], dtype=torch.int32, device="cuda")
    t1 = timeit(lambda: azk.nn_tail_gemm(rows, net._foldu.weight, D // H, azk.EMBED_FOLD_ROW, azk.TAIL_BF16, nbatch=H, a_batch_stride=azk.EMBED_FOLD_ROW, out=u, count=cnt))
    t2 = timeit(lambda: azk.nn_tail_gemm(u, f["WoP"], D, D, azk.TAIL_BF16, bias=f["bias1_f"], out=x1, stats_out=st1, count=cnt))
    t5 = timeit(lambda: azk.nn_tail_gemm(x1, f["WhGP"], f["bhG_f"].numel(), D, azk.TAIL_HEADS, bias=f["bhG_f"], a_stats=st1, logits=lb, values=vb, action_dim=225, count=cnt))
    print(f"live {live}: link 1 (rows -> u, K = 384 x 8 heads) {t1:.2f} us   link 2 (out-proj, 512 -> 512 + stats) {t2:.2f} us   link 5 (LN + heads, 512 -> 256) {t5:.2f} us")
