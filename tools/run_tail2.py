"""Micro-driver: the tail chain (azk_nn_tail_gemm x 5) against the library-GEMM tail.  usage: run_tail2.py [rows] [live] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "alpha-zero_amd"))
import torch
import azk
from pvnet import NetConfig, PolicyValueNet

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
live = int(sys.argv[2]) if len(sys.argv) > 2 else 1020
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
cfg = NetConfig(15, 15, 2, 225, 5, 512, 8, 1)
net = PolicyValueNet(cfg, seed=0, device="cuda", dtype=torch.bfloat16, path="clsfold")
z = (torch.randn(n, 8, 512, device="cuda") * 0.1).to(torch.bfloat16)
net.live_count = torch.tensor([live], dtype=torch.int32, device="cuda")
net.out_buffers = (torch.zeros(n, 225, device="cuda"), torch.zeros(n, device="cuda"))


def timeit(fn):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for name, flag in (("chain", True), ("library", False)):
    net.use_chain_tail = flag
    print(f"{name}: rows {n} live {live}: {timeit(lambda: net.tail_fast(z)):.1f} us per tail (graph replay)")
