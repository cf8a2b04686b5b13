"""Micro-driver: launch azk_nn_patch_embed a few times (for rocprofv3 / timing).  usage: run_embed.py [n] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "alpha-zero_amd"))
import torch
from pvnet import NetConfig, PolicyValueNet

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
cfg = NetConfig(15, 15, 2, 225, 5, 512, 8, 1)
net = PolicyValueNet(cfg, seed=0, device="cuda", dtype=torch.bfloat16, path="cls")
x = (torch.rand(n, 2, 15, 15, device="cuda") < 0.1).to(torch.bfloat16)
for want_x, want_xhat in ((False, True), (True, False), (True, True)):
    net.embed_hip(x, want_x, want_xhat)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        net.embed_hip(x, want_x, want_xhat)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    nbytes = n * 226 * 512 * 2 * (int(want_x) + int(want_xhat))
    print(f"x={want_x} xhat={want_xhat}: {ms*1e3:.1f} us  {nbytes/ms/1e6:.0f} GB/s written")
