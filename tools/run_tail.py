"""Micro-driver: the cls-row tail (tail_fast: 4 GEMMs + 2 LayerNorm + finalize) on n rows, under a captured graph.
usage: run_tail.py [n] [reps]   (env PYTORCH_TUNABLEOP_ENABLED=1 to let PyTorch pick the GEMM solutions by measurement)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "alpha-zero_amd"))
import torch
import azk
from pvnet import NetConfig, PolicyValueNet

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
cfg = NetConfig(15, 15, 2, 225, 5, 512, 8, 1)
net = PolicyValueNet(cfg, seed=0, device="cuda", dtype=torch.bfloat16, path="clsfold")
z = torch.randn(n, 8, 512, device="cuda").to(torch.bfloat16) * 0.1
lb = torch.empty(n, 225, device="cuda"); vb = torch.empty(n, device="cuda")
net.out_buffers = (lb, vb)
net.use_hip_tail = bool(os.environ.get("HIP_TAIL"))
net.fuse_ln_heads = not os.environ.get("NO_FUSE_HEADS")
live = int(os.environ.get("LIVE", n))
net.live_count = torch.tensor([live], dtype=torch.int32, device="cuda")
if len(sys.argv) > 3:     # accuracy: hand-written tail vs library tail on the same z
    with torch.no_grad():
        net.use_hip_tail = False; net.fuse_ln_heads = True; l1, v1 = net.tail_fast(z); l1, v1 = l1.clone(), v1.clone()
        net.fuse_ln_heads = False; l2, v2 = net.tail_fast(z)
    torch.cuda.synchronize()
    print("fused ln+heads vs library tail: logits max diff", (l1[:live] - l2[:live]).abs().max().item(), "values", (v1[:live] - v2[:live]).abs().max().item(),
          "| logits scale", l2[:live].abs().mean().item())
    net.use_hip_tail = bool(os.environ.get("HIP_TAIL")); net.fuse_ln_heads = not os.environ.get("NO_FUSE_HEADS")
t0 = time.time()
with torch.no_grad():
    for _ in range(3):
        net.tail_fast(z)
    torch.cuda.synchronize()
    print(f"warm-up (incl. any tuning): {time.time() - t0:.1f} s")
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        net.tail_fast(z)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            net.tail_fast(z)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g.replay(); torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        g.replay()
    b.record(); torch.cuda.synchronize()
print(f"tail n={n}: {a.elapsed_time(b) / reps * 1e3:.1f} us per call (graph replay)")
