"""Micro-driver: azk_nn_patch_embed_scores alone with a device-side live count.  usage: run_embed_scores.py [n] [live] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "alpha-zero_amd"))
import ctypes as C
import torch
import azk
from pvnet import NetConfig, PolicyValueNet

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
live = int(sys.argv[2]) if len(sys.argv) > 2 else n
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
cfg = NetConfig(15, 15, 2, 225, 5, 512, 8, 1)
net = PolicyValueNet(cfg, seed=0, device="cuda", dtype=torch.bfloat16, path="clsfold")
x = (torch.rand(n, 2, 15, 15, device="cuda") < 0.1).to(torch.bfloat16)
T, D, H, Tp = 226, 512, 8, 240
xh = torch.empty(n, T, D, device="cuda", dtype=torch.bfloat16)
sc = torch.empty(n, H, Tp, device="cuda")
cnt = torch.tensor([live], dtype=torch.int32, device="cuda")
L = azk.lib()
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())
f, hp = net._fold, net._hip
fn = lambda: L.azk_nn_patch_embed_scores(p(x), 0, p(f["wt_ext"]), p(hp["cpos"]), None, None, p(xh), p(f["score_cpos"]), p(f["score_msum"]), p(sc), H, n, 2, 15, 15, 5, 64, D, 1e-5, p(cnt), st())
assert fn() == 0
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(reps):
    fn()
b.record(); torch.cuda.synchronize()
us = a.elapsed_time(b) / reps * 1e3
print(f"embed+scores n={n} live={live}: {us:.1f} us  ({live * (T * D * 2) / us / 1e3:.0f} GB/s written)")
