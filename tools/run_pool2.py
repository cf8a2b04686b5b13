"""Micro-driver: azk_nn_cls_pool alone with a device-side live count.  usage: run_pool2.py [n] [live] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "alpha-zero_amd"))
import ctypes as C
import torch
import azk
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
live = int(sys.argv[2]) if len(sys.argv) > 2 else n
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
T, D, H = 226, 512, 8
xh = torch.randn(n, T, D, device="cuda").to(torch.bfloat16)
sc = torch.randn(n, H, 240, device="cuda")
cc = torch.zeros(H, device="cuda")
z = torch.empty(n, H, D, device="cuda", dtype=torch.bfloat16)
cnt = torch.tensor([live], dtype=torch.int32, device="cuda")
L = azk.lib()
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())
fn = lambda: L.azk_nn_cls_pool(p(xh), p(sc), p(cc), p(z), n, T, D, H, p(cnt), st())
assert fn() == 0
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(reps):
    fn()
b.record(); torch.cuda.synchronize()
us = a.elapsed_time(b) / reps * 1e3
print(f"cls_pool n={n} live={live}: {us:.1f} us  ({live * (T * D * 2) / us / 1e3:.0f} GB/s read)")
