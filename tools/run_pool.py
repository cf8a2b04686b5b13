"""Micro-driver: time azk_nn_cls_pool / azk_nn_cls_attention / embed+scores alone.  usage: run_pool.py [n] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "alpha-zero_amd"))
import ctypes as C
import torch
import azk
from pvnet import NetConfig, PolicyValueNet

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
cfg = NetConfig(15, 15, 2, 225, 5, 512, 8, 1)
net = PolicyValueNet(cfg, seed=0, device="cuda", dtype=torch.bfloat16, path="clsfold")
x = (torch.rand(n, 2, 15, 15, device="cuda") < 0.1).to(torch.bfloat16)
T, D, H = 226, 512, 8
xh = torch.randn(n, T, D, device="cuda").to(torch.bfloat16)
sc = torch.randn(n, H, 240, device="cuda")
z = torch.empty(n, H, D, device="cuda", dtype=torch.bfloat16)
L = azk.lib()
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())
f, hp = net._fold, net._hip


def timeit(name, fn):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    print(f"{name}: {a.elapsed_time(b) / reps * 1e3:.1f} us")


timeit("cls_pool", lambda: L.azk_nn_cls_pool(p(xh), p(sc), p(f["c"]), p(z), n, T, D, H, None, st()))
timeit("cls_attention(v1)", lambda: azk.nn_cls_attention(xh, f["m"], f["c"], H))
timeit("embed xhat only", lambda: net.embed_hip(x, False, True))
timeit("embed+scores+pool", lambda: azk.nn_embed_scores_pool(x, f["wt_ext"], hp["cpos"], f["score_cpos"], f["score_msum"], f["c_n"], 15, 15, 5, 512, 8))
