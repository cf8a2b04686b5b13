"""Micro-driver: the five links of the fp32-accurate tail (azk_nnx_gemm_h / azk_nnx_gemm) in isolation, time per launch.
usage: run_gemm_h.py [rows live] [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "alpha-zero_amd"))
import torch

import azk

live = int(sys.argv[1]) if len(sys.argv) > 1 else 918
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
m, D = 2048, 512
g = torch.Generator("cuda").manual_seed(1)
rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
cnt = torch.tensor([live], dtype=torch.int32, device="cuda")
pl = lambda c: (torch.zeros((m, c), device="cuda", dtype=torch.float16), torch.zeros((m, c), device="cuda", dtype=torch.float16))


def timeit(fn):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


z = rn(m, 8 * D)
wv = torch.cat([azk.pack_linear_weight_h(rn(64, D) * 0.05)[0].reshape(-1) for _ in range(8)])
wo, _ = azk.pack_linear_weight_h(rn(D, D) * 0.05)
w0, cs0 = azk.pack_linear_weight_h(rn(4 * D, D) * 0.05)
w3, _ = azk.pack_linear_weight_h(rn(D, 4 * D) * 0.02)
wh, csh = azk.pack_linear_weight_h(rn(256, D) * 0.05)
b512, b2048, b256 = rn(D), rn(4 * D), rn(256)
u, x1, hh, x2 = pl(D), pl(D), pl(4 * D), pl(D)
x1f = torch.zeros(m, D, device="cuda")
st1, st2 = torch.zeros(m, 8, 2, device="cuda"), torch.zeros(m, 8, 2, device="cuda")
lg, vl = torch.zeros(m, 225, device="cuda"), torch.zeros(m, device="cuda")
links = [
    ("value-proj (f32 A, 8 x [64 x 512])", lambda: azk.nnx_gemm_h(z, wv, 64, D, azk.TAIL_BF16, nbatch=8, a_batch_stride=D, out=u, count=cnt)),
    ("out-proj [512 x 512] + stats", lambda: azk.nnx_gemm_h(u, wo, D, D, azk.TAIL_BF16, bias=b512, out=x1, out_f32=x1f, stats_out=st1, count=cnt)),
    ("LN + MLP up [2048 x 512] + GELU", lambda: azk.nnx_gemm_h(x1, w0, 4 * D, D, azk.TAIL_GELU, bias=b2048, col_sums=cs0, out=hh, a_stats=st1, count=cnt)),
    ("MLP down [512 x 2048] + resid + stats", lambda: azk.nnx_gemm_h(hh, w3, D, 4 * D, azk.TAIL_RESID, bias=b512, resid=x1f, out=x2, stats_out=st2, count=cnt)),
    ("LN + heads [256 x 512]", lambda: azk.nnx_gemm_h(x2, wh, 256, D, azk.TAIL_HEADS, bias=b256, col_sums=csh, a_stats=st2, logits=lg, values=vl, action_dim=225, count=cnt)),
]
for f in links:
    f[1]()
tot = 0.0
for name, f in links:
    t = timeit(f)
    tot += t
    print(f"{name:42s} {t:7.2f} us")
print(f"{'sum':42s} {tot:7.2f} us   ({live} live rows)")
