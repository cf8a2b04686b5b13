"""Micro-driver: fused azk_nn_embed_pool vs the two-launch path (same operands).  usage: run_embed_pool.py [n] [live] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "alpha-zero_amd"))
import torch
import azk
from pvnet import NetConfig, PolicyValueNet

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
live = int(sys.argv[2]) if len(sys.argv) > 2 else n
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
cfg = NetConfig(15, 15, 2, 225, 5, 512, 8, 1)
net = PolicyValueNet(cfg, seed=0, device="cuda", dtype=torch.bfloat16, path="clsfold")
g = torch.Generator(device="cuda").manual_seed(1)
u = torch.rand(n, 15, 15, device="cuda", generator=g)
dens = torch.rand(n, 1, 1, device="cuda", generator=g) * 0.6
x = torch.stack([(u < dens / 2), (u >= dens / 2) & (u < dens)], 1).to(torch.bfloat16).contiguous()
cnt = torch.tensor([live], dtype=torch.int32, device="cuda")
f, hp = net._fold, net._hip
two = lambda: azk.nn_embed_scores_pool(x, f["wt_ext"], hp["cpos"], f["score_cpos"], f["score_msum"], f["c_n"], 15, 15, 5, 512, 8, count=cnt)
sref = None if os.environ.get("NO_STATIC_REF") else f["score_ref"]
print("static softmax reference:", None if sref is None else [round(v, 2) for v in sref[:8].tolist()])
one = lambda: azk.nn_embed_pool(x, f["wt_ext"], f["cpos_frag"], f["score_frag"], f["score_msum"], sref, 15, 15, 5, 512, 8, count=cnt)
z2, z1 = two()[:live].float(), one()[:live].float()
torch.cuda.synchronize()
d = (z1 - z2).abs()
# plain fp32 reference of the same folded computation (weights as the kernels see them: bf16-rounded)
import torch.nn.functional as F
m = min(live, 256)
cols = F.unfold(x[:m].float(), kernel_size=5, padding=2).transpose(1, 2)                 # [m, 225, 50]
W = f["wt_ext"][:512, :50].float()
tok = torch.cat([torch.zeros(m, 1, 512, device="cuda"), cols @ W.t()], 1) + hp["cpos"]
xn = F.layer_norm(tok, (512,))
sc = xn @ f["m_n"].t()                                                                    # [m, 226, 8]
ref = torch.einsum("bth,btd->bhd", torch.softmax(sc, 1), xn)
for nm, zz in (("two-launch", z2), ("fused", z1)):
    e = (zz[:m] - ref).abs()
    print(f"{nm} vs fp32: max {e.max().item():.4g} mean {e.mean().item():.4g}")
print(f"fused vs two-launch: max abs diff {d.max().item():.4g}, mean {d.mean().item():.4g}, ref mean abs {z2.abs().mean().item():.4g}")
for name, fn in (("two-launch", two), ("fused", one)):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    print(f"{name} n={n} live={live}: {a.elapsed_time(b) / reps * 1e3:.1f} us")
