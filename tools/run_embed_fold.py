#!/usr/bin/env python3
"""k_embed_fold (+ the batched GEMM behind it) against k_embed_pool_c (+ the value projection) on benchmark-shaped boards:
launch time by HIP events, per batch size.   python tools/run_embed_fold.py [n ...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "alpha-zero_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import azk
from pvnet import NetConfig, PolicyValueNet


def boards(n, seed):
    rng = np.random.RandomState(seed)
    x = np.zeros((n, 2, 15, 15), np.float32)
    for b in range(n):
        k = rng.randint(8, 45)
        cells = [(7, 7)]
        for _ in range(k):
            r, c = cells[rng.randint(len(cells))]
            cells.append((int(np.clip(r + rng.randint(-1, 2), 0, 14)), int(np.clip(c + rng.randint(-1, 2), 0, 14))))
        for i, (r, c) in enumerate(dict.fromkeys(cells)):
            x[b, i & 1, r, c] = 1
    return torch.from_numpy(x)


def timed(fn, reps=40):
    for _ in range(5):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return t[len(t) // 2]


cfg = NetConfig(15, 15, 2, 225, 5, 512, 8, 1)
net = PolicyValueNet(cfg, seed=0, device="cuda", dtype=torch.bfloat16, path="clsfold")
sched = azk.new_sched("cuda")
out = {}
for n in [int(a) for a in sys.argv[1:]] or [457, 914, 1400, 2048]:
    x = boards(n, 3).cuda().to(torch.bfloat16).contiguous()
    u = torch.empty(n, 512, dtype=torch.bfloat16, device="cuda")
    rows = azk.nn_embed_fold(x, net._foldu, 15, 15, sched)
    z = azk.nn_embed_pool_compact(x, net._compact, 15, 15, sched)
    H, ROW, f = 8, azk.EMBED_FOLD_ROW, net._fold
    out[n] = {"k_embed_fold_us": timed(lambda: azk.nn_embed_fold(x, net._foldu, 15, 15, sched)),
              "fold_gemm_k384_us": timed(lambda: azk.nn_tail_gemm(rows.view(n, H * ROW), net._foldu.weight, 64, ROW, azk.TAIL_BF16, nbatch=H, a_batch_stride=ROW, out=u)),
              "k_embed_pool_c_us": timed(lambda: azk.nn_embed_pool_compact(x, net._compact, 15, 15, sched)),
              "value_projection_k512_us": timed(lambda: azk.nn_tail_gemm(z.view(n, H * 512), f["WvHP"], 64, 512, azk.TAIL_BF16, nbatch=H, a_batch_stride=512, out=u))}
print(json.dumps(out, indent=1))
