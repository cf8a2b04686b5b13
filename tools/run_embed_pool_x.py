"""Micro-driver: azk_nnx_embed_pool (fp32-accurate embedding + pooling) on synthetic boards: time per launch by board content
(empty = pure per-board overhead; k stone rows = a controlled number of dirty 16-token tiles; benchmark-like clusters).
usage: run_embed_pool_x.py [n_boards] [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "alpha-zero_amd"))
import numpy as np
import torch

import azk
from pvnet import NetConfig, PolicyValueNet

n = int(sys.argv[1]) if len(sys.argv) > 1 else 918
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
cfg = NetConfig(15, 15, 2, 225, 5, 512, 8, 1)
net = PolicyValueNet(cfg, seed=0, device="cuda", dtype=torch.float32, path="clsfold")
tb = net._exact["tables"]
sched = azk.new_sched("cuda")
stats = tb.enable_work_stats()


def boards_rows(k_rows):
    x = np.zeros((n, 2, 15, 15), np.float32)
    for r in range(k_rows):
        x[:, r & 1, min(14, 2 + r), ::2] = 1
    return torch.from_numpy(x).cuda().contiguous()


def clustered(seed=0):
    rng = np.random.RandomState(seed)
    x = np.zeros((n, 2, 15, 15), np.float32)
    for b in range(n):
        k = rng.randint(4, 36)
        cells = [(7, 7)]
        for _ in range(k):
            r, c = cells[rng.randint(len(cells))]
            cells.append((int(np.clip(r + rng.randint(-1, 2), 0, 14)), int(np.clip(c + rng.randint(-1, 2), 0, 14))))
        for i, (r, c) in enumerate(dict.fromkeys(cells)):
            x[b, i & 1, r, c] = 1
    return torch.from_numpy(x).cuda().contiguous()


def timeit(fn):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for name, x in [("empty", boards_rows(0))] + [(f"rows{k}", boards_rows(k)) for k in (1, 3, 6, 11)] + [("clustered", clustered())]:
    stats.zero_()
    t = timeit(lambda: azk.nnx_embed_pool(x, tb, 15, 15, sched))
    b_, t_ = (int(v) for v in stats.tolist())
    print(f"{name:10s} {t:8.1f} us per launch   tiles/board {t_ / max(1, b_):5.2f}   us per (tile-round of 4) per WG: "
          f"{t / max(1e-9, (n / 256) * np.ceil(t_ / max(1, b_) / 4)):6.2f}")
