"""Micro-driver: azk_nn_embed_pool_compact against azk_nn_embed_pool on boards with a controlled number of dirty tokens.
usage: run_embed_pool_c.py [n_boards] [reps]   -> time per launch for: the full kernel; the compact kernel on empty boards
(pure per-board overhead), on boards with ~k dirty tiles, and on benchmark-like clustered boards."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "alpha-zero_amd"))
import numpy as np
import torch

import azk
from pvnet import NetConfig, PolicyValueNet

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
cfg = NetConfig(15, 15, 2, 225, 5, 512, 8, 1)
net = PolicyValueNet(cfg, seed=0, device="cuda", dtype=torch.bfloat16, path="clsfold")
f = net._fold
sched = azk.new_sched("cuda")


def boards_rows(k_rows):
    """stones in the middle of `k_rows` board rows spaced so that about 15 * (k_rows + 4) tokens are dirty"""
    x = np.zeros((n, 2, 15, 15), np.float32)
    for r in range(k_rows):
        x[:, r & 1, min(14, 2 + r), ::2] = 1
    return torch.from_numpy(x).cuda().to(torch.bfloat16).contiguous()


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
    return torch.from_numpy(x).cuda().to(torch.bfloat16).contiguous()


def dirty_tokens(x):
    occ = (x.float().sum(1, keepdim=True) > 0).float()
    d = torch.nn.functional.max_pool2d(torch.nn.functional.pad(occ, (2, 2, 2, 2)), 5, 1)
    return d.flatten(1).sum(1)


def timeit(fn):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


cases = [("empty", torch.zeros(n, 2, 15, 15, device="cuda", dtype=torch.bfloat16))] + \
        [(f"rows{k}", boards_rows(k)) for k in (1, 3, 6, 11)] + [("clustered", clustered())]
for name, x in cases:
    d = dirty_tokens(x)
    full = timeit(lambda: azk.nn_embed_pool(x, f["wt_ext"], f["cpos_frag"], f["score_frag"], f["score_msum"], f["score_ref"], 15, 15, 5, 512, 8))
    comp = timeit(lambda: azk.nn_embed_pool_compact(x, net._compact, 15, 15, sched))
    print(f"{name:10s} n={n} dirty tokens mean {d.mean().item():6.1f} tiles {torch.ceil(d / 16).mean().item():5.2f}: full {full:6.1f} us, compact {comp:6.1f} us")

if os.environ.get("AZK_EMBED_POOL_STAMPS"):
    # per-phase cycle stamps (wave 0 of every workgroup): run each case a few times, then ask the library to print and reset
    for name, x in cases:
        os.environ["AZK_EMBED_POOL_STAMPS"] = "1"
        for _ in range(5):
            azk.nn_embed_pool_compact(x, net._compact, 15, 15, sched)
        torch.cuda.synchronize()
        os.environ["AZK_EMBED_POOL_STAMPS"] = "2"
        print(name, file=sys.stderr, end=": ", flush=True)
        azk.nn_embed_pool_compact(x, net._compact, 15, 15, sched)
        torch.cuda.synchronize()
