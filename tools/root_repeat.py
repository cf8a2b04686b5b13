"""How often would a simulation take the same root child as the one before it?  The iid estimate sum_i p_i^2 over the root's visit shares
(utils.get_probablity_distribution_of_children), averaged over the games of the benchmark state - it prices speculation below the walk's
first level.  usage: python tools/root_repeat.py [games] [sims]     (one JSON line)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "alpha-zero_amd"), os.path.join(ROOT, "tests")]
import torch

from pvnet import NetConfig, PolicyValueNet
from selfplay import SelfPlayRunner

G = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
sims = int(sys.argv[2]) if len(sys.argv) > 2 else 800
cfg = NetConfig(15, 15, 2, 225, 5, 512, 8, 1)
net = PolicyValueNet(cfg, seed=0, device="cuda:0", dtype=torch.bfloat16, path="clsfold")
runner = SelfPlayRunner("gomoku", net, G, sims, size=15, seed=0, device=0, leaf_dtype="bfloat16", recycle=True, use_graph=True,
                        cache_entries=32768, cache_shared=True)
runner.n_sims = 16
for _ in range(128):
    runner.play_move()
runner.n_sims = sims
acc, top, n = 0.0, 0.0, 0
seen = []
runner.on_records = lambda mv, lo, pi, q, chosen, winner, done: seen.append(pi.clone())
for _ in range(6):
    runner.play_move()
torch.cuda.synchronize()
for pi in seen[2:]:
    p = pi.double()
    s = p.sum(1)
    ok = s > 0
    p = p[ok] / s[ok, None]
    acc += float((p * p).sum(1).sum()); top += float(p.max(1).values.sum()); n += int(ok.sum())
print(json.dumps({"games_x_moves": n, "sims": sims, "mean_sum_p_squared": acc / max(n, 1), "mean_top_child_share": top / max(n, 1),
                  "reading": "probability that two independent draws from the root's visit distribution pick the same child"}))
