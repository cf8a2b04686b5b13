"""k_tree: the distribution of per-wave time inside a launch (a launch lasts as long as its slowest wave), and what the slowest
wave does that the median one does not.  DBG instantiation with AZK_TREE_ABLATE=8192: every wave leaves ONE record per launch
(cycles per phase, kind of simulation, depth, legal moves, children created); read back after every `stride`-th launch of a few
eagerly stepped moves on the de-phased benchmark state.
usage: AZK_TREE_ABLATE=8192 python tools/tree_wave_dist.py [games] [sims] [moves] [cheap_preroll] [stride]     (one JSON line)"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "alpha-zero_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch

import azk
from pvnet import NetConfig, PolicyValueNet
from selfplay import SelfPlayRunner

assert int(os.environ.get("AZK_TREE_ABLATE", "0")) & 8192, "run with AZK_TREE_ABLATE=8192"
G = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
sims = int(sys.argv[2]) if len(sys.argv) > 2 else 800
moves = int(sys.argv[3]) if len(sys.argv) > 3 else 2
pre = int(sys.argv[4]) if len(sys.argv) > 4 else 128
stride = int(sys.argv[5]) if len(sys.argv) > 5 else 8

cfg = NetConfig(15, 15, 2, 225, 5, 512, 8, 1)
net = PolicyValueNet(cfg, seed=0, device="cuda:0", dtype=torch.bfloat16, path="clsfold")
runner = SelfPlayRunner("gomoku", net, G, sims, size=15, seed=0, device=0, leaf_dtype="bfloat16", recycle=True, use_graph=True,
                        cache_entries=32768, cache_shared=True)
runner.n_sims = 16
for _ in range(pre):
    runner.play_move()
runner.n_sims = sims
for _ in range(26):
    runner.play_move()

runner.use_graph = False
runner.leaf_source_ok = False
e = runner.eng
orig_eval = runner.evaluator
step = [0]
recs = []
raw = np.zeros((G, 8), np.int64)


def spy(boards):
    if step[0] % stride == 0:
        torch.cuda.synchronize()
        assert azk.lib().azk_debug_stamps_raw(e.h, raw.ctypes.data_as(C.c_void_p), G) == 0
        recs.append(raw.copy())
    step[0] += 1
    return orig_eval(boards)


runner.evaluator = spy
for _ in range(moves):
    runner.play_move()
torch.cuda.synchronize()
R = np.stack(recs)                                   # [launches, G, 8]
tot = R[:, :, 7].astype(float)
kind = (R[:, :, 6] & 0xff) - 1                       # -1 idle, 0 terminal, 1 cache hit, 2 evaluator leaf
nv = (R[:, :, 6] >> 8) & 0xfff
env = (R[:, :, 6] >> 20) & 0xfff
depth = R[:, :, 5]
live = kind >= 0
names = {0: "terminal leaf", 1: "eval-cache hit", 2: "evaluator leaf"}
out = {"launches_sampled": int(R.shape[0]), "games": G, "unit": "shader cycles per wave and launch",
       "per_launch": {"max": float(tot.max(1).mean()), "p99": float(np.percentile(tot, 99, axis=1).mean()), "p90": float(np.percentile(tot, 90, axis=1).mean()),
                      "p50": float(np.percentile(tot, 50, axis=1).mean()), "mean": float(tot.mean())},
       "by_kind": {}, "slowest_wave_of_a_launch": {}}
ph = ("expansion_of_previous_leaf", "walk", "terminal_test", "legal_moves", "leaf_writes_and_cache_probe")
for k, nm in names.items():
    m = kind == k
    if m.sum() == 0:
        continue
    out["by_kind"][nm] = {"share_of_waves": float(m.mean()), "mean_total": float(tot[m].mean()), "p90_total": float(np.percentile(tot[m], 90)),
                          "mean_depth": float(depth[m].mean()), "mean_legal_moves": float(nv[m].mean()),
                          "phases_mean": {p: float(R[:, :, i][m].mean()) for i, p in enumerate(ph)}}
am = tot.argmax(1)
ix = np.arange(R.shape[0])
mk = kind[ix, am]
out["slowest_wave_of_a_launch"] = {
    "kind_share": {names.get(int(k), "idle"): float((mk == k).mean()) for k in np.unique(mk)},
    "mean_total": float(tot[ix, am].mean()), "mean_depth": float(depth[ix, am].mean()), "mean_legal_moves": float(nv[ix, am].mean()),
    "mean_children_created_by_its_expansion": float(env[ix, am].mean()),
    "phases_mean": {p: float(R[ix, am, i].mean()) for i, p in enumerate(ph)}}
med = np.abs(tot - np.percentile(tot, 50, axis=1)[:, None]).argmin(1)
out["median_wave_of_a_launch"] = {"mean_total": float(tot[ix, med].mean()), "mean_depth": float(depth[ix, med].mean()),
                                  "mean_legal_moves": float(nv[ix, med].mean()), "mean_children_created_by_its_expansion": float(env[ix, med].mean()),
                                  "phases_mean": {p: float(R[ix, med, i].mean()) for i, p in enumerate(ph)}}
# the ten slowest waves of every launch: how far the maximum sits above them (is it one straggler or a crowd?)
srt = np.sort(tot, axis=1)
out["top_of_the_launch"] = {f"rank_{r}": float(srt[:, -r].mean()) for r in (1, 2, 4, 8, 16, 32, 64, 128, 256)}
print(json.dumps(out))
