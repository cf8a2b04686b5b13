"""What the evaluator batch looks like under the benchmark workload (Gomoku 15x15, 2048 games, 800 sims/move, real net):
  * duplicate leaves per simulation step (identical canonical boards among the ~1100 leaves of one step) - the in-batch
    dedupe opportunity of VERDICT r01 item 6 (ai/mcts.py:38-51: the reference's cache is process-global);
  * "dirty" tokens per leaf (tokens whose 5x5 patch holds a stone) - the constant-token skipping opportunity of item 1(c);
  * with AZK_TREE_ABLATE=16/32/64/1024 in the environment: k_tree's per-phase cycle stamps on the same de-phased state.
usage: measure_leaves.py [games] [sims] [measured_moves] [cheap_preroll_moves]     (writes one JSON line)
LEAF_CACHE=shared (default, the benchmark's cross-game cache) | game (one table per game, the round-1 engine)"""
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

G = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
sims = int(sys.argv[2]) if len(sys.argv) > 2 else 800
moves = int(sys.argv[3]) if len(sys.argv) > 3 else 2
pre = int(sys.argv[4]) if len(sys.argv) > 4 else 96
stride = int(os.environ.get("LEAF_STRIDE", 8))

cfg = NetConfig(15, 15, 2, 225, 5, 512, 8, 1)
net = PolicyValueNet(cfg, seed=0, device="cuda:0", dtype=torch.bfloat16, path="clsfold")
runner = SelfPlayRunner("gomoku", net, G, sims, size=15, seed=0, device=0, leaf_dtype="bfloat16", recycle=True, use_graph=True,
                        cache_entries=32768, cache_shared=os.environ.get("LEAF_CACHE", "shared") == "shared")
runner.n_sims = 16
for _ in range(pre):
    runner.play_move()
runner.n_sims = sims
for _ in range(3):
    runner.play_move()
runner.reset_counters()

# measured moves: eager stepping with a host sync per simulation, the gathered canonical boards inspected every `stride` sims
runner.use_graph = False
runner.leaf_source_ok = False
e = runner.eng
dup, tot, dirty_hist, nleaf = 0, 0, [], []
pad = torch.nn.functional.pad
orig_eval = runner.evaluator
sim_counter = [0]


def spy(boards):
    n = boards.shape[0]
    if sim_counter[0] % stride == 0 and n > 0:
        b = boards.view(torch.int16).reshape(n, -1)
        uniq = torch.unique(b, dim=0).shape[0]
        nonlocal_stats.append((n, uniq))
        occ = (boards.float().sum(1) > 0).float()[:, None]                       # [n,1,15,15]
        d = torch.nn.functional.max_pool2d(pad(occ, (2, 2, 2, 2)), 5, 1)[:, 0]    # stone within Chebyshev distance 2
        dirty_hist.append(d.flatten(1).sum(1).cpu().numpy())
    sim_counter[0] += 1
    return orig_eval(boards)


nonlocal_stats = []
runner.evaluator = spy
for a in ("fused_embed_pool", "live_count", "fast_outputs", "kernel_timers", "out_buffers", "leaf_source"):
    pass
for _ in range(moves):
    runner.play_move()
torch.cuda.synchronize()
n_arr = np.array([x[0] for x in nonlocal_stats], float)
u_arr = np.array([x[1] for x in nonlocal_stats], float)
dirty = np.concatenate(dirty_hist) if dirty_hist else np.zeros(1)
out = {"games": G, "sims": sims, "measured_moves": moves, "sampled_steps": len(nonlocal_stats),
       "mean_leaves_per_step": float(n_arr.mean()), "mean_unique_per_step": float(u_arr.mean()),
       "leaves_per_step_percentiles": {f"p{q}": float(np.percentile(n_arr, q)) for q in (0, 1, 5, 25, 50, 75, 95, 99, 100)},
       "share_of_steps_above": {str(th): float((n_arr > th).mean()) for th in (512, 768, 1024, 1280, 1536)},
       "leaves_by_simulation_index": {f"{lo}-{lo + len(c) * stride - 1}": float(np.mean(c)) for lo, c in
                                      ((i * stride, n_arr[i:i + max(1, 100 // stride)]) for i in range(0, min(len(n_arr), sims // stride), max(1, 100 // stride)))},
       "duplicate_fraction": float(1.0 - u_arr.sum() / n_arr.sum()),
       "dirty_tokens_per_leaf": {"mean": float(dirty.mean()), "p50": float(np.median(dirty)), "p90": float(np.percentile(dirty, 90)),
                                 "max": float(dirty.max()), "mean_tiles_of_16": float(np.ceil(dirty / 16).mean())},
       "counters": runner.counters()}
import ctypes as C
st = np.zeros(8, np.int64)
azk.lib().azk_debug_stamps(e.h, st.ctypes.data_as(C.c_void_p))
out["tree_stamps"] = {"ablate": os.environ.get("AZK_TREE_ABLATE"), "raw": st.tolist()}
print(json.dumps(out))
