"""Micro-driver: average k_tree duration under the bench workload shape.  usage: run_tree.py [G] [sims] [moves]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "alpha-zero_amd"), os.path.join(ROOT, "tests")]
import torch
import azk
from fixture_eval import fixture_logits_value

G = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
sims = int(sys.argv[2]) if len(sys.argv) > 2 else 800
moves = int(sys.argv[3]) if len(sys.argv) > 3 else 6
eng = azk.Engine("gomoku", G, sims, size=15, arena_nodes=int(os.environ.get("ARENA", 0)))
eng.reset_games()
lg = torch.randn(G, 225, device="cuda") * 0.05
vl = torch.tanh(torch.randn(G, device="cuda") * 0.1)
pairs = []
for mv in range(moves):
    noise, uni = eng.gen_noise(0, 0, mv)
    eng.begin_search(noise)
    first = True
    for s in range(sims):
        timed = mv >= moves - 2 and s % 8 == 0
        if timed:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
        eng.step_tree(None if first else lg, None if first else vl)
        if timed:
            b.record(); pairs.append((a, b))
        eng.step_gather()
        first = False
    eng.step_expand_backup(lg, vl)
    eng.root_stats(); eng.advance(uni, 8)
torch.cuda.synchronize()
print(f"k_tree mean {sum(a.elapsed_time(b) for a, b in pairs) / len(pairs) * 1e3:.1f} us over {len(pairs)} samples; counters {eng.counters()}")

import ctypes as C, numpy as np
out = np.zeros(8, np.int64)
azk.lib().azk_debug_stamps(eng.h, out.ctypes.data_as(C.c_void_p))
if out[6]:
    n = out[6]
    print("stamps per leaf-sim (cycles): expand %.0f | board+root %.0f | walk %.0f | terminal %.0f | moves %.0f | writes %.0f | depth %.2f" % (
        out[0] / n, out[1] / n, out[2] / n, out[3] / n, out[4] / n, 0, out[5] / n))
    print("slowest simulation per game: %s %.0f cycles (AZK_STAMP_MAX=1: max over games, else mean over games)" % (
        "max" if os.environ.get("AZK_STAMP_MAX") else "mean", out[7] if os.environ.get("AZK_STAMP_MAX") else out[7] / G))

if os.environ.get("AZK_TREE_ABLATE") == "1024":
    n = out[6]
    print("expansion per leaf (cycles): entry loads %.0f | logits/noise/header loads %.0f | exp %.0f | pairwise sum %.0f | children + backup %.0f" % (
        out[0] / n, out[1] / n, out[2] / n, out[3] / n, out[4] / n))
if os.environ.get("AZK_TREE_ABLATE") == "64":
    n = out[6]
    print("walk per simulation (cycles): load wait %.0f | loads+ucb %.0f (incl. the wait) | argmax+readlanes %.0f | path/board update %.0f | levels %.2f" % (
        out[0] / n, out[1] / n, out[2] / n, out[3] / n, out[5] / n))
if os.environ.get("AZK_TREE_ABLATE") == "32":
    n = max(1, eng.counters()["leaves_evaluated"] + eng.counters()["cache_hits"])      # one call per non-terminal simulation
    print("valid_moves sub-phases (cycles per call): keys %.0f | prefix+rank %.0f | inserts %.0f | emit %.0f | candidates %.1f" % (
        out[0] / n, out[1] / n, out[2] / n, out[3] / n, out[4] / n))
