"""Diagnosis (stamps build only: make -C alpha-zero_amd/csrc EXTRA=-DAZK_EP_STAMPS): per-phase cycles of k_embed_pool_c under the
benchmark workload, and the busiest workgroup of a launch.  usage: AZK_EMBED_POOL_STAMPS=1 python tools/ep_stamps.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "alpha-zero_amd")]
import torch
from pvnet import NetConfig, PolicyValueNet
from selfplay import SelfPlayRunner
cfg = NetConfig(15, 15, 2, 225, 5, 512, 8, 1)
net = PolicyValueNet(cfg, seed=0, device="cuda:0", dtype=torch.bfloat16, path="clsfold")
os.environ["AZK_EMBED_POOL_STAMPS"] = "0"
r = SelfPlayRunner("gomoku", net, 2048, 800, size=15, seed=0, device=0, leaf_dtype="bfloat16", recycle=True, use_graph=False, cache_entries=32768, cache_shared=True)
r.n_sims = 16
for _ in range(96):
    r.play_move()
r.n_sims = 200
os.environ["AZK_EMBED_POOL_STAMPS"] = "1"
r.play_move()
torch.cuda.synchronize()
os.environ["AZK_EMBED_POOL_STAMPS"] = "2"
r.n_sims = 2
r.play_move()
torch.cuda.synchronize()
print("launches with stamps: ~200; leaves:", r.counters()["leaves_evaluated"])
