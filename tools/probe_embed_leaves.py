import sys, os, time, json
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "alpha-zero_amd"))
import torch, azk
from pvnet import NetConfig, PolicyValueNet
cfg = NetConfig(15, 15, 2, 225, 5, 512, 8, 1)
net = PolicyValueNet(cfg, seed=0, device="cuda", dtype=torch.bfloat16, path="clsfold")
G, A = 2048, 225
eng = azk.Engine("gomoku", G, 800, size=15, leaf_dtype="bfloat16", cache_entries=256)
eng.reset_games()
noise, uni = eng.gen_noise(3, 0, 0)
eng.begin_search(noise)
logits = values = None
for s in range(40):
    eng.step_tree(logits, values)
    eng.step_gather()
    logits, values = torch.randn(G, A, device="cuda") * 0.3, torch.tanh(torch.randn(G, device="cuda"))
src = eng.leaf_source()
sched = azk.new_sched("cuda")
torch.cuda.synchronize()
def timed(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    res = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); a.record(); fn(); b.record(); t1 = time.perf_counter()
        torch.cuda.synchronize()
        res.append((a.elapsed_time(b) * 1e3, (t1 - t0) * 1e6))
        time.sleep(0.002)
    res.sort()
    return {"event_us_median": res[len(res)//2][0], "event_us_max": res[-1][0], "host_us_median": sorted(r[1] for r in res)[len(res)//2]}
print("n_leaf", int(eng.n_leaf.item()))
print(json.dumps({"fold_leaves": timed(lambda: azk.nn_embed_fold_leaves(src, net._foldu, sched)),
                  "compact_leaves": timed(lambda: azk.nn_embed_pool_compact_leaves(src, net._compact, sched))}, indent=1))
