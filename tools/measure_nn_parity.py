"""How far does the benched evaluator (bf16, path 'clsfold': hand-written kernels) move the search results away from the
reference's fp32 network?  (north_star: "within 1e-5 on visit-count policies at a fixed RNG seed")
  1. logits / value of every evaluator path against the reference's seed-0 known answers (tests/golden/nn_small.npz 'full_*');
  2. 800-simulation searches from the positions of the reference's recorded 15x15 games, same Dirichlet noise, evaluator =
     the seed-0 network in (a) fp32 'full' (the reference's own arithmetic), (b) fp32 'cls' (same function, different
     summation order: the sensitivity floor), (c) fp32 'clsfold' (the hand-written fp32-accurate kernels, csrc/azk_nnx.hip),
     (d) bf16 'clsfold' (the benched path): max |delta pi|, total variation, share of
     positions whose most-visited move changes.
usage: measure_nn_parity.py [n_sims]      -> one JSON line"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "alpha-zero_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch

import azk
from conftest import golden_meta, load_golden
from pvnet import NetConfig, PolicyValueNet


def golden_positions():
    """(cells int8 [225], to_move, move_count) of every recorded ply of the reference's 15x15 games (tests/golden/games.npz)."""
    z = load_golden("games.npz")
    out = []
    for m in golden_meta(z):
        if m["size"] != 15:
            continue
        cells = z[f"g{m['game']}_board_cells"]
        for ply in range(len(cells)):
            out.append((cells[ply].astype(np.int8).reshape(-1), ply & 1, ply))
    return out


def search_pis(net, leaf_dtype, positions, n_sims, noise):
    G = len(positions)
    eng = azk.Engine("gomoku", G, n_sims, size=15, leaf_dtype=leaf_dtype)
    eng.reset_games()
    eng.set_positions(np.stack([p[0] for p in positions]), [p[1] for p in positions], [p[2] for p in positions])
    eng.search(net, n_sims, noise)
    eng.check_error()
    pi, q, _ = eng.root_stats()
    out = pi.cpu().numpy().copy(), q.cpu().numpy().copy()
    eng.close()
    return out


def main():
    n_sims = int(sys.argv[1]) if len(sys.argv) > 1 else 800
    cfg = NetConfig(15, 15, 2, 225, 5, 512, 8, 1)
    z = load_golden("nn_small.npz")
    x = torch.from_numpy(z["full_x"]).cuda()
    nets = {"fp32_full": PolicyValueNet(cfg, seed=0, device="cuda", dtype=torch.float32, path="full"),
            "fp32_cls": PolicyValueNet(cfg, seed=0, device="cuda", dtype=torch.float32, path="cls"),
            "bf16_full": PolicyValueNet(cfg, seed=0, device="cuda", dtype=torch.bfloat16, path="full"),
            "bf16_cls": PolicyValueNet(cfg, seed=0, device="cuda", dtype=torch.bfloat16, path="cls"),
            "bf16_clsfold": PolicyValueNet(cfg, seed=0, device="cuda", dtype=torch.bfloat16, path="clsfold"),
            "fp32_clsfold": PolicyValueNet(cfg, seed=0, device="cuda", dtype=torch.float32, path="clsfold")}     # hand-written fp32-accurate kernels (azk_nnx.hip)
    assert nets["fp32_clsfold"]._exact is not None
    kat = {}
    for name, net in nets.items():
        logits, v = net(x.to(net.dtype))
        dl = np.abs(logits.float().cpu().numpy() - z["full_logits"])
        dv = np.abs(v.float().cpu().numpy().reshape(-1) - z["full_value"].reshape(-1))
        p, pr = torch.softmax(logits.float(), 1).cpu().numpy(), torch.softmax(torch.from_numpy(z["full_logits"]), 1).numpy()
        kat[name] = {"logits_max_abs": float(dl.max()), "logits_mean_abs": float(dl.mean()), "value_max_abs": float(dv.max()),
                     "policy_tv_max": float(0.5 * np.abs(p - pr).sum(1).max())}
    # a wider sample for the error budget: benchmark-like boards, bf16 clsfold against the fp32 full forward
    rng = np.random.RandomState(0)
    xb = np.zeros((256, 2, 15, 15), np.float32)
    for b in range(256):
        cells = [(7, 7)]
        for _ in range(rng.randint(0, 40)):
            r, c = cells[rng.randint(len(cells))]
            cells.append((int(np.clip(r + rng.randint(-1, 2), 0, 14)), int(np.clip(c + rng.randint(-1, 2), 0, 14))))
        for i, (r, c) in enumerate(dict.fromkeys(cells)):
            xb[b, i & 1, r, c] = 1
    xb = torch.from_numpy(xb).cuda()
    l32, v32 = nets["fp32_full"](xb)
    l16, v16 = nets["bf16_clsfold"](xb.to(torch.bfloat16))
    lx, vx = nets["fp32_clsfold"](xb)
    wide_x = {"boards": 256, "logits_max_abs": float((lx - l32).abs().max()), "value_max_abs": float((vx.reshape(-1) - v32.reshape(-1)).abs().max())}
    wide = {"boards": 256, "logits_max_abs": float((l16 - l32).abs().max()), "logits_mean_abs": float((l16 - l32).abs().mean()),
            "logits_std_of_reference": float(l32.std()), "value_max_abs": float((v16.reshape(-1) - v32.reshape(-1)).abs().max()),
            "policy_tv_max": float(0.5 * (torch.softmax(l16, 1) - torch.softmax(l32, 1)).abs().sum(1).max())}

    positions = golden_positions()
    G = len(positions)
    noise = torch.from_numpy(np.random.RandomState(7).dirichlet([0.03] * 225, size=G)).cuda()
    pis = {}
    for name, dt in (("fp32_full", "float32"), ("fp32_cls", "float32"), ("fp32_clsfold", "float32"), ("bf16_clsfold", "bfloat16")):
        pis[name] = search_pis(nets[name], dt, positions, n_sims, noise)
    again = search_pis(nets["bf16_clsfold"], "bfloat16", positions, n_sims, noise)
    ref_pi, ref_q = pis["fp32_full"]

    def dev(name):
        pi, q = pis[name]
        d = np.abs(pi - ref_pi)
        return {"max_abs_dpi": float(d.max()), "mean_over_positions_of_max_abs_dpi": float(d.max(1).mean()),
                "tv_mean": float(0.5 * d.sum(1).mean()), "tv_max": float(0.5 * d.sum(1).max()),
                "argmax_changed_share": float((pi.argmax(1) != ref_pi.argmax(1)).mean()),
                "positions_with_identical_pi": int((d.max(1) == 0).sum()), "max_abs_dq": float(np.abs(q - ref_q).max())}
    out = {"n_sims": n_sims, "positions": G, "kat_vs_reference_seed0": kat, "clsfold_vs_fp32_full_on_256_boards": wide, "fp32_clsfold_vs_fp32_full_on_256_boards": wide_x,
           "search_vs_fp32_full": {"fp32_cls": dev("fp32_cls"), "fp32_clsfold": dev("fp32_clsfold"), "bf16_clsfold": dev("bf16_clsfold")},
           "bf16_clsfold_rerun_identical": bool(np.array_equal(again[0], pis["bf16_clsfold"][0]))}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
