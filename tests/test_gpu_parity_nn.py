"""Real-network parity: what the benched evaluator - bf16, path 'clsfold', hand-written kernels - does to
the search results, measured against the reference's own fp32 arithmetic on the positions of the reference's recorded 15x15 games
(tools/measure_nn_parity.py holds the measurement; numbers quoted in DESIGN.md section 2).

north_star asks for visit-count policies within 1e-5 at a fixed seed.  That bar is met - exactly, delta = 0 - whenever the
evaluator's outputs are the reference's (fixture evaluators everywhere else in the suite; the fp32 network paths here).  The
bf16 evaluator moves logits by <= 1e-2, which changes a handful of visits in a few positions: the bound asserted for it is
the measured one with margin, stated in the test, and bench.py prints an fp32-evaluator line next to the bf16 one."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from pvnet import NetConfig, PolicyValueNet

pytestmark = pytest.mark.gpu


def test_benched_evaluator_against_reference_known_answers():
    """Every evaluator path against the reference's seed-0 outputs (nn_small.npz 'full_*': Net(15, 5, 512, 225, 8, 1, 2) under
    torch.manual_seed(0)).  Error budget of the bf16 paths: weights and activations rounded to 8 significant bits (2^-9 relative)
    through ~6 dependent GEMM / normalisation stages on O(1) values, logits std 0.6: measured 6.5e-3 (clsfold) - 8.9e-3 (full)
    max abs, asserted at 2e-2 as SURVEY 8(c) prescribes; fp32 paths at 1e-5."""
    z = load_golden("nn_small.npz")
    cfg = NetConfig(15, 15, 2, 225, 5, 512, 8, 1)
    x = torch.from_numpy(z["full_x"]).cuda()
    for path, dtype, tol_l, tol_v in (("full", torch.float32, 1e-5, 1e-6), ("cls", torch.float32, 1e-5, 1e-6),
                                      ("full", torch.bfloat16, 2e-2, 2e-3), ("cls", torch.bfloat16, 2e-2, 2e-3),
                                      ("clsfold", torch.bfloat16, 2e-2, 2e-3)):
        net = PolicyValueNet(cfg, seed=0, device="cuda", dtype=dtype, path=path)
        logits, v = net(x.to(dtype))
        np.testing.assert_allclose(logits.float().cpu().numpy(), z["full_logits"], rtol=0, atol=tol_l, err_msg=f"{path} {dtype}")
        np.testing.assert_allclose(v.float().cpu().numpy().reshape(-1), z["full_value"].reshape(-1), rtol=0, atol=tol_v, err_msg=f"{path} {dtype}")
        p = torch.softmax(logits.float(), 1).cpu().numpy()
        pr = torch.softmax(torch.from_numpy(z["full_logits"]), 1).numpy()
        assert 0.5 * np.abs(p - pr).sum(1).max() < (1e-6 if dtype == torch.float32 else 3e-3)


def test_search_policies_fp32_exact_and_bf16_bounded():
    """800-simulation searches from all 92 recorded positions of the reference's 15x15 games, same Dirichlet noise:
      fp32 'cls' vs fp32 'full' (same function, different summation order): pi identical to the last visit (the 1e-5 bar);
      bf16 'clsfold' vs fp32 'full': measured 88 / 92 positions identical (round 2's conv-form embedding kernel: 86), max |delta pi| 3.8e-3 (3 visits of 799), mean total
      variation 1.4e-4, no position changes its most-visited move.  Asserted with margin: >= 70 % identical, max |delta pi|
      <= 2e-2, mean TV <= 1e-3, most-visited move changed in <= 5 % of the positions.  And the bf16 search is deterministic."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from measure_nn_parity import golden_positions, search_pis
    cfg = NetConfig(15, 15, 2, 225, 5, 512, 8, 1)
    positions = golden_positions()
    assert len(positions) >= 32
    G = len(positions)
    noise = torch.from_numpy(np.random.RandomState(7).dirichlet([0.03] * 225, size=G)).cuda()
    ref_pi, ref_q = search_pis(PolicyValueNet(cfg, seed=0, device="cuda", dtype=torch.float32, path="full"), "float32", positions, 800, noise)
    assert np.allclose(ref_pi.sum(1), 1.0)
    cls_pi, _ = search_pis(PolicyValueNet(cfg, seed=0, device="cuda", dtype=torch.float32, path="cls"), "float32", positions, 800, noise)
    assert np.abs(cls_pi - ref_pi).max() <= 1e-5                                  # north_star's bar, met by the fp32 evaluator
    net16 = PolicyValueNet(cfg, seed=0, device="cuda", dtype=torch.bfloat16, path="clsfold")
    pi16, q16 = search_pis(net16, "bfloat16", positions, 800, noise)
    d = np.abs(pi16 - ref_pi)
    identical = int((d.max(1) == 0).sum())
    print(f"bf16 clsfold vs fp32 full: {identical}/{G} identical, max |dpi| {d.max():.2e}, mean TV {0.5 * d.sum(1).mean():.2e}, "
          f"argmax changed {(pi16.argmax(1) != ref_pi.argmax(1)).mean():.3f}")
    assert identical >= 0.7 * G
    assert d.max() <= 2e-2 and 0.5 * d.sum(1).mean() <= 1e-3
    assert (pi16.argmax(1) != ref_pi.argmax(1)).mean() <= 0.05
    again, _ = search_pis(net16, "bfloat16", positions, 800, noise)
    assert np.array_equal(again, pi16)
