"""Deterministic fixture evaluators (TEST INFRASTRUCTURE, not product code).

A fixture evaluator stands in for the policy-value network in tree-search
parity tests: it maps a canonical board batch to (logits, value) using exact
integer arithmetic followed by exactly-representable float32 scaling, so the
very same numbers come out of numpy/torch on the CPU (fixture generation with
the reference, the C oracle's callback) and torch on the GPU (the HIP engine).

Variants
  "hash"    - logits in [-3, 3) on a 2^-16*6 grid, pseudo-random per (board, action)
  "uniform" - all logits 0 (every prior ties: stresses child-order / first-max rules)
Both variants share the same hash-derived value in [-1, 1).
"""
import numpy as np
import torch


def fixture_logits_value(x, action_dim, variant="hash"):
    """x: [n, F, R, C] tensor holding 0/1 (any float/int dtype, any device).

    Returns (logits [n, A] float32, value [n] float32) on x's device.
    Every intermediate is an int64 < 2^40 and every float op is exact.
    """
    n = x.shape[0]
    cells = x.reshape(n, -1).to(torch.int64)
    L = cells.shape[1]
    dev = x.device
    idx = torch.arange(1, L + 1, dtype=torch.int64, device=dev)
    w = (idx * 1000003) % 1048576
    h = (cells * w).sum(dim=1)                                  # < 2^29
    a = torch.arange(action_dim, dtype=torch.int64, device=dev)
    if variant == "uniform":
        logits = torch.zeros((n, action_dim), dtype=torch.float32, device=dev)
    elif variant == "hash":
        t = (h[:, None] * 31 + a[None, :] * 7919 + (h[:, None] >> 3) * a[None, :]) % 65536
        logits = (t * 6 - 3 * 65536).to(torch.float32) * (1.0 / 65536.0)
    else:
        raise ValueError(variant)
    v = (((h * 17 + 5) % 65536) - 32768).to(torch.float32) * (1.0 / 32768.0)
    return logits, v


class FixtureModel:
    """Callable with the reference's model signature:
    model(tensor[1,F,R,C]) -> (logits[1,A], value[1,1])   (ai/mcts.py:46)."""

    def __init__(self, action_dim, variant="hash"):
        self.action_dim = action_dim
        self.variant = variant
        self.calls = 0

    def __call__(self, x):
        self.calls += x.shape[0]
        logits, v = fixture_logits_value(x, self.action_dim, self.variant)
        return logits, v[:, None]

    def __bool__(self):  # the reference tests `if model:`
        return True

    def eval(self):      # test.compare calls model.eval() (test.py:108-111)
        return self


def numpy_softmax_like_reference(logits_f32):
    """The reference's softmax expression, verbatim semantics (ai/mcts.py:48-49):
    float32, no max-subtraction, numpy exp + numpy pairwise sum."""
    l = np.asarray(logits_f32, dtype=np.float32)
    return np.exp(l) / np.sum(np.exp(l))
