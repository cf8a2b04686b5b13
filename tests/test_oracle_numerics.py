"""Numeric primitives of the oracle against the libraries the reference actually calls
(numpy 2.x scalar arithmetic, np.sum, np.exp, legacy np.random.choice).  CPU only."""
import ctypes as C
import math

import numpy as np

from oracle import az_oracle as ao


def test_pairwise_sum_is_numpy_sum():
    """np.sum over a contiguous float32 vector == a[0] + pairwise(a[1:]) or pairwise(a): find which
    and make sure the oracle's softmax uses the same (bit-exact for every length <= 512)."""
    rng = np.random.RandomState(0)
    L = ao.lib()
    for n in list(range(1, 300)) + [449, 450, 512]:
        a = np.exp(rng.uniform(-3, 3, n)).astype(np.float32)
        got = L.azo_pairwise_sum_f32(a.ctypes.data, n)
        assert np.float32(got).tobytes() == np.sum(a).tobytes(), n


def test_exp_det_close_to_numpy_exp():
    rng = np.random.RandomState(1)
    L = ao.lib()
    xs = np.concatenate([rng.uniform(-20, 20, 20000), rng.uniform(-3, 3, 20000), [0.0, -0.0, 1.0, -1.0, 88.0, -87.0, -100.0]]).astype(np.float32)
    ref = np.exp(xs.astype(np.float64))
    got = np.array([L.azo_exp_det(float(x)) for x in xs], np.float32)
    # within half an ulp (+tiny) of the true value, and within 4 ulp of numpy's float32 exp
    ulp = np.spacing(ref.astype(np.float32)).astype(np.float64)
    assert np.all(np.abs(got.astype(np.float64) - ref) <= 0.5000001 * ulp + 1e-300)
    npf = np.exp(xs)
    assert np.all(np.abs(got.astype(np.float64) - npf.astype(np.float64)) <= 4 * ulp)


def test_softmax_det_vs_reference_expression():
    rng = np.random.RandomState(2)
    for n in (7, 9, 49, 225):
        for _ in range(50):
            l = rng.uniform(-3, 3, n).astype(np.float32)
            ref = np.exp(l) / np.sum(np.exp(l))
            got = ao.softmax_det(l)
            np.testing.assert_allclose(got, ref, rtol=5e-7, atol=0)


def test_ucb_float32_semantics_match_numpy_scalars():
    """utils.py:29-44 evaluated with numpy>=2 scalar promotion, versus the oracle's ucb_f32 restated
    here with explicit float32 casts."""
    rng = np.random.RandomState(3)
    for _ in range(20000):
        prior = np.float32(rng.uniform(0, 0.2))
        n_parent = int(rng.randint(1, 800))
        visit = int(rng.randint(0, 50))
        value = float(np.float32(rng.uniform(-1, 1))) * visit * rng.uniform(0, 1)
        if visit == 0:
            ref = prior * math.sqrt(n_parent) / (visit + 1)
        else:
            ref = (value / visit) + prior * math.sqrt(n_parent) / (visit + 1)
        assert isinstance(ref, np.float32)
        s = np.float32(math.sqrt(n_parent))
        u = np.float32(np.float32(prior * s) / np.float32(visit + 1))
        mine = u if visit == 0 else np.float32(np.float32(value / visit) + u)
        assert mine.tobytes() == ref.tobytes()


def test_sample_action_is_legacy_choice():
    rng = np.random.RandomState(4)
    for trial in range(300):
        n = int(rng.choice([7, 9, 49, 225]))
        visits = np.zeros(n)
        k = rng.randint(1, n + 1)
        idx = rng.choice(n, k, replace=False)
        visits[idx] = rng.randint(1, 200, k)
        p = visits / np.sum(visits)
        np.random.seed(trial)
        st = np.random.get_state()
        u = np.random.random_sample()
        np.random.set_state(st)
        want = int(np.random.choice(np.arange(n), p=p))
        assert ao.sample_action(p, u) == want
