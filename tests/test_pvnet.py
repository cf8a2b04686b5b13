"""The functional ViT evaluator against the reference's Net (ai/nn.py) outputs recorded in
tests/golden/nn_small.npz (weights committed as data), and the seed-0 full training config.  CPU, float32."""
import json

import numpy as np
import pytest
import torch

from conftest import load_golden
from pvnet import NetConfig, PolicyValueNet, init_weights, reference_key_shapes

Z = load_golden("nn_small.npz")


@pytest.mark.parametrize("ni", [0, 1])
def test_small_nets_match_reference_outputs(ni):
    k = f"n{ni}_"
    c = json.loads(bytes(Z[k + "cfg_json"]).decode())
    cfg = NetConfig(c["img_size"], c["img_size"], c["channels"], c["action_dim"], c["patch_size"], c["embed_dim"],
                    c["num_heads"], c["depth"])
    sd = {key[len(k) + 3:]: torch.from_numpy(Z[key]) for key in Z.files if key.startswith(k + "sd_")}
    assert set(sd) == set(reference_key_shapes(cfg))
    x = torch.from_numpy(Z[k + "x"])
    for path in ("full", "cls"):
        net = PolicyValueNet.from_state_dict(cfg, sd, path=path)
        logits, v = net(x)
        np.testing.assert_allclose(logits.numpy(), Z[k + "logits"], rtol=0, atol=2e-5)   # float32: 1e-5-level
        np.testing.assert_allclose(v.numpy(), Z[k + "value"], rtol=0, atol=2e-5)
    # round trip through state_dict
    net2 = PolicyValueNet.from_state_dict(cfg, net.state_dict(), path="full")
    assert torch.equal(net2(x)[0], PolicyValueNet.from_state_dict(cfg, sd, path="full")(x)[0])


def test_full_training_config_seed0():
    """main.py:134 shape at 15x15: same keys/shapes as the reference's state_dict, the same tensors from
    torch.manual_seed(0), and the same outputs on four boards."""
    cfg = NetConfig(15, 15, 2, 225, patch_size=5, embed_dim=512, num_heads=8, depth=1)
    want = json.loads(bytes(Z["full_keys_json"]).decode())
    assert {k: list(v) for k, v in reference_key_shapes(cfg).items()} == want
    w = init_weights(cfg, 0)
    sums = np.array([float(w[k].double().sum()) for k in want], np.float64)
    np.testing.assert_array_equal(sums, Z["full_param_sums"])
    assert sum(v.numel() for v in w.values()) == 3411682
    x = torch.from_numpy(Z["full_x"])
    for path in ("full", "cls"):
        logits, v = PolicyValueNet(cfg, w, path=path)(x)
        np.testing.assert_allclose(logits.numpy(), Z["full_logits"], rtol=0, atol=5e-5)
        np.testing.assert_allclose(v.numpy(), Z["full_value"], rtol=0, atol=2e-5)
    assert abs(cfg.flops_full() / 1e9 - 1.538) < 0.01 and abs(cfg.flops_cls() / 1e9 - 0.254) < 0.01


def test_bad_state_dict_is_rejected():
    cfg = NetConfig(7, 7, 2, 49, 5, 32, 4, 1)
    w = init_weights(cfg, 0)
    w.pop("norm.bias")
    with pytest.raises(KeyError):
        PolicyValueNet(cfg, w)


def test_exact_fold_is_the_same_function():
    """The float64 fold behind the fp32-accurate HIP path (pvnet.exact_fold: cls query through W_k, LayerNorm affines into the
    weights, constant-token softmax terms, Z = ZALL + sum over dirty tokens) evaluated step by step in float64 torch against the
    reference's seed-0 outputs (nn_small.npz full_*) and against the plain forward on boards with 0 .. 112 stones per side."""
    cfg = NetConfig(15, 15, 2, 225, patch_size=5, embed_dim=512, num_heads=8, depth=1)
    net = PolicyValueNet(cfg, seed=0, path="full")
    x = torch.from_numpy(Z["full_x"])
    le, ve, _ = net.forward_exact_emulated(x)
    np.testing.assert_allclose(le.numpy(), Z["full_logits"], rtol=0, atol=5e-6)
    np.testing.assert_allclose(ve.numpy(), Z["full_value"], rtol=0, atol=2e-6)
    rng = np.random.RandomState(1)
    xb = torch.zeros(6, 2, 15, 15)
    for b, n in enumerate([0, 1, 5, 20, 60, 112]):
        cells = rng.choice(225, size=2 * n, replace=False)
        xb[b, 0].view(-1)[cells[:n]] = 1
        xb[b, 1].view(-1)[cells[n:]] = 1
    lf, vf = net(xb)
    le, ve, _ = net.forward_exact_emulated(xb)
    assert (lf - le).abs().max().item() < 5e-6 and (vf - ve).abs().max().item() < 2e-6
    # configurations the fold does not cover are refused, not approximated
    assert PolicyValueNet(NetConfig(15, 15, 2, 225, 5, 512, 8, 2), seed=0).exact_fold() is None
    assert PolicyValueNet(NetConfig(7, 7, 2, 49, 5, 32, 4, 1), seed=0).exact_fold() is None


def test_patch_pooling_fold_is_the_same_function():
    """pvnet.fold_u (the operands of k_embed_fold: LayerNorm1's variance as a quadratic form of the patch bits, scores linear in them,
    the value-projected pooled row as token weights against D_t plus the pooled patch against M_h) evaluated in float64 against the
    value-projected row of the plain pooled tokens (forward_exact_emulated's z through Wv'), boards with 0 .. 112 stones per side;
    and carried through the rest of the tail against the reference's seed-0 logits."""
    import torch.nn.functional as F
    cfg = NetConfig(15, 15, 2, 225, patch_size=5, embed_dim=512, num_heads=8, depth=1)
    net = PolicyValueNet(cfg, seed=0, path="full")
    rng = np.random.RandomState(1)
    xb = torch.zeros(6, 2, 15, 15)
    for b, n in enumerate([0, 1, 5, 20, 60, 112]):
        cells = rng.choice(225, size=2 * n, replace=False)
        xb[b, 0].view(-1)[cells[:n]] = 1
        xb[b, 1].view(-1)[cells[n:]] = 1
    r = net.exact_fold("cpu")
    _, _, z = net.forward_exact_emulated(xb, r)
    u_ref = torch.einsum("nhd,hed->nhe", z.double(), r["Wvn"].double()).reshape(6, 512)
    u, bw, inv_l, pw = net.forward_fold_u_emulated(xb)
    assert (u - u_ref).abs().max().item() < 2e-7 * u_ref.abs().max().item() + 1e-7
    # tokens no stone reaches carry no weight: the kernel leaves them out
    cols = F.unfold(xb.double(), kernel_size=5, padding=2).transpose(1, 2)
    clean = torch.cat([torch.ones(6, 1, dtype=torch.bool), cols.abs().sum(2) == 0], 1)               # [n, T]
    assert bw.transpose(1, 2)[clean].abs().max().item() < 1e-15
    # the rest of the tail on that row: the reference's own outputs for its seed-0 input
    x = torch.from_numpy(Z["full_x"])
    u, _, _, _ = net.forward_fold_u_emulated(x)
    f8 = lambda t: t.double()
    ln = lambda t: (t - t.mean(1, keepdim=True)) / torch.sqrt(t.var(1, unbiased=False, keepdim=True) + 1e-5)
    x1 = u @ f8(r["Wo"]).t() + f8(r["bias1"])
    x2 = x1 + F.gelu(ln(x1) @ f8(r["W0G"]).t() + f8(r["b0G"])) @ f8(r["W3"]).t() + f8(r["b3"])
    out = ln(x2) @ f8(r["WhG"]).t() + f8(r["bhG"])
    np.testing.assert_allclose(out[:, :225].float().numpy(), Z["full_logits"], rtol=0, atol=5e-6)
    assert PolicyValueNet(NetConfig(15, 15, 2, 225, 5, 512, 8, 2), seed=0).fold_u() is None


def test_depth2_network_of_the_compare_mode_against_the_reference():
    """main.py:186-188: Net(rows, patch_size=5, embed_dim=256, num_heads=8, depth=2).  tests/golden/nn_depth2.npz holds the reference's
    seed-0 outputs and parameter sums: the build's initialiser draws the same weights, and the float32 forward (every path) the same
    outputs."""
    import json
    z = load_golden("nn_depth2.npz")
    c = json.loads(bytes(z["cfg_json"]).decode())
    cfg = NetConfig(15, 15, 2, 225, c["patch_size"], c["embed_dim"], c["num_heads"], c["depth"])
    net = PolicyValueNet(cfg, seed=0, device="cpu", dtype=torch.float32, path="full")
    keys = json.loads(bytes(z["keys_json"]).decode())
    assert set(keys) == set(net.state_dict().keys()) and all(list(net.master[k].shape) == v for k, v in keys.items())
    sums = np.array([float(net.master[k].double().sum()) for k in keys])
    assert np.allclose(sums, z["param_sums"], rtol=0, atol=1e-6)
    x = torch.from_numpy(z["x"])
    for path in ("full", "cls"):
        logits, value = net(x, path=path)
        assert (logits - torch.from_numpy(z["logits"])).abs().max().item() < 2e-5
        assert (value - torch.from_numpy(z["value"])).abs().max().item() < 2e-6
