"""The LDS-staged wide links of the cls-row tail (csrc/azk_tail.hip, azk_nn_tail_gemm_lds) against the whole-K-in-registers
form (azk_nn_tail_gemm) and against float64 references.  Reference computation: ai/nn.py:58-60 for the cls row."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def unpack_weight(wp, n_out, k):
    """pack_linear_weight's inverse: the bf16 values as float64 [n_out, k]."""
    npad = (n_out + 63) // 64 * 64
    return wp.view(npad // 64, k // 32, 4, 4, 16, 8).permute(0, 4, 2, 1, 3, 5).reshape(npad, k)[:n_out].double()


@pytest.mark.parametrize("m,live", [(917, None), (2048, 1100), (333, 200), (31, None), (64, 1), (2048, 2048)])
def test_mlp_down_link_is_bit_identical_to_the_register_form(m, live):
    """K = 2048 -> 512 + bias + residual + row statistics: four K-quarter chains added in the order 0..3 in both kernels."""
    import azk
    g = torch.Generator("cuda").manual_seed(m)
    D = 512
    hh = (torch.randn(m, 4 * D, device="cuda", generator=g) * 0.5).to(torch.bfloat16)
    x1 = torch.randn(m, D, device="cuda", generator=g).to(torch.bfloat16)
    w = torch.randn(D, 4 * D, device="cuda", generator=g) * 0.03
    wp = azk.pack_linear_weight(w)
    bias = torch.randn(D, device="cuda", generator=g) * 0.1
    cnt = torch.tensor([live], dtype=torch.int32, device="cuda") if live is not None else None
    outs = []
    for lds in (False, True):
        out = torch.full((m, D), 7.0, device="cuda", dtype=torch.bfloat16)
        st = torch.full((m, D // 64, 2), 7.0, device="cuda")
        azk.nn_tail_gemm(hh, wp, D, 4 * D, azk.TAIL_RESID, bias=bias, resid=x1, out=out, stats_out=st, count=cnt, lds=lds)
        torch.cuda.synchronize()
        outs.append((out, st))
    nl = m if live is None else live
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert bool((outs[1][0][nl:].float() == 7.0).all()) and bool((outs[1][1][nl:] == 7.0).all())      # rows past the live count untouched
    ref = hh[:nl].double() @ unpack_weight(wp, D, 4 * D).t() + bias.double() + x1[:nl].double()
    assert (outs[1][0][:nl].double() - ref).abs().max().item() < 0.05                                 # bf16 output of O(1) values
    # the statistics are those of the rounded rows
    r = outs[1][0][:nl].double().view(nl, D // 64, 64)
    assert (outs[1][1][:nl, :, 0].double() - r.sum(2)).abs().max().item() < 1e-3
    assert (outs[1][1][:nl, :, 1].double() - (r * r).sum(2)).abs().max().item() < 1e-2


@pytest.mark.parametrize("m,live", [(917, None), (2048, 1100), (100, 37), (2048, 2048)])
def test_mlp_up_link_layernorm_in_the_epilogue(m, live):
    """K = 512 -> 2048: GELU(LayerNorm(x1) W'^T + b) with LayerNorm applied in the epilogue, rstd (x1 W'^T - mean colsum(W')),
    against the float64 value from the same bf16 operands: the result is within bf16 output rounding (2^-8 relative + GELU's 1.5e-7
    erf error), and closer to it than the register form, which re-rounds the normalised row to bf16 before the matrix pipe."""
    import azk
    g = torch.Generator("cuda").manual_seed(1000 + m)
    D = 512
    x1 = (torch.randn(m, D, device="cuda", generator=g) * 1.5 + 0.3).to(torch.bfloat16)
    w = torch.randn(4 * D, D, device="cuda", generator=g) * 0.05
    wp = azk.pack_linear_weight(w)
    csum = azk.packed_weight_col_sums(wp, 4 * D, D)
    assert torch.allclose(csum.double(), unpack_weight(wp, 4 * D, D).sum(1), rtol=0, atol=1e-5)
    bias = torch.randn(4 * D, device="cuda", generator=g) * 0.1
    st = torch.stack([x1.float().view(m, 8, 64).sum(2), (x1.float() ** 2).view(m, 8, 64).sum(2)], dim=2).contiguous()
    cnt = torch.tensor([live], dtype=torch.int32, device="cuda") if live is not None else None
    nl = m if live is None else live
    res = {}
    for lds in (False, True):
        out = torch.full((m, 4 * D), 7.0, device="cuda", dtype=torch.bfloat16)
        azk.nn_tail_gemm(x1, wp, 4 * D, D, azk.TAIL_GELU, bias=bias, out=out, a_stats=st, count=cnt, col_sums=csum if lds else None, lds=lds)
        torch.cuda.synchronize()
        assert bool((out[nl:].float() == 7.0).all())
        res[lds] = out[:nl].double()
    xd = x1[:nl].double()
    mean = xd.mean(1, keepdim=True)
    var = (xd * xd).mean(1, keepdim=True) - mean * mean
    xn = (xd - mean) / torch.sqrt(var + 1e-5)
    ref = torch.nn.functional.gelu(xn @ unpack_weight(wp, 4 * D, D).t() + bias.double())
    err_new = (res[True] - ref).abs()
    err_old = (res[False] - ref).abs()
    assert (err_new <= ref.abs() * 2.0 ** -8 + 2e-5).all(), float((err_new - ref.abs() * 2.0 ** -8).max())
    assert err_new.mean().item() <= err_old.mean().item()


def test_whole_chain_with_lds_links_matches_the_register_chain():
    """The five-launch chain with the two wide links LDS-staged against all five in registers: same logits / values to the bf16
    tolerance of the links in between, deterministic, rows independent of the live count."""
    from pvnet import NetConfig, PolicyValueNet
    cfg = NetConfig(15, 15, 2, 225, 5, 512, 8, 1)
    net = PolicyValueNet(cfg, seed=4, device="cuda", dtype=torch.bfloat16, path="clsfold")
    assert net.chain_tail and net.use_lds_tail
    n = 2048
    z = (torch.randn(n, 8, 512, device="cuda", generator=torch.Generator("cuda").manual_seed(5)) * 0.3).to(torch.bfloat16)
    outs = {}
    for live in (917, 1100, 2048):
        lb, vb = torch.full((n, 225), 7.0, device="cuda"), torch.full((n,), 7.0, device="cuda")
        net.out_buffers, net.live_count = (lb, vb), torch.tensor([live], dtype=torch.int32, device="cuda")
        net.tail_fast(z)
        torch.cuda.synchronize()
        assert bool((lb[live:] == 7.0).all()) and bool((vb[live:] == 7.0).all()) and bool(torch.isfinite(lb[:live]).all())
        outs[live] = (lb, vb)
    for live in (917, 1100):
        assert torch.equal(outs[live][0][:live], outs[2048][0][:live]) and torch.equal(outs[live][1][:live], outs[2048][1][:live])
    lb2, vb2 = torch.full((n, 225), 7.0, device="cuda"), torch.full((n,), 7.0, device="cuda")
    net.out_buffers, net.live_count = (lb2, vb2), torch.tensor([2048], dtype=torch.int32, device="cuda")
    net.tail_fast(z)
    assert torch.equal(lb2, outs[2048][0]) and torch.equal(vb2, outs[2048][1])                         # deterministic
    net.use_lds_tail = False
    lr, vr = torch.full((n, 225), 7.0, device="cuda"), torch.full((n,), 7.0, device="cuda")
    net.out_buffers = (lr, vr)
    net.tail_fast(z)
    torch.cuda.synchronize()
    assert (lr - outs[2048][0]).abs().max().item() < 3e-2 and (vr - outs[2048][1]).abs().max().item() < 1e-2


@pytest.mark.parametrize("m,live", [(917, None), (2048, 1100), (100, 37), (2048, 2048)])
def test_fp16_plane_links_are_bit_identical_to_the_register_form(m, live):
    """The fp32-accurate tail's two wide links on (hi, lo) fp16 planes, LDS-staged (azk_nnx_gemm_h_lds) against the register form
    (azk_nnx_gemm_h): the same accumulation chains in the same order and the same epilogue arithmetic - every output bit for bit
    (float32 rows, both planes, row statistics), for both tilings of each link (the in-kernel switch at 1024 live rows)."""
    import azk
    g = torch.Generator("cuda").manual_seed(77 + m)
    D = 512
    cnt = torch.tensor([live], dtype=torch.int32, device="cuda") if live is not None else None
    nl = m if live is None else live

    def planes(x):
        hi, lo = azk.split_fp16(x.double() , azk.GEMM_H_A_SCALE)
        return hi.contiguous(), lo.contiguous()
    # link 3: LayerNorm (epilogue) + 512 -> 2048 + GELU
    x1 = torch.randn(m, D, device="cuda", generator=g) * 1.3 + 0.2
    w0 = torch.randn(4 * D, D, device="cuda", generator=g) * 0.05
    wp, csum = azk.pack_linear_weight_h(w0)
    bias = torch.randn(4 * D, device="cuda", generator=g) * 0.1
    st = torch.stack([x1.view(m, 8, 64).sum(2), (x1 ** 2).view(m, 8, 64).sum(2)], dim=2).contiguous()
    res = []
    for lds in (False, True):
        oh, ol = torch.full((m, 4 * D), 7.0, device="cuda", dtype=torch.float16), torch.full((m, 4 * D), 7.0, device="cuda", dtype=torch.float16)
        azk.nnx_gemm_h(planes(x1), wp, 4 * D, D, azk.TAIL_GELU, bias=bias, col_sums=csum, out=(oh, ol), a_stats=st, count=cnt, lds=lds)
        torch.cuda.synchronize()
        res.append((oh, ol))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert bool((res[1][0][nl:].float() == 7.0).all())
    xd = x1[:nl].double()
    mean = xd.mean(1, keepdim=True)
    xn = (xd - mean) / torch.sqrt((xd * xd).mean(1, keepdim=True) - mean * mean + 1e-5)
    ref = torch.nn.functional.gelu(xn @ w0.double().t() + bias.double())
    got = (res[1][0][:nl].double() + res[1][1][:nl].double()) / azk.GEMM_H_A_SCALE
    assert (got - ref).abs().max().item() < 2e-4
    # link 4: 2048 -> 512 + bias + float32 residual, float32 rows + planes + row statistics
    hh = torch.randn(m, 4 * D, device="cuda", generator=g) * 0.4
    w3 = torch.randn(D, 4 * D, device="cuda", generator=g) * 0.03
    wp3, _ = azk.pack_linear_weight_h(w3)
    b3 = torch.randn(D, device="cuda", generator=g) * 0.1
    xr = torch.randn(m, D, device="cuda", generator=g)
    res = []
    for lds in (False, True):
        oh, ol = torch.full((m, D), 7.0, device="cuda", dtype=torch.float16), torch.full((m, D), 7.0, device="cuda", dtype=torch.float16)
        of, so = torch.full((m, D), 7.0, device="cuda"), torch.full((m, D // 64, 2), 7.0, device="cuda")
        azk.nnx_gemm_h(planes(hh), wp3, D, 4 * D, azk.TAIL_RESID, bias=b3, resid=xr, out=(oh, ol), out_f32=of, stats_out=so, count=cnt, lds=lds)
        torch.cuda.synchronize()
        res.append((oh, ol, of, so))
    for a_, b_ in zip(res[0], res[1]):
        assert torch.equal(a_, b_)
    assert bool((res[1][2][nl:] == 7.0).all()) and bool((res[1][3][nl:] == 7.0).all())
    ref = hh[:nl].double() @ w3.double().t() + b3.double() + xr[:nl].double()
    assert (res[1][2][:nl].double() - ref).abs().max().item() < 2e-4


def test_launch_shape_knobs_do_not_change_results():
    """azk_nn_tail_lds_footprint / azk_nn_embed_fold_grid (launch shapes for game groups stepped on separate streams: two LDS ring buffers
    for the K = 2048 link, a cap on k_embed_fold's grid) change where and when the work runs, never its results."""
    import azk
    from pvnet import NetConfig, PolicyValueNet
    cfg = NetConfig(15, 15, 2, 225, 5, 512, 8, 1)
    net = PolicyValueNet(cfg, seed=8, device="cuda", dtype=torch.bfloat16, path="clsfold")
    x = (torch.rand(700, 2, 15, 15, device="cuda") < 0.08).to(torch.bfloat16)
    x[:, 1] *= 1 - x[:, 0]
    L = azk.lib()
    try:
        ref_l, ref_v = net(x)
        assert L.azk_nn_tail_lds_footprint(1) == 0 and L.azk_nn_embed_fold_grid(96) == 0
        l2, v2 = net(x)
        assert torch.equal(ref_l, l2) and torch.equal(ref_v, v2)
    finally:
        L.azk_nn_tail_lds_footprint(0)
        L.azk_nn_embed_fold_grid(0)
    l3, _ = net(x)
    assert torch.equal(ref_l, l3)
