"""GPU tests of the hand-written embedding kernel (csrc/azk_nn.hip) and the bf16 evaluator paths.
Numerics reference = the same op in plain PyTorch fp32 (tolerances are bf16-level and stated per test)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden
from pvnet import NetConfig, PolicyValueNet

pytestmark = pytest.mark.gpu


def random_boards(n, C, R, Cc, seed):
    rng = np.random.RandomState(seed)
    x = np.zeros((n, C, R, Cc), np.float32)
    for b in range(n):
        dens = rng.uniform(0, 0.9)
        u = rng.rand(R, Cc)
        x[b, 0] = u < dens / 2
        x[b, 1] = (u >= dens / 2) & (u < dens)
        if C == 3:
            x[b, 2] = rng.randint(2)
    return torch.from_numpy(x)


@pytest.mark.parametrize("R,Cc,C,k,D,heads", [(15, 15, 2, 5, 512, 8), (7, 7, 2, 5, 256, 8), (6, 7, 3, 5, 128, 4),
                                              (3, 3, 3, 3, 128, 4), (15, 15, 2, 3, 256, 4)])
def test_patch_embed_kernel_vs_torch_fp32(R, Cc, C, k, D, heads):
    cfg = NetConfig(R, Cc, C, R * Cc if C == 2 else (7 if R == 6 else 9), k, D, heads, 1)
    net32 = PolicyValueNet(cfg, seed=3, device="cuda", dtype=torch.float32, path="full")
    net16 = PolicyValueNet(cfg, weights=net32.state_dict(), device="cuda", dtype=torch.bfloat16, path="full")
    assert net16._hip is not None
    n = 37
    xb = random_boards(n, C, R, Cc, 1).cuda()
    ref = net32.embed(xb)                                        # unfold + fp32 GEMM
    ref_hat = F.layer_norm(ref, (D,), net32.w["blocks.0.norm1.weight"], net32.w["blocks.0.norm1.bias"], 1e-5)
    for inp in (xb, xb.to(torch.bfloat16)):
        x, xhat = net16.embed_hip(inp, want_x=True, want_xhat=True)
        # operands are bf16 (weights rounded to 8 bits), accumulation fp32, output rounded to bf16:
        # |err| <= ~2^-8 * (|x| + sum|w|) ; tokens are O(1..4)
        torch.testing.assert_close(x.float(), ref, rtol=2e-2, atol=3e-2)
        torch.testing.assert_close(xhat.float(), ref_hat, rtol=2e-2, atol=3e-2)
    # cls row is exactly cls + pos[0] rounded once
    want0 = (net32.w["embedding.cls_token"][0, 0] + net32.w["embedding.pos_embedding"][0, 0]).to(torch.bfloat16)
    assert torch.equal(x[:, 0], want0.expand(n, -1))


def test_evaluator_paths_agree_and_match_reference_kat():
    """bf16 evaluator (hand-written embed + torch GEMMs) vs the reference's fp32 outputs for the seed-0 training
    config (tests/golden/nn_small.npz 'full_*'): logits within 2e-2 absolute ("bf16 ~2e-2"), every path incl. the
    benched 'clsfold'.  The error budget and the end-to-end effect on pi: tests/test_gpu_parity_nn.py."""
    z = load_golden("nn_small.npz")
    cfg = NetConfig(15, 15, 2, 225, 5, 512, 8, 1)
    x = torch.from_numpy(z["full_x"]).cuda()
    outs = {}
    for path in ("full", "cls", "clsfold"):                      # clsfold = the path bench.py runs
        net = PolicyValueNet(cfg, seed=0, device="cuda", dtype=torch.bfloat16, path=path)
        logits, v = net(x.to(torch.bfloat16))
        outs[path] = logits
        np.testing.assert_allclose(logits.cpu().numpy(), z["full_logits"], rtol=0, atol=2e-2)      # measured 6.5e-3 .. 8.9e-3
        np.testing.assert_allclose(v.cpu().numpy().reshape(-1), z["full_value"].reshape(-1), rtol=0, atol=2e-3)
    torch.testing.assert_close(outs["full"], outs["cls"], rtol=0, atol=2e-2)
    torch.testing.assert_close(outs["full"], outs["clsfold"], rtol=0, atol=2e-2)
    net32 = PolicyValueNet(cfg, seed=0, device="cuda", dtype=torch.float32, path="cls")
    l32, v32 = net32(x)
    np.testing.assert_allclose(l32.cpu().numpy(), z["full_logits"], rtol=0, atol=1e-4)


@pytest.mark.parametrize("R,D,heads,depth,k", [(15, 512, 8, 1, 5), (7, 256, 8, 2, 5), (7, 128, 4, 1, 3), (15, 256, 4, 2, 5)])
def test_folded_cls_attention_matches_full_forward(R, D, heads, depth, k):
    """path='clsfold' (hand-written embed + folded cls attention, K/V never formed) computes the same function as
    the plain fp32 full forward: logits within 3e-2 absolute (bf16 activations), value within 2e-2."""
    cfg = NetConfig(R, R, 2, R * R, k, D, heads, depth)
    net32 = PolicyValueNet(cfg, seed=5, device="cuda", dtype=torch.float32, path="full")
    net16 = PolicyValueNet(cfg, weights=net32.state_dict(), device="cuda", dtype=torch.bfloat16, path="clsfold")
    assert net16._fold is not None
    x = random_boards(64, 2, R, R, 9).cuda()
    l32, v32 = net32(x)
    l16, v16 = net16(x.to(torch.bfloat16))
    torch.testing.assert_close(l16, l32, rtol=0, atol=4e-2 if depth == 1 else 8e-2)
    torch.testing.assert_close(v16, v32, rtol=0, atol=3e-2)
    # and the policy it induces: softmax rows within 1e-2 in total variation
    tv = 0.5 * (torch.softmax(l16, 1) - torch.softmax(l32, 1)).abs().sum(1).max().item()
    assert tv < 2e-2, tv


def test_cls_attention_kernel_vs_torch():
    """z = softmax(xhat m^T + c) weighted token sums, against the same op in plain PyTorch fp32."""
    import azk
    torch.manual_seed(0)
    for (n, T, D, H, per_board) in [(33, 226, 512, 8, False), (17, 50, 256, 8, True), (9, 10, 128, 4, False), (5, 226, 512, 4, True)]:
        xhat = torch.randn(n, T, D, device="cuda").to(torch.bfloat16)
        m = torch.randn((n, H, D) if per_board else (H, D), device="cuda") * 0.1
        c = torch.randn((n, H) if per_board else (H,), device="cuda")
        z = azk.nn_cls_attention(xhat, m, c, H).float()
        xf = xhat.float()
        mm = m if per_board else m.expand(n, H, D)
        cc = c if per_board else c.expand(n, H)
        s = torch.einsum("ntd,nhd->nht", xf, mm) + cc[:, :, None]
        ref = torch.einsum("nht,ntd->nhd", torch.softmax(s, dim=2), xf)
        torch.testing.assert_close(z, ref, rtol=1e-2, atol=1e-2)     # output rounded to bf16; fp32 accumulation inside


@pytest.mark.parametrize("D", [128, 256, 512])
def test_layernorm_rows_and_heads_finalize_vs_torch_fp32(D):
    """azk_nn_layernorm_rows / azk_nn_heads_finalize against the plain fp32 ops on the same bf16 inputs.
    Tolerance: outputs are rounded once to bf16 (2^-8 relative) -> 1e-2 absolute on O(1) values."""
    import azk
    torch.manual_seed(D)
    n = 301
    x = (torch.randn(n, D, device="cuda") * 1.7 + 0.3).to(torch.bfloat16)
    w = torch.randn(D, device="cuda") * 0.5 + 1.0
    b = torch.randn(D, device="cuda") * 0.2
    add = torch.randn(D, device="cuda") * 0.3
    ref = F.layer_norm(x.float(), (D,), w, b, 1e-5)
    x_in = x.clone()
    y = azk.nn_layernorm_rows(x_in, w, b, 1e-5)
    assert torch.equal(x_in, x)                                   # no add_bias: input untouched
    assert (y.float() - ref).abs().max().item() < 3e-2
    assert (y.float() - ref).abs().mean().item() < 3e-3
    y2 = azk.nn_layernorm_rows(x_in, w, b, 1e-5, add_bias=add)
    assert torch.equal(y2, y)
    assert torch.equal(x_in, (x.float() + add).to(torch.bfloat16))
    # device-side row count: rows beyond it are left alone
    cnt = torch.tensor([100], dtype=torch.int32, device="cuda")
    x3 = x.clone()
    y3 = torch.full_like(x, 7.0)
    rc = azk.lib().azk_nn_layernorm_rows(x3.data_ptr(), w.data_ptr(), b.data_ptr(), 1e-5, y3.data_ptr(), add.data_ptr(), n, D,
                                         cnt.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    assert torch.equal(y3[:100], y[:100]) and bool((y3[100:] == 7.0).all()) and torch.equal(x3[100:], x[100:])

    A = 225
    heads = torch.randn(n, 232, device="cuda").to(torch.bfloat16)
    lo = torch.zeros(n, A, device="cuda")
    vo = torch.zeros(n, device="cuda")
    azk.nn_heads_finalize(heads, A, lo, vo)
    assert torch.equal(lo, heads[:, :A].float())
    assert (vo - torch.tanh(heads[:, A].float())).abs().max().item() < 1e-6


@pytest.mark.parametrize("static_ref", [True, False])
def test_fused_embed_pool_vs_torch_fp32(static_ref):
    """azk_nn_embed_pool (embedding + LayerNorm1 + folded cls attention in one launch) against the same computation in plain
    fp32 PyTorch on the operands the kernel sees (bf16-rounded weights): z = softmax_t(xn . m') @ xn.
    Tolerance: xn and the softmax weights are rounded to bf16 for the second MFMA, z to bf16 on output: 1e-2 absolute
    (values are O(0.1)); the two-launch path is checked against the same reference at 6e-2 (its bf16 token round trip)."""
    import azk
    cfg = NetConfig(15, 15, 2, 225, 5, 512, 8, 1)
    net = PolicyValueNet(cfg, seed=5, device="cuda", dtype=torch.bfloat16, path="clsfold")
    assert net.fused_embed_pool
    f, hp = net._fold, net._hip
    n = 300
    x = random_boards(n, 2, 15, 15, 9).cuda().to(torch.bfloat16).contiguous()
    x[0] = 0                                                        # empty board: every token but cls sees a zero patch
    cols = F.unfold(x.float(), kernel_size=5, padding=2).transpose(1, 2)
    W = f["wt_ext"][:512, :50].float()
    tok = torch.cat([torch.zeros(n, 1, 512, device="cuda"), cols @ W.t()], 1) + hp["cpos"]
    xn = F.layer_norm(tok, (512,))
    ref = torch.einsum("bth,btd->bhd", torch.softmax(xn @ f["m_n"].t(), 1), xn)
    assert f["score_ref"] is not None
    cnt = torch.tensor([257], dtype=torch.int32, device="cuda")
    z = azk.nn_embed_pool(x, f["wt_ext"], f["cpos_frag"], f["score_frag"], f["score_msum"], f["score_ref"] if static_ref else None,
                          15, 15, 5, 512, 8)
    err = (z.float() - ref).abs()
    assert err.max().item() < 1e-2 and err.mean().item() < 1e-3, (err.max().item(), err.mean().item())
    z2 = torch.full_like(z, 3.0)
    rc = azk.lib().azk_nn_embed_pool(x.data_ptr(), 0, f["wt_ext"].data_ptr(), f["cpos_frag"].data_ptr(), f["score_frag"].data_ptr(),
                                     f["score_msum"].data_ptr(), f["score_ref"].data_ptr() if static_ref else None, z2.data_ptr(), 8, n, 2, 15, 15, 5,
                                     64, 512, 1e-5, cnt.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    assert torch.equal(z2[:257], z[:257]) and bool((z2[257:] == 3.0).all())       # device-side count respected
    two = azk.nn_embed_scores_pool(x, f["wt_ext"], hp["cpos"], f["score_cpos"], f["score_msum"], f["c_n"], 15, 15, 5, 512, 8)
    assert (two.float() - ref).abs().max().item() < 6e-2
    # float32 boards give the same bits as bf16 boards
    zf = azk.nn_embed_pool(x.float().contiguous(), f["wt_ext"], f["cpos_frag"], f["score_frag"], f["score_msum"],
                           f["score_ref"] if static_ref else None, 15, 15, 5, 512, 8)
    assert torch.equal(zf, z)


def test_hand_written_tail_matches_library_tail():
    """azk_nn_gemm_rows / azk_nn_layernorm_sum / azk_nn_heads_finalize_sum (cls-row tail with a device-side row count)
    against the same tail on library GEMMs, and the GEMM alone against fp32 matmul.  bf16 operands, fp32 accumulation:
    logits agree to 3e-2 absolute (scale ~0.5), values to 5e-3."""
    import azk
    cfg = NetConfig(15, 15, 2, 225, 5, 512, 8, 1)
    net = PolicyValueNet(cfg, seed=2, device="cuda", dtype=torch.bfloat16, path="clsfold")
    assert net.hip_tail
    n, live = 333, 200
    torch.manual_seed(0)
    z = (torch.randn(n, 8, 512, device="cuda") * 0.1).to(torch.bfloat16)
    net.use_hip_tail = False
    l_ref, v_ref = net.tail_fast(z)
    net.use_hip_tail = True
    l_hip, v_hip = net.tail_fast(z)
    assert (l_hip - l_ref).abs().max().item() < 3e-2 and (v_hip.reshape(-1) - v_ref.reshape(-1)).abs().max().item() < 5e-3
    net.live_count = torch.tensor([live], dtype=torch.int32, device="cuda")
    lb = torch.full((n, 225), 9.0, device="cuda")
    vb = torch.full((n,), 9.0, device="cuda")
    net.out_buffers = (lb, vb)
    net.tail_fast(z)
    assert torch.equal(lb[:live], l_hip[:live]) and bool((lb[live:] == 9.0).all()) and bool((vb[live:] == 9.0).all())
    # the GEMM alone: partial planes sum to A W^T; GELU epilogue
    a = (torch.randn(150, 1024, device="cuda") * 0.3).to(torch.bfloat16)
    w = (torch.randn(200, 1024, device="cuda") * 0.05)
    wp = azk.pack_linear_weight(w)
    P = torch.zeros(2, 150, 256, device="cuda")
    azk.nn_gemm_rows(a, wp, 256, ksplit=2, partials=P)
    ref = a.float() @ w.to(torch.bfloat16).float().t()
    got = P.sum(0)[:, :200]
    assert (got - ref).abs().max().item() < 2e-3 * ref.abs().max().item() + 1e-3
    assert bool((P.sum(0)[:, 200:] == 0).all())
    bias = torch.randn(256, device="cuda") * 0.1
    g = torch.empty(150, 256, device="cuda", dtype=torch.bfloat16)
    azk.nn_gemm_rows(a, wp, 256, bias=bias, gelu_out=g)
    refg = F.gelu(torch.cat([ref, torch.zeros(150, 56, device="cuda")], 1) + bias)
    assert (g.float() - refg).abs().max().item() < 2e-2


def test_fused_final_norm_and_heads_vs_torch_fp32():
    """azk_nn_ln_heads (final LayerNorm + merged policy/value head + tanh in one launch, the default cls tail) against
    fp32 PyTorch on the same bf16 input: logits to 2e-2 (bf16 operands, scale ~0.5), value to 5e-3; honours the row count."""
    import azk
    cfg = NetConfig(15, 15, 2, 225, 5, 512, 8, 1)
    net = PolicyValueNet(cfg, seed=4, device="cuda", dtype=torch.bfloat16, path="clsfold")
    assert net.hip_tail and net.fuse_ln_heads
    f, m = net._fold, net.master
    n, live = 210, 77
    torch.manual_seed(1)
    x = (torch.randn(n, 512, device="cuda") * 1.3 + 0.2).to(torch.bfloat16)
    y = F.layer_norm(x.float(), (512,), m["norm.weight"].cuda(), m["norm.bias"].cuda(), 1e-5)
    ref_l = y @ m["policy_head.weight"].cuda().t() + m["policy_head.bias"].cuda()
    ref_v = torch.tanh(y @ m["value_head.weight"].cuda().t() + m["value_head.bias"].cuda()).reshape(-1)
    lb = torch.full((n, 225), 5.0, device="cuda")
    vb = torch.full((n,), 5.0, device="cuda")
    azk.nn_ln_heads(x, f["lnf_w"], f["lnf_b"], f["WhP"], f["bh_f"], 225, lb, vb)
    assert (lb - ref_l).abs().max().item() < 2e-2 and (vb - ref_v).abs().max().item() < 5e-3
    # the default path: LayerNorm's affine folded into the head weight / bias, slab fetched once
    lbf = torch.full((n, 225), 5.0, device="cuda")
    vbf = torch.full((n,), 5.0, device="cuda")
    azk.nn_ln_heads(x, None, None, f["WhGP"], f["bhG_f"], 225, lbf, vbf)
    assert (lbf - ref_l).abs().max().item() < 2e-2 and (vbf - ref_v).abs().max().item() < 5e-3
    lbf2 = torch.full((n, 225), 5.0, device="cuda")
    vbf2 = torch.full((n,), 5.0, device="cuda")
    azk.nn_ln_heads(x, None, None, f["WhGP"], f["bhG_f"], 225, lbf2, vbf2, count=torch.tensor([live], dtype=torch.int32, device="cuda"))
    assert torch.equal(lbf2[:live], lbf[:live]) and bool((lbf2[live:] == 5.0).all()) and bool((vbf2[live:] == 5.0).all())
    lb2 = torch.full((n, 225), 5.0, device="cuda")
    vb2 = torch.full((n,), 5.0, device="cuda")
    azk.nn_ln_heads(x, f["lnf_w"], f["lnf_b"], f["WhP"], f["bh_f"], 225, lb2, vb2, count=torch.tensor([live], dtype=torch.int32, device="cuda"))
    assert torch.equal(lb2[:live], lb[:live]) and bool((lb2[live:] == 5.0).all()) and bool((vb2[live:] == 5.0).all())


def test_live_count_leaves_valid_rows_unchanged():
    """The step graph runs the network over the fixed-size leaf buffer with a device-side live count: the rows below the
    count must come out bit-identical to a run without the count (same launch shapes), rows above are simply not produced."""
    cfg = NetConfig(15, 15, 2, 225, 5, 512, 8, 1)
    net = PolicyValueNet(cfg, seed=1, device="cuda", dtype=torch.bfloat16, path="clsfold")
    n, live = 256, 150
    x = random_boards(n, 2, 15, 15, 3).cuda().to(torch.bfloat16).contiguous()
    lb0, vb0 = torch.zeros(n, 225, device="cuda"), torch.zeros(n, device="cuda")
    net.out_buffers = (lb0, vb0)
    net(x)
    lb1, vb1 = torch.full((n, 225), 7.0, device="cuda"), torch.full((n,), 7.0, device="cuda")
    net.out_buffers = (lb1, vb1)
    net.live_count = torch.tensor([live], dtype=torch.int32, device="cuda")
    net(x)
    torch.cuda.synchronize()
    assert torch.equal(lb1[:live], lb0[:live]) and torch.equal(vb1[:live], vb0[:live])
    assert bool((lb1[live:] == 7.0).all()) and bool((vb1[live:] == 7.0).all())


def test_embed_pool_from_engine_leaves_equals_gathered_batch():
    """azk_nn_embed_pool_leaves (boards straight from the engine's pending leaves: own flag prefix, cell codes, slots, count)
    against azk_step_gather + azk_nn_embed_pool on the same engine state: same leaf order, same count, bit-identical pooled
    tokens; and whole continuous-self-play moves through the real network are identical with and without the compaction
    launch (which also proves the slots the kernel hands to the next expansion)."""
    import azk
    from selfplay import SelfPlayRunner
    cfg = NetConfig(15, 15, 2, 225, 5, 512, 8, 1)
    net = PolicyValueNet(cfg, seed=6, device="cuda", dtype=torch.bfloat16, path="clsfold")
    f = net._fold
    G, A = 300, 225
    eng = azk.Engine("gomoku", G, 64, size=15, leaf_dtype="bfloat16", cache_entries=64)
    eng.reset_games()
    noise, uni = eng.gen_noise(3, 0, 0)
    eng.begin_search(noise)
    logits = values = None
    for s in range(24):                                           # a few simulations so leaves, cache hits and terminals mix
        eng.step_tree(logits, values)
        eng.step_gather()
        logits, values = torch.randn(G, A, device="cuda") * 0.3, torch.tanh(torch.randn(G, device="cuda"))
    n = int(eng.n_leaf.item())
    assert 0 < n <= G
    z_ref = azk.nn_embed_pool(eng.leaf_boards[:n].contiguous(), f["wt_ext"], f["cpos_frag"], f["score_frag"], f["score_msum"], f["score_ref"],
                              15, 15, 5, 512, 8)
    eng.n_leaf.zero_()
    z_new = azk.nn_embed_pool_leaves(eng.leaf_source(), f["wt_ext"], f["cpos_frag"], f["score_frag"], f["score_msum"], f["score_ref"], 5, 512, 8)
    torch.cuda.synchronize()
    assert int(eng.n_leaf.item()) == n
    assert torch.equal(z_new[:n], z_ref)
    eng.close()

    def play(leaves):
        rec = []
        r = SelfPlayRunner("gomoku", net, 128, 48, size=15, seed=9, leaf_dtype="bfloat16", recycle=True, use_graph=True, cache_entries=128,
                           on_records=lambda mv, base, pi, q, ch, w, d: rec.append((pi.numpy().copy(), ch.numpy().copy())))
        r.leaf_source_ok = leaves
        for _ in range(3):
            r.play_move()
        r.check_error()
        return rec
    a, b = play(True), play(False)
    for (pa, ca), (pb, cb) in zip(a, b):
        assert np.array_equal(ca, cb) and pa.tobytes() == pb.tobytes()


def _stone_boards(n, seed):
    """Boards shaped like the benchmark's leaves (a cluster of ~5-40 stones) plus the edge cases: empty board, one stone in a
    corner, a nearly full board (every token dirty)."""
    rng = np.random.RandomState(seed)
    x = np.zeros((n, 2, 15, 15), np.float32)
    for b in range(3, n):
        k = rng.randint(1, 45)
        cells = [(7, 7)]
        for _ in range(k):
            r, c = cells[rng.randint(len(cells))]
            r2, c2 = np.clip(r + rng.randint(-1, 2), 0, 14), np.clip(c + rng.randint(-1, 2), 0, 14)
            cells.append((int(r2), int(c2)))
        for i, (r, c) in enumerate(dict.fromkeys(cells)):
            x[b, i & 1, r, c] = 1
    x[1, 0, 0, 0] = 1
    u = rng.rand(15, 15)
    x[2, 0] = u < 0.45
    x[2, 1] = (u >= 0.45) & (u < 0.9)
    return torch.from_numpy(x)


def test_compact_embed_pool_vs_torch_fp32_and_full_kernel():
    """azk_nn_embed_pool_compact (only the tokens a stone can reach are evaluated; the empty-patch tokens enter as
    precomputed constants) against the plain fp32 computation and against azk_nn_embed_pool, which evaluates every token.
    Same tolerance as the full kernel (bf16 operands of the second MFMA, bf16 output): 1e-2 max / 1e-3 mean absolute on
    O(0.1) values; the two kernels differ by fp32 reassociation and the constant tokens' host-side fp32 statistics only."""
    import azk
    cfg = NetConfig(15, 15, 2, 225, 5, 512, 8, 1)
    net = PolicyValueNet(cfg, seed=5, device="cuda", dtype=torch.bfloat16, path="clsfold")
    assert net.fused_embed_pool and net._compact is not None
    f, hp = net._fold, net._hip
    n = 700                                                         # more boards than resident workgroups: the queue is exercised
    x = _stone_boards(n, 4).cuda().to(torch.bfloat16).contiguous()
    cols = F.unfold(x.float(), kernel_size=5, padding=2).transpose(1, 2)
    W = f["wt_ext"][:512, :50].float()
    tok = torch.cat([torch.zeros(n, 1, 512, device="cuda"), cols @ W.t()], 1) + hp["cpos"]
    xn = F.layer_norm(tok, (512,))
    ref = torch.einsum("bth,btd->bhd", torch.softmax(xn @ f["m_n"].t(), 1), xn)
    sched = azk.new_sched("cuda")
    z = azk.nn_embed_pool_compact(x, net._compact, 15, 15, sched)
    err = (z.float() - ref).abs()
    assert err.max().item() < 1e-2 and err.mean().item() < 1e-3, (err.max().item(), err.mean().item())
    zfull = azk.nn_embed_pool(x, f["wt_ext"], f["cpos_frag"], f["score_frag"], f["score_msum"], f["score_ref"], 15, 15, 5, 512, 8)
    d = (z.float() - zfull.float()).abs()
    assert d.max().item() < 6e-3 and d.mean().item() < 3e-4, (d.max().item(), d.mean().item())
    torch.cuda.synchronize()
    assert sched.tolist() == [0, 0]                                 # the queue words are left zero for the next launch
    # scheduling does not change a bit: boards are independent work items
    for _ in range(3):
        assert torch.equal(azk.nn_embed_pool_compact(x, net._compact, 15, 15, sched), z)
    assert torch.equal(azk.nn_embed_pool_compact(x.float().contiguous(), net._compact, 15, 15, sched), z)
    # device-side count: rows past it are not produced
    cnt = torch.tensor([301], dtype=torch.int32, device="cuda")
    z2 = torch.full_like(z, 3.0)
    rc = azk.lib().azk_nn_embed_pool_compact(x.data_ptr(), 0, azk.C.byref(net._compact.c), z2.data_ptr(), n, 2, 15, 15, cnt.data_ptr(),
                                             sched.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    assert torch.equal(z2[:301], z[:301]) and bool((z2[301:] == 3.0).all())
    torch.cuda.synchronize()
    assert sched.tolist() == [0, 0]
    # a permutation of the batch permutes the rows (no cross-board state)
    perm = torch.randperm(n, device="cuda")
    assert torch.equal(azk.nn_embed_pool_compact(x[perm].contiguous(), net._compact, 15, 15, sched), z[perm])


def test_compact_embed_pool_from_engine_leaves_equals_gathered_batch():
    """azk_nn_embed_pool_compact_leaves against azk_step_gather + azk_nn_embed_pool_compact on the same engine state."""
    import azk
    cfg = NetConfig(15, 15, 2, 225, 5, 512, 8, 1)
    net = PolicyValueNet(cfg, seed=6, device="cuda", dtype=torch.bfloat16, path="clsfold")
    G, A = 700, 225
    eng = azk.Engine("gomoku", G, 64, size=15, leaf_dtype="bfloat16", cache_entries=64)
    eng.reset_games()
    noise, uni = eng.gen_noise(3, 0, 0)
    eng.begin_search(noise)
    logits = values = None
    for s in range(24):
        eng.step_tree(logits, values)
        eng.step_gather()
        logits, values = torch.randn(G, A, device="cuda") * 0.3, torch.tanh(torch.randn(G, device="cuda"))
    n = int(eng.n_leaf.item())
    assert 0 < n <= G
    sched = azk.new_sched("cuda")
    z_ref = azk.nn_embed_pool_compact(eng.leaf_boards[:n].contiguous(), net._compact, 15, 15, sched)      # rows in game order (azk_step_gather)
    import ctypes as C
    src = eng.leaf_source()
    hip = C.CDLL("libamdhip64.so")

    def peek(ptr, count, dtype):                                    # engine-owned device arrays of the leaf source
        buf = np.empty(count, dtype)
        assert hip.hipMemcpy(buf.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), C.c_size_t(buf.nbytes), 2) == 0
        return buf
    torch.cuda.synchronize()
    gather_slot = peek(src.leaf_slot, G, np.int32)                  # slot of each flagged game in the gathered batch
    fl = peek(src.leaf_flag, G, np.uint8)
    eng.n_leaf.zero_()
    z_new = azk.nn_embed_pool_compact_leaves(src, net._compact, sched)
    torch.cuda.synchronize()
    assert int(eng.n_leaf.item()) == n and sched.tolist() == [0, 0]
    new_slot = peek(src.leaf_slot, G, np.int32)
    games = np.nonzero(fl)[0]
    assert len(games) == n and sorted(new_slot[games].tolist()) == list(range(n))                  # a permutation of the rows
    # the kernel hands the boards out from the stone-heavy cost classes down (flag = 1 + class), game order inside a class
    order = sorted(games.tolist(), key=lambda g: (-int(fl[g]), g))
    assert [int(new_slot[g]) for g in order] == list(range(n))
    for g in games[:: max(1, n // 64)]:
        assert torch.equal(z_new[int(new_slot[g])], z_ref[int(gather_slot[g])])                       # same board, same bits, other row
    eng.close()


def test_tail_chain_vs_torch_fp32_and_library_tail():
    """The five-launch tail chain (azk_nn_tail_gemm: per-head value projection, output projection + LN statistics, LN2 + MLP up +
    GELU, MLP down + residual + LN statistics, final LN + merged heads) against the same cls-row computation in plain fp32 PyTorch
    from the same pooled tokens z, and against the library-GEMM tail.  bf16 activations between the links (as in the library
    tail): logits within 3e-2 absolute, value within 1e-2; a device-side live count leaves the rows below it bit-identical."""
    import azk
    cfg = NetConfig(15, 15, 2, 225, 5, 512, 8, 1)
    net = PolicyValueNet(cfg, seed=2, device="cuda", dtype=torch.bfloat16, path="clsfold")
    assert net.chain_tail
    f, m = net._fold, {k: v.cuda() for k, v in net.master.items()}
    n, live, D, H = 333, 200, 512, 8
    x = random_boards(n, 2, 15, 15, 5).cuda().to(torch.bfloat16).contiguous()
    z = azk.nn_embed_pool(x, f["wt_ext"], f["cpos_frag"], f["score_frag"], f["score_msum"], f["score_ref"], 15, 15, 5, 512, 8)     # [n, H, D]
    # fp32 reference of the tail from the same z (z = softmax-weighted sum of the normalised tokens, before LN1's affine)
    g1, b1 = m["blocks.0.norm1.weight"], m["blocks.0.norm1.bias"]
    Wi, bi = m["blocks.0.attn.in_proj_weight"], m["blocks.0.attn.in_proj_bias"]
    zz = z.float() * g1 + b1                                                                          # sum_t a_t xhat_t (the weights sum to 1)
    Wv, bv = Wi[2 * D:].view(H, D // H, D), bi[2 * D:].view(H, D // H)
    a = (torch.einsum("nhd,hed->nhe", zz, Wv) + bv).reshape(n, D)
    x0 = m["embedding.cls_token"][0, 0] + m["embedding.pos_embedding"][0, 0]
    x1 = x0 + a @ m["blocks.0.attn.out_proj.weight"].t() + m["blocks.0.attn.out_proj.bias"]
    h = F.layer_norm(x1, (D,), m["blocks.0.norm2.weight"], m["blocks.0.norm2.bias"], 1e-5)
    x2 = x1 + F.gelu(h @ m["blocks.0.mlp.0.weight"].t() + m["blocks.0.mlp.0.bias"]) @ m["blocks.0.mlp.3.weight"].t() + m["blocks.0.mlp.3.bias"]
    y = F.layer_norm(x2, (D,), m["norm.weight"], m["norm.bias"], 1e-5)
    ref_l = y @ m["policy_head.weight"].t() + m["policy_head.bias"]
    ref_v = torch.tanh(y @ m["value_head.weight"].t() + m["value_head.bias"])
    net.use_chain_tail = True
    lc, vc = net.tail_fast(z)
    net.use_chain_tail = False
    ll, vl = net.tail_fast(z)
    assert (lc - ref_l).abs().max().item() < 3e-2 and (vc - ref_v).abs().max().item() < 1e-2, ((lc - ref_l).abs().max().item(), (vc - ref_v).abs().max().item())
    assert (ll - ref_l).abs().max().item() < 3e-2                                                    # the library tail meets the same bound
    assert (lc - ll).abs().max().item() < 3e-2
    # live count: rows below it identical, rows above untouched
    net.use_chain_tail = True
    lb, vb = torch.full((n, 225), 7.0, device="cuda"), torch.full((n,), 7.0, device="cuda")
    net.out_buffers, net.live_count = (lb, vb), torch.tensor([live], dtype=torch.int32, device="cuda")
    net.tail_fast(z)
    torch.cuda.synchronize()
    assert torch.equal(lb[:live], lc[:live]) and torch.equal(vb[:live], vc[:live, 0])
    assert bool((lb[live:] == 7.0).all()) and bool((vb[live:] == 7.0).all())
    # deterministic: the LayerNorm statistics are summed in a fixed order (no atomics)
    net.out_buffers, net.live_count = None, None
    l2, v2 = net.tail_fast(z)
    assert torch.equal(l2, lc) and torch.equal(v2, vc)


def test_tail_chain_rows_do_not_depend_on_the_live_count():
    """azk_nn_tail_gemm picks its wave tile height (32 / 48 / 64 rows) from the live row count so that one launch is one round
    of waves; a row's result must not depend on that choice: the first 900 rows are bit-identical whether 900, 1100 (48-row
    tiles), 1700 (64-row tiles) rows or the whole buffer are live, and every live count matches the fp32-checked chain."""
    import azk
    cfg = NetConfig(15, 15, 2, 225, 5, 512, 8, 1)
    net = PolicyValueNet(cfg, seed=3, device="cuda", dtype=torch.bfloat16, path="clsfold")
    assert net.chain_tail
    n = 2048
    z = (torch.randn(n, 8, 512, device="cuda", generator=torch.Generator("cuda").manual_seed(5)) * 0.3).to(torch.bfloat16)
    net.use_chain_tail = True
    outs = {}
    for live in (900, 1024, 1100, 1537, 1700, 2048):
        lb, vb = torch.full((n, 225), 7.0, device="cuda"), torch.full((n,), 7.0, device="cuda")
        net.out_buffers, net.live_count = (lb, vb), torch.tensor([live], dtype=torch.int32, device="cuda")
        net.tail_fast(z)
        torch.cuda.synchronize()
        assert bool((lb[live:] == 7.0).all()) and bool((vb[live:] == 7.0).all())
        assert bool(torch.isfinite(lb[:live]).all())
        outs[live] = (lb, vb)
    for live in (900, 1024, 1100, 1537, 1700):
        assert torch.equal(outs[live][0][:live], outs[2048][0][:live]) and torch.equal(outs[live][1][:live], outs[2048][1][:live]), live
    net.out_buffers, net.live_count = None, None
    net.use_chain_tail = False
    ll, vl = net.tail_fast(z)
    assert (outs[2048][0] - ll).abs().max().item() < 5e-2 and (outs[2048][1] - vl[:, 0]).abs().max().item() < 2e-2


def test_runner_budget_stepping_plays_the_same_moves():
    """SelfPlayRunner with budget stepping (graph replays until no game owes simulations) against one-simulation-per-replay
    stepping, real network, continuous self-play with recycling: the same pi and the same moves, in fewer launches."""
    from selfplay import SelfPlayRunner
    cfg = NetConfig(15, 15, 2, 225, 5, 512, 8, 1)
    net = PolicyValueNet(cfg, seed=6, device="cuda", dtype=torch.bfloat16, path="clsfold")

    def play(budget):
        rec = []
        r = SelfPlayRunner("gomoku", net, 128, 96, size=15, seed=9, leaf_dtype="bfloat16", recycle=True, use_graph=True, cache_entries=256,
                           cache_shared=True, budget_stepping=budget,
                           on_records=lambda mv, base, pi, q, ch, w, d: rec.append((pi.numpy().copy(), ch.numpy().copy())))
        for _ in range(4):
            r.play_move()
        r.check_error()
        return rec, r.launches, r.counters()
    (a, la, ca), (b, lb, cb) = play(False), play(True)
    for (pa, cha), (pb, chb) in zip(a, b):
        assert np.array_equal(cha, chb) and pa.tobytes() == pb.tobytes()
    assert ca["sims"] == cb["sims"] == 4 * 128 * 96
    assert lb <= la + 4, (la, lb)          # early-game searches hit few terminals / cache entries: about as many launches, never more


def test_runner_steps_per_graph_plays_the_same_moves():
    """Several simulation steps captured in one hipGraph (the bench default: 32) against one step per graph and against eager
    stepping: the same pi and the same moves, whatever the chunking (searches of 96 and of 50 simulations: whole chunks, a
    remainder of single steps, a kernel-timer sample in between)."""
    from selfplay import KernelTimer, SelfPlayRunner
    cfg = NetConfig(15, 15, 2, 225, 5, 512, 8, 1)
    net = PolicyValueNet(cfg, seed=6, device="cuda", dtype=torch.bfloat16, path="clsfold")

    def play(per_graph, sims, use_graph=True, timer=None):
        rec = []
        r = SelfPlayRunner("gomoku", net, 128, sims, size=15, seed=9, leaf_dtype="bfloat16", recycle=True, use_graph=use_graph, cache_entries=256,
                           cache_shared=True, steps_per_graph=per_graph, kernel_timer=timer,
                           on_records=lambda mv, base, pi, q, ch, w, d: rec.append((pi.numpy().copy(), ch.numpy().copy())))
        if timer is not None:
            timer.enabled = True
        for _ in range(3):
            r.play_move()
        r.check_error()
        return rec, r.counters()["sims"]
    for sims in (96, 50):
        ref, n_ref = play(1, sims)
        for per_graph, graph, timer in ((32, True, None), (8, True, KernelTimer(stride=20)), (1, False, None)):
            got, n = play(per_graph, sims, graph, timer)
            assert n == n_ref == 3 * 128 * sims
            for (pa, cha), (pb, chb) in zip(ref, got):
                assert np.array_equal(cha, chb) and pa.tobytes() == pb.tobytes(), (sims, per_graph, graph)


def test_runner_game_groups_on_concurrent_streams_play_the_same_moves():
    """`n_split` game groups, each with its own captured graphs on its own stream, chosen so that the streams really run side by side
    (selfplay.concurrent_streams: distinct hardware queues).  The groups share ONE evaluator; its per-engine workspaces, schedules
    and graph-private intermediates must keep the concurrent forwards apart: the same pi bytes and moves as one group."""
    from selfplay import SelfPlayRunner, concurrent_streams
    sts = concurrent_streams(torch, torch.device("cuda:0"), 2)
    assert len(sts) == 2 and sts[0].cuda_stream != sts[1].cuda_stream
    cfg = NetConfig(15, 15, 2, 225, 5, 512, 8, 1)
    net = PolicyValueNet(cfg, seed=6, device="cuda", dtype=torch.bfloat16, path="clsfold")

    def play(n_split):
        rec = {}
        def on(mv, base, pi, q, ch, w, d):
            rec[(mv, base)] = (pi.numpy().copy(), ch.numpy().copy())
        r = SelfPlayRunner("gomoku", net, 256, 64, size=15, seed=13, leaf_dtype="bfloat16", recycle=True, use_graph=True, cache_entries=256,
                           cache_shared=True, steps_per_graph=8, n_split=n_split, on_records=on)
        for _ in range(4):
            r.play_move()
        r.check_error()
        per = 256 // n_split
        return [(np.concatenate([rec[(mv, g * per)][0] for g in range(n_split)]), np.concatenate([rec[(mv, g * per)][1] for g in range(n_split)]))
                for mv in range(4)]
    one, two = play(1), play(2)
    for (pa, ca), (pb, cb) in zip(one, two):
        assert np.array_equal(ca, cb) and pa.tobytes() == pb.tobytes()


def test_runner_real_network_eval_cache_is_transparent():
    """The eval cache under the REAL bf16 network: 256 games, continuous self-play on the graph runner, the
    cache off / one table per game / ONE table shared by every game (multi-writer: claim word, copy, re-read) must give the same
    pi bytes and the same moves - a torn or stale row of the shared table would change a prior and show up here.  The shared
    table is made small (64 entries per game) so that entries are overwritten while other games read them."""
    from selfplay import SelfPlayRunner
    cfg = NetConfig(15, 15, 2, 225, 5, 512, 8, 1)
    net = PolicyValueNet(cfg, seed=6, device="cuda", dtype=torch.bfloat16, path="clsfold")

    def play(entries, shared):
        rec = []
        r = SelfPlayRunner("gomoku", net, 256, 64, size=15, seed=11, leaf_dtype="bfloat16", recycle=True, use_graph=True, cache_entries=entries,
                           cache_shared=shared, steps_per_graph=8,
                           on_records=lambda mv, base, pi, q, ch, w, d: rec.append((pi.numpy().copy(), q.numpy().copy(), ch.numpy().copy())))
        for _ in range(6):
            r.play_move()
        r.check_error()
        return rec, r.counters()
    off, c_off = play(0, False)
    per, c_per = play(64, False)
    sh, c_sh = play(64, True)
    assert c_off["cache_hits"] == 0 and c_per["cache_hits"] > 0 and c_sh["cache_hits"] > 0
    assert c_off["sims"] == c_per["sims"] == c_sh["sims"] == 6 * 256 * 64
    for (pa, qa, ca), (pb, qb, cb), (pc, qc, cc) in zip(off, per, sh):
        assert np.array_equal(ca, cb) and np.array_equal(ca, cc)
        assert pa.tobytes() == pb.tobytes() == pc.tobytes() and qa.tobytes() == qb.tobytes() == qc.tobytes()


def test_in_place_promotion_keeps_the_captured_graphs():
    """main.py:55-59 on the graph runner: PolicyValueNet.load_state_dict refreshes every device buffer the
    kernels read in place (same addresses), so the captured step graphs replay the NEW weights without re-capture: the moves
    after an in-place promotion equal those of a runner whose graphs were re-captured, and those of a net built from the new
    weights; both evaluator families (bf16 kernels, fp32-accurate kernels)."""
    from pvnet import init_weights
    from selfplay import SelfPlayRunner
    cfg = NetConfig(15, 15, 2, 225, 5, 512, 8, 1)
    w2 = init_weights(cfg, 11)
    for dtype, leaf in ((torch.bfloat16, "bfloat16"), (torch.float32, "float32")):
        def play(recapture):
            net = PolicyValueNet(cfg, seed=6, device="cuda", dtype=dtype, path="clsfold")
            rec = []
            r = SelfPlayRunner("gomoku", net, 64, 48, size=15, seed=9, leaf_dtype=leaf, recycle=True, use_graph=True, cache_entries=256,
                               cache_shared=True, steps_per_graph=8,
                               on_records=lambda mv, base, pi, q, ch, w, d: rec.append((pi.numpy().copy(), ch.numpy().copy())))
            for _ in range(2):
                r.play_move()
            graphs = r._graph
            ptr = (net._compact.t["cpos_tok"] if dtype == torch.bfloat16 else net._exact["tables"].t["cpos_tok"]).data_ptr()
            assert net.load_state_dict(w2) is True
            assert (net._compact.t["cpos_tok"] if dtype == torch.bfloat16 else net._exact["tables"].t["cpos_tok"]).data_ptr() == ptr
            for h in r.halves:
                h.eng.clear_cache()
            if recapture:
                r._graph = None
            for _ in range(3):
                r.play_move()
            assert recapture or r._graph is graphs
            r.check_error()
            fresh = PolicyValueNet(cfg, weights=w2, device="cuda", dtype=dtype, path="clsfold")
            x = (torch.rand(8, 2, 15, 15, device="cuda") < 0.1).to(dtype)
            assert torch.equal(net(x)[0], fresh(x)[0])
            return rec
        a, b = play(False), play(True)
        assert len(a) == len(b) == 5
        for (pa, ca), (pb, cb) in zip(a, b):
            assert np.array_equal(ca, cb) and pa.tobytes() == pb.tobytes()
