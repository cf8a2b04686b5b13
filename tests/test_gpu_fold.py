"""GPU tests of the patch-pooling embedding kernel (csrc/azk_nn.hip k_embed_fold, include/azk.h azk_nn_embed_fold): the kernel's rows
against the float64 restatement of its formulas (pvnet.PolicyValueNet.forward_fold_u_emulated, itself checked against the plain
forward on the CPU in tests/test_pvnet.py), the batched GEMM that turns them into the value-projected row, the engine-leaf
variant, and the whole evaluator against the float32 forward."""
import ctypes as C

import numpy as np
import pytest
import torch

from pvnet import NetConfig, PolicyValueNet
from test_gpu_nn import _stone_boards

pytestmark = pytest.mark.gpu


def _net(seed=6):
    cfg = NetConfig(15, 15, 2, 225, 5, 512, 8, 1)
    net = PolicyValueNet(cfg, seed=seed, device="cuda", dtype=torch.bfloat16, path="clsfold")
    assert net._foldu is not None
    return cfg, net


def test_embed_fold_rows_vs_float64():
    """Every entry of the kernel's rows against the float64 formulas.  The statistics run on exact products (0/1 patch against fp16
    hi + lo terms) in float32, the outputs are rounded to bf16: 2^-8 relative plus a small absolute term per entry."""
    import azk
    cfg, net = _net()
    T, H, ROW = cfg.tokens, cfg.num_heads, azk.EMBED_FOLD_ROW
    n = 700                                                         # more boards than resident workgroups: the queue is exercised
    x = _stone_boards(n, 4).cuda().to(torch.bfloat16).contiguous()
    sched = azk.new_sched("cuda")
    rows = azk.nn_embed_fold(x, net._foldu, 15, 15, sched)
    torch.cuda.synchronize()
    assert rows.shape == (n, H, ROW) and sched.tolist() == [0, 0]
    u64, bw, invL, pw = net.forward_fold_u_emulated(x.float())
    got = rows.double()
    tol = lambda ref: 2.0 ** -8 * ref.abs() + 1e-6 * ref.abs().max()
    assert bool(((got[:, :, :T] - bw).abs() <= tol(bw)).all()), (got[:, :, :T] - bw).abs().max().item()
    assert bool(((got[:, :, 256:320] - pw).abs() <= tol(pw)).all()), (got[:, :, 256:320] - pw).abs().max().item()
    # 1 / L as two bf16 terms (hi at T and T + 2, the remainder at T + 1)
    assert torch.equal(rows[:, :, T], rows[:, :, T + 2])
    assert bool((((got[:, :, T] + got[:, :, T + 1]) - invL).abs() <= 2.0 ** -15 * invL).all())
    assert bool((rows[:, :, T + 3:256] == 0).all()) and bool((rows[:, :, 320:] == 0).all())
    # tokens no stone reaches carry no weight: the empty board's rows are 1 / L and zeros only
    assert bool((rows[0, :, :T] == 0).all()) and bool((rows[0, :, 256:] == 0).all())
    # the batched GEMM against [D_t; U_all; M_h] gives the value-projected pooled row
    u = torch.empty(n, 512, dtype=torch.bfloat16, device="cuda")
    azk.nn_tail_gemm(rows.view(n, H * ROW), net._foldu.weight, 64, ROW, azk.TAIL_BF16, nbatch=H, a_batch_stride=ROW, out=u)
    err = (u.double() - u64).abs()
    assert err.max().item() < 4e-3 * u64.abs().max().item() and err.mean().item() < 6e-4 * u64.abs().max().item(), (err.max().item(), err.mean().item(), u64.abs().max().item())
    # scheduling does not change a bit; float32 boards give the same rows; a permutation of the batch permutes the rows
    for _ in range(3):
        assert torch.equal(azk.nn_embed_fold(x, net._foldu, 15, 15, sched), rows)
    assert torch.equal(azk.nn_embed_fold(x.float().contiguous(), net._foldu, 15, 15, sched), rows)
    perm = torch.randperm(n, device="cuda")
    assert torch.equal(azk.nn_embed_fold(x[perm].contiguous(), net._foldu, 15, 15, sched), rows[perm])
    # device-side count: rows past it are not produced
    cnt = torch.tensor([301], dtype=torch.int32, device="cuda")
    r2 = torch.full_like(rows, 3.0)
    rc = azk.lib().azk_nn_embed_fold(x.data_ptr(), 0, C.byref(net._foldu.c), r2.data_ptr(), n, 2, 15, 15, cnt.data_ptr(), sched.data_ptr(),
                                     torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    assert torch.equal(r2[:301], rows[:301]) and bool((r2[301:] == 3.0).all())
    torch.cuda.synchronize()
    assert sched.tolist() == [0, 0]


def test_embed_fold_from_engine_leaves_equals_gathered_batch():
    """azk_nn_embed_fold_leaves against azk_step_gather + azk_nn_embed_fold on the same engine state: the same rows, handed out from
    the stone-heavy cost classes down, slots recorded for the next expansion."""
    import azk
    cfg, net = _net()
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
    ref = azk.nn_embed_fold(eng.leaf_boards[:n].contiguous(), net._foldu, 15, 15, sched)
    src = eng.leaf_source()
    hip = C.CDLL("libamdhip64.so")

    def peek(ptr, count, dtype):
        buf = np.empty(count, dtype)
        assert hip.hipMemcpy(buf.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), C.c_size_t(buf.nbytes), 2) == 0
        return buf
    torch.cuda.synchronize()
    gather_slot = peek(src.leaf_slot, G, np.int32)
    fl = peek(src.leaf_flag, G, np.uint8)
    eng.n_leaf.zero_()
    new = azk.nn_embed_fold_leaves(src, net._foldu, sched)
    torch.cuda.synchronize()
    assert int(eng.n_leaf.item()) == n and sched.tolist() == [0, 0]
    new_slot = peek(src.leaf_slot, G, np.int32)
    games = np.nonzero(fl)[0]
    order = sorted(games.tolist(), key=lambda g: (-int(fl[g]), g))
    assert [int(new_slot[g]) for g in order] == list(range(n))
    for g in games[:: max(1, n // 64)]:
        assert torch.equal(new[int(new_slot[g])], ref[int(gather_slot[g])])
    eng.close()


def test_evaluator_on_the_fold_kernel_vs_float32_forward():
    """net(boards) through k_embed_fold + the tail chain against the float32 forward of the same weights, and against the path it
    replaces (k_embed_pool_c + value projection): the fold keeps the statistics in float32 on exact products, so it sits closer to
    the float32 network than the bf16 conv form does."""
    cfg, net = _net(seed=0)
    ref_net = PolicyValueNet(cfg, seed=0, device="cuda", dtype=torch.float32, path="full")
    x = _stone_boards(256, 7).cuda()
    lr, vr = ref_net(x)
    l1, v1 = net(x.to(torch.bfloat16))
    net.use_fold_u = False
    l0, v0 = net(x.to(torch.bfloat16))
    net.use_fold_u = True
    e1, e0 = (l1.float() - lr).abs().max().item(), (l0.float() - lr).abs().max().item()
    assert e1 < 2e-2 and (v1.float().reshape(-1) - vr.reshape(-1)).abs().max().item() < 5e-3, (e1, e0)
    assert e1 <= 1.5 * e0 + 1e-3, (e1, e0)
