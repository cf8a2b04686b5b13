"""CPU checks of the tables azk.EmbedFoldTables hands to k_embed_fold (include/azk.h azk_embed_fold_consts) and of the weight of the
batched GEMM behind it: fragment order, two-term fp16 reconstruction, null-token row, the three 1 / L slots (no GPU needed)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "alpha-zero_amd"))
import azk
from pvnet import NetConfig, PolicyValueNet

CFG = NetConfig(15, 15, 2, 225, patch_size=5, embed_dim=512, num_heads=8, depth=1)


def _tables(exact, monkeypatch):
    # (the binding refuses to work without a GPU - there is no CPU fallback; building the TABLES is plain tensor arithmetic, so the
    #  test lets that one helper through on the CPU)
    monkeypatch.setattr(azk, "_torch", lambda: torch)
    net = PolicyValueNet(CFG, seed=0, path="full")
    r = net.fold_u("cpu")
    return r, azk.EmbedFoldTables(r, 8, 5, 512, "cpu", exact=exact)


def test_fragments_reconstruct_the_quadratic_form_and_the_score_columns(monkeypatch):
    r, ft = _tables(False, monkeypatch)
    inv_g, inv_e = (float(v) for v in ft.t["inv_scales"])
    g = ft.t["g_frag"].double()                                    # [2 (hi, lo)][4 q][2 s][64 lanes][8]
    assert tuple(g.shape) == (2, 4, 2, 64, 8) and ft.t["g_frag"].dtype == torch.float16
    G = torch.zeros(64, 64, dtype=torch.float64)
    for q in range(4):
        for s in range(2):
            for l in range(64):
                for i in range(8):
                    G[32 * s + 8 * (l >> 4) + i, 16 * q + (l & 15)] = (g[0, q, s, l, i] + g[1, q, s, l, i]) * inv_g
    assert (G - r["G"]).abs().max().item() <= 2.0 ** -21 * r["G"].abs().max().item()
    e = ft.t["e_frag"].double()
    E = torch.zeros(64, 16, dtype=torch.float64)
    for s in range(2):
        for l in range(64):
            for i in range(8):
                E[32 * s + 8 * (l >> 4) + i, l & 15] = (e[0, 0, s, l, i] + e[1, 0, s, l, i]) * inv_e
    assert (E - r["ext"]).abs().max().item() <= 2.0 ** -21 * r["ext"].abs().max().item()
    assert bool((E[:, 8:] == 0).all()) and bool((E[50:] == 0).all()) and bool((G[50:] == 0).all())     # beyond the heads / the 50 patch bits


def test_per_token_tables_and_the_null_token(monkeypatch):
    r, ft = _tables(False, monkeypatch)
    T = CFG.tokens
    st, wc, u2 = ft.t["score_tok"], ft.t["wconst_tok"], ft.t["u2_tok"]
    assert tuple(st.shape) == (T + 1, 16) and tuple(wc.shape) == (T + 1, 16) and tuple(u2.shape) == (T + 1, 64)
    np.testing.assert_allclose(st[:T, :8].double().numpy(), r["sct"].numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(st[:T, 15].double().numpy(), r["nt"].numpy(), rtol=1e-6)
    np.testing.assert_allclose(wc[:T, :8].double().numpy(), r["wc"].numpy(), rtol=1e-6)
    np.testing.assert_allclose(wc[:T, 15].double().numpy(), r["rstdc"].numpy(), rtol=1e-6)
    # the null token (pads the last tile): weight exp(-huge) = 0 in every head, constant weight 0, no cross term
    assert bool((st[T, :8] <= -1e29).all()) and float(st[T, 15]) == 512.0 and bool((wc[T, :8] == 0).all()) and bool((u2[T] == 0).all())
    assert bool((ft.t["score_ref"][8:] >= 1e29).all())              # lanes beyond the heads get weight 0
    assert float(wc[:T, :8].max()) <= 1.0 + 1e-6                   # the static softmax reference is the largest constant-token score


def test_gemm_weight_rows(monkeypatch):
    r, ft = _tables(False, monkeypatch)
    T, W = CFG.tokens, ft.weight_f64                              # [H][64][384]
    assert tuple(W.shape) == (8, 64, azk.EMBED_FOLD_ROW)
    assert torch.equal(W[:, :, :T], r["Dtab"].view(T, 8, 64).permute(1, 2, 0))
    ua = r["uall"].view(8, 64)
    assert torch.equal(W[:, :, T], W[:, :, T + 1]) and torch.equal(W[:, :, T], ua.to(torch.bfloat16).double())
    assert (W[:, :, T] + W[:, :, T + 2] - ua).abs().max().item() == 0.0                       # hi + remainder = U_all exactly
    assert torch.equal(W[:, :, 256:320], r["M"].view(8, 64, 64)) and bool((W[:, :, T + 3:256] == 0).all()) and bool((W[:, :, 320:] == 0).all())
    # float32-accurate form: U_all in one slot, the weight as (hi, lo) fp16 planes of azk_nnx_gemm_h
    r2, fx = _tables(True, monkeypatch)
    Wx = fx.weight_f64
    assert torch.equal(Wx[:, :, T], r2["uall"].view(8, 64)) and bool((Wx[:, :, T + 1:256] == 0).all())
    packed = fx.weight                                             # [H][1 g][12 s][4 c][2 planes][4 l4][16 l15][8]
    assert packed.dtype == torch.float16 and packed.shape[0] == 8 and packed.numel() == 8 * 64 * azk.EMBED_FOLD_ROW * 2
    rec = (packed[:, 0, :, :, 0].double() + packed[:, 0, :, :, 1].double()) / azk.GEMM_H_W_SCALE      # [H][s][c][l4][l15][i]
    rec = rec.permute(0, 4, 2, 1, 3, 5).reshape(8, 64, azk.EMBED_FOLD_ROW)                              # row = 4 l15 + c, k = 32 s + 8 l4 + i
    assert (rec - Wx).abs().max().item() <= 2.0 ** -21 * Wx.abs().max().item()
