"""Policy-value evaluator for the self-play engine: the reference's ViT (ai/nn.py:5-84) as a
functional forward over a flat weight table keyed by the reference's own state_dict names, so
checkpoints written by utils.save_model (utils.py:57-61) load unchanged.

    embedding.cls_token [1,1,D]            embedding.pos_embedding [1,T,D]
    embedding.patch_embed.patch_embed.{weight [D,C,k,k], bias}
    blocks.i.{norm1,norm2}.{weight,bias}   blocks.i.attn.{in_proj_weight [3D,D], in_proj_bias, out_proj.weight, out_proj.bias}
    blocks.i.mlp.{0,3}.{weight,bias}       norm.{weight,bias}   policy_head.{weight [A,D],bias}   value_head.{weight [1,D],bias}

Evaluation paths (all compute the SAME function: logits [n,A], value [n] in (-1,1)):
  "full"  every token through every block (what nn.Module.forward does; 1.538 GFLOP/board at the
          15x15 / D=512 / depth 1 config)
  "cls"   the last block only produces what is consumed: K,V for all tokens, Q / out-proj / MLP for the
          cls row alone (x[:,0] is the only row the heads read, nn.py:80-83).  0.254 GFLOP/board.
  "clsfold" as "cls", with the cls query folded through Wk and the value projection applied after the
          softmax-weighted token sum, so K and V are never formed (csrc/azk_nn.hip): 0.022 GFLOP/board,
          one streaming pass over LayerNorm1(tokens).  GPU + bf16 only.
Rectangular boards (Connect4 6x7) get rows*cols+1 tokens; the reference's Net is square-only
(nn.py:26-28), so that case has no reference numerics ("parity unpinned - build-defined").
"""
import math

import torch
import torch.nn.functional as F


class NetConfig:
    def __init__(self, rows, cols, channels, action_dim, patch_size=5, embed_dim=512, num_heads=8, depth=1, dropout=0.0):
        self.rows, self.cols, self.channels, self.action_dim = rows, cols, channels, action_dim
        self.patch_size, self.embed_dim, self.num_heads, self.depth = patch_size, embed_dim, num_heads, depth
        self.dropout = dropout      # Net(..., dropout=) of nn.py:64: active in training mode only (train.py:92), never in self-play

    @property
    def tokens(self):
        return self.rows * self.cols + 1

    def flops_full(self):
        """multiply-add = 2 flops; matches SURVEY 3.2 (1.538 GFLOP at the training config)."""
        T, D, A, C, k = self.tokens, self.embed_dim, self.action_dim, self.channels, self.patch_size
        conv = 2 * (T - 1) * D * C * k * k
        blk = 2 * T * D * 3 * D + 2 * 2 * T * T * D + 2 * T * D * D + 2 * 2 * T * D * 4 * D
        return conv + self.depth * blk + 2 * D * (A + 1)

    def flops_cls(self):
        T, D, A, C, k = self.tokens, self.embed_dim, self.action_dim, self.channels, self.patch_size
        conv = 2 * (T - 1) * D * C * k * k
        blk = 2 * T * D * 3 * D + 2 * 2 * T * T * D + 2 * T * D * D + 2 * 2 * T * D * 4 * D
        last = 2 * T * D * 2 * D + 2 * D * D + 2 * 2 * T * D + 2 * D * D + 2 * 2 * D * 4 * D
        return conv + (self.depth - 1) * blk + last + 2 * D * (A + 1)


def flops_clsfold(cfg):
    """Executed flops of the folded cls path: conv + per token (8 head scores + 8 weighted sums) + cls-row GEMVs."""
    T, D, A, C, k, H = cfg.tokens, cfg.embed_dim, cfg.action_dim, cfg.channels, cfg.patch_size, cfg.num_heads
    conv = 2 * (T - 1) * D * C * k * k
    blk = 2 * T * D * 3 * D + 2 * 2 * T * T * D + 2 * T * D * D + 2 * 2 * T * D * 4 * D
    last = 2 * 2 * T * D * H + 2 * D * D + 2 * D * D + 2 * 2 * D * 4 * D
    return conv + (cfg.depth - 1) * blk + last + 2 * D * (A + 1)


def reference_key_shapes(cfg):
    D, T, C, k, A = cfg.embed_dim, cfg.tokens, cfg.channels, cfg.patch_size, cfg.action_dim
    s = {"embedding.cls_token": (1, 1, D), "embedding.pos_embedding": (1, T, D),
         "embedding.patch_embed.patch_embed.weight": (D, C, k, k), "embedding.patch_embed.patch_embed.bias": (D,)}
    for i in range(cfg.depth):
        b = f"blocks.{i}."
        s.update({b + "norm1.weight": (D,), b + "norm1.bias": (D,), b + "attn.in_proj_weight": (3 * D, D),
                  b + "attn.in_proj_bias": (3 * D,), b + "attn.out_proj.weight": (D, D), b + "attn.out_proj.bias": (D,),
                  b + "norm2.weight": (D,), b + "norm2.bias": (D,), b + "mlp.0.weight": (4 * D, D), b + "mlp.0.bias": (4 * D,),
                  b + "mlp.3.weight": (D, 4 * D), b + "mlp.3.bias": (D,)})
    s.update({"norm.weight": (D,), "norm.bias": (D,), "policy_head.weight": (A, D), "policy_head.bias": (A,),
              "value_head.weight": (1, D), "value_head.bias": (1,)})
    return s


def init_weights(cfg, seed=0):
    """Random initialisation with torch's default initialisers, drawn in module-construction order
    (conv, cls, pos, per block: attention out-proj then in-proj, MLP linears; final heads) so that
    torch.manual_seed(seed) yields the very tensors `Net(...)` would hold under the same seed."""
    D, C, k, A, T = cfg.embed_dim, cfg.channels, cfg.patch_size, cfg.action_dim, cfg.tokens
    torch.manual_seed(seed)
    w = {}
    conv = torch.nn.Conv2d(C, D, kernel_size=k, stride=1, padding=k // 2)
    w["embedding.patch_embed.patch_embed.weight"], w["embedding.patch_embed.patch_embed.bias"] = conv.weight.data, conv.bias.data
    w["embedding.cls_token"] = torch.randn(1, 1, D)
    w["embedding.pos_embedding"] = torch.randn(1, T, D)
    for i in range(cfg.depth):
        b = f"blocks.{i}."
        att = torch.nn.MultiheadAttention(D, cfg.num_heads, batch_first=True)
        l1, l2 = torch.nn.Linear(D, 4 * D), torch.nn.Linear(4 * D, D)
        for nm in ("norm1", "norm2"):
            w[b + nm + ".weight"], w[b + nm + ".bias"] = torch.ones(D), torch.zeros(D)
        w[b + "attn.in_proj_weight"], w[b + "attn.in_proj_bias"] = att.in_proj_weight.data, att.in_proj_bias.data
        w[b + "attn.out_proj.weight"], w[b + "attn.out_proj.bias"] = att.out_proj.weight.data, att.out_proj.bias.data
        w[b + "mlp.0.weight"], w[b + "mlp.0.bias"] = l1.weight.data, l1.bias.data
        w[b + "mlp.3.weight"], w[b + "mlp.3.bias"] = l2.weight.data, l2.bias.data
    w["norm.weight"], w["norm.bias"] = torch.ones(D), torch.zeros(D)
    ph, vh = torch.nn.Linear(D, A), torch.nn.Linear(D, 1)
    w["policy_head.weight"], w["policy_head.bias"] = ph.weight.data, ph.bias.data
    w["value_head.weight"], w["value_head.bias"] = vh.weight.data, vh.bias.data
    return {k_: v.clone() for k_, v in w.items()}


def _same_structure(a, b):
    """Two nests of dicts / tensors / table objects have the same keys, shapes, dtypes and devices."""
    if isinstance(a, torch.Tensor) or isinstance(b, torch.Tensor):
        return (isinstance(a, torch.Tensor) and isinstance(b, torch.Tensor) and a.shape == b.shape and a.dtype == b.dtype
                and a.device == b.device)
    if isinstance(a, dict) or isinstance(b, dict):
        return isinstance(a, dict) and isinstance(b, dict) and a.keys() == b.keys() and all(_same_structure(a[k_], b[k_]) for k_ in a)
    if isinstance(a, (list, tuple)) or isinstance(b, (list, tuple)):
        return type(a) is type(b) and len(a) == len(b) and all(_same_structure(x_, y_) for x_, y_ in zip(a, b))
    if hasattr(a, "t") and isinstance(getattr(a, "t"), dict):           # azk.EmbedPoolTables / EmbedPoolXTables: their tensors live in .t
        return type(a) is type(b) and _same_structure(a.t, b.t)
    return type(a) is type(b) and (a is None or not isinstance(a, (int, float, str, bool)) or a == b)


def _copy_into(dst, src):
    """dst <- src for every tensor of two nests with the same structure, in place (addresses unchanged)."""
    if isinstance(dst, torch.Tensor):
        dst.copy_(src)
    elif isinstance(dst, dict):
        for k_ in dst:
            _copy_into(dst[k_], src[k_])
    elif isinstance(dst, (list, tuple)):
        for d_, s_ in zip(dst, src):
            _copy_into(d_, s_)
    elif hasattr(dst, "t") and isinstance(getattr(dst, "t"), dict):
        _copy_into(dst.t, src.t)


class PolicyValueNet:
    """Callable evaluator: net(boards[n,C,R,Cc]) -> (logits [n,A] float32, value [n,1] float32)."""

    def __init__(self, cfg, weights=None, seed=0, device="cpu", dtype=torch.float32, path="cls"):
        self.cfg = cfg
        w = weights if weights is not None else init_weights(cfg, seed)
        want = reference_key_shapes(cfg)
        if set(w) != set(want):
            raise KeyError(f"state_dict keys differ: missing {sorted(set(want) - set(w))}, unexpected {sorted(set(w) - set(want))}")
        for k_, shp in want.items():
            if tuple(w[k_].shape) != tuple(shp):
                raise ValueError(f"{k_}: shape {tuple(w[k_].shape)} != {shp}")
        self.master = {k_: torch.as_tensor(v).detach().to(torch.float32).cpu().clone() for k_, v in w.items()}
        self.path = path
        self.fast_outputs = False   # True: logits may come back as a bf16 view (the caller converts while copying)
        self.last_value_pre_tanh = False
        self.out_buffers = None     # optional (logits f32 [n,A], values f32 [n]) the fast tail writes into directly
        self.hip_tail = False           # set by _prepare_folded when the hand-written tail kernels cover this configuration
        self.use_hip_tail = False       # True: the cls-row tail on azk_nn_gemm_rows (every launch honours the live count; measured
                                        # 87 us vs 82 us for the hipBLASLt tail at 2048 rows / 1150 live, so the library GEMMs stay the default)
        self.leaf_source = None         # azk.LeafSource of the engine being stepped: the fused kernel reads the pending leaves itself
        self.fuse_ln_heads = True       # final LayerNorm + heads + finalize as one hand-written launch (needs hip_tail's packed weights)
        self.fused_embed_pool = False   # set by _prepare_folded when azk_nn_embed_pool covers this configuration
        self.chain_tail = False         # set by _prepare_folded when azk_nn_tail_gemm covers this configuration
        self.use_chain_tail = True
        self.use_lds_tail = True          # the two wide links of the chain tail LDS-staged (csrc/azk_tail.hip); False: k_tail_gemm for all five
        self._tail_ws = {}              # workspaces of the tail chain, one per board source, sized for the largest batch seen
        self._tail_ws_retired = []      # outgrown workspaces, kept alive (captured graphs may hold their addresses)
        self._compact = None            # azk.EmbedPoolTables when the compacting kernel covers this configuration (static softmax reference)
        self.use_compact = True
        self._foldu = None              # azk.EmbedFoldTables when the patch-pooling kernel (k_embed_fold) covers this configuration
        self.use_fold_u = True
        self._scheds = {}               # work-queue words of the compacting kernel, one buffer per board source (= per stepping stream)
        self.kernel_timers = None   # optional timers with start()/stop() (HIP events): (embed+pool, tail) on the fused path, (embed, pool, tail) otherwise
        self.live_count = None      # optional int32 CUDA tensor: number of valid rows at the head of the batch (graph stepping)
        self.to(device, dtype)

    # ---- state_dict compatibility with utils.save_model / load_model (utils.py:57-69) ----------------
    def state_dict(self):
        return {k_: v.clone() for k_, v in self.master.items()}

    @classmethod
    def from_state_dict(cls, cfg, sd, **kw):
        return cls(cfg, weights=sd, **kw)

    def load_state_dict(self, sd, in_place=True):
        """nn.Module.load_state_dict for the reference's keys (main.py:59 `older_model.load_state_dict(...)`: promotion).
        in_place: every device buffer the kernels read - weight copies, folded constants, per-token tables, packed fragments - is
        recomputed from the new weights (on the device) and COPIED INTO THE EXISTING TENSORS, so their addresses do not change
        and a captured step graph keeps working without re-capture.  Returns True when that was possible; False when the new
        weights changed which kernels apply (e.g. the static softmax reference no longer fits) and the structures were rebuilt -
        the owner of a captured graph must then re-capture.  The caller clears the engine's eval cache (main.py:55-57)."""
        want = reference_key_shapes(self.cfg)
        for k_, shape in want.items():
            assert k_ in sd and tuple(sd[k_].shape) == tuple(shape), k_
        names = ("w", "_hip", "_fold", "_compact", "_exact", "_foldu", "_blocks")
        old = {n_: getattr(self, n_, None) for n_ in names}
        old_flags = (self.fused_embed_pool, self.chain_tail, self.hip_tail, self._gelu_epilogue)
        self.master = {k_: torch.as_tensor(sd[k_]).detach().to("cpu", torch.float32).clone() for k_ in want}
        self.to(self.device, self.dtype)
        if not in_place:
            return False
        new = {n_: getattr(self, n_, None) for n_ in names}
        if old_flags != (self.fused_embed_pool, self.chain_tail, self.hip_tail, self._gelu_epilogue) or not _same_structure(old, new):
            return False
        _copy_into(old, new)
        for n_ in names:
            setattr(self, n_, old[n_])
        return True

    def to(self, device, dtype=None):
        self.device = torch.device(device)
        self.dtype = dtype or self.dtype
        self.w = {k_: v.to(self.device, self.dtype) for k_, v in self.master.items()}
        D, H = self.cfg.embed_dim, self.cfg.num_heads
        self.scale = 1.0 / math.sqrt(D // H)
        self._hip = None
        self._fold = None
        self._gelu_epilogue = False
        self._exact = None
        self._foldu = None
        self._blocks = None
        if self.device.type == "cuda" and self.dtype == torch.bfloat16:
            self._prepare_hip_embed()
            if self._hip is not None:
                self._prepare_folded()
                if self.cfg.depth > 1 or not self.chain_tail:
                    # every network outside the benchmark shape (depth 1, D = 512, whose cls path has its own kernels): the full-token
                    # blocks (depth > 1) and the cls path of the last block as hand-written GEMMs (csrc/azk_block.hip)
                    self._prepare_blocks()
        elif self.device.type == "cuda" and self.dtype == torch.float32:
            self._prepare_exact()
        return self

    def exact_fold(self, dev=None):
        """Operands of the fp32-accurate folded cls path (csrc/azk_nnx.hip: k_embed_pool_x, k_gemm_x) before packing: the same folds
        as _prepare_folded / _prepare_compact - cls query through W_k, LayerNorm affines into the consuming weights, constant-token
        softmax terms - carried out in FLOAT64 from the float32 master weights and rounded once.  Returns None when the
        configuration is not covered (depth 1, D = 512, 4 / 8 heads, <= 256 tokens, conv K <= 64, a static softmax reference
        that cannot underflow); then the torch float32 paths stay."""
        cfg, m = self.cfg, self.master
        D, H, T, A = cfg.embed_dim, cfg.num_heads, cfg.tokens, cfg.action_dim
        kreal = cfg.channels * cfg.patch_size ** 2
        kp = (kreal + 31) // 32 * 32
        if not (cfg.depth == 1 and D == 512 and H in (4, 8) and T <= 256 and kp <= 64 and cfg.channels in (2, 3) and cfg.patch_size in (3, 5)
                and A + 1 <= 256):
            return None
        dev = self.device if dev is None else torch.device(dev)
        dh, eps = D // H, 1e-5
        dd = lambda k_: m[k_].to(dev, torch.float64)
        Wc = dd("embedding.patch_embed.patch_embed.weight").reshape(D, kreal)
        cpos = dd("embedding.pos_embedding")[0].clone()
        cpos[0] += dd("embedding.cls_token")[0, 0]
        cpos[1:] += dd("embedding.patch_embed.patch_embed.bias")
        b = "blocks.0."
        g1, b1 = dd(b + "norm1.weight"), dd(b + "norm1.bias")
        Wi, bi = dd(b + "attn.in_proj_weight"), dd(b + "attn.in_proj_bias")
        x0 = cpos[0]
        h0 = F.layer_norm(x0, (D,), g1, b1, eps)
        q = (Wi[:D] @ h0 + bi[:D]).view(H, dh)
        m_n = torch.einsum("he,hed->hd", q, Wi[D:2 * D].view(H, dh, D)) * (1.0 / math.sqrt(dh)) * g1      # [H, D]: scale Wk_h^T q_h, gamma folded
        bound = math.sqrt(D) * m_n.norm(dim=1)
        if float(bound.max()) > 40.0:
            return None                                       # the static softmax reference would underflow
        wt_ext = torch.zeros(D + 16, kp, dtype=torch.float64, device=dev)
        wt_ext[:D, :kreal] = Wc
        wt_ext[D:D + H, :kreal] = m_n @ Wc                   # x . m' = (m'^T Wconv) . patch + m' . cpos
        wt_ext[D + 15, :kreal] = Wc.mean(0)                  # row mean as one more GEMM column
        sc = torch.zeros(T, 16, dtype=torch.float64, device=dev)
        sc[:, :H] = cpos @ m_n.t()
        sc[:, 15] = cpos.mean(1)
        ms = torch.zeros(16, dtype=torch.float64, device=dev)
        ms[:H] = m_n.sum(1)
        # the kernel reads float32 tables: the constant-token terms are derived from the ROUNDED tables, with the kernel's formulas
        cpos32, sc32, ms32 = cpos.float(), sc.float(), ms.float()
        mean = sc32[:, 15].double()
        var = ((cpos32.double() ** 2).sum(1) / D - mean * mean).clamp_min(0.0)
        rstd = 1.0 / torch.sqrt(var + eps)
        xnc32 = (cpos32.double() * rstd[:, None] - (mean * rstd)[:, None]).float()
        s_c = rstd[:, None] * (sc32.double() - mean[:, None] * ms32.double()[None, :])
        # static softmax reference per head = the largest score of a constant (empty-patch) token: their weights are then in (0, 1]
        # (representable as two fp16 terms for the kernel's constant-token MFMA), a stone-touched token's weight is at most
        # exp(2 * bound) <= e^80 in float32 (|s| <= bound for every token; bound <= 40 checked above)
        ref = torch.zeros(16, dtype=torch.float64, device=dev)
        ref[:H] = s_c[:, :H].max(0).values
        ref32 = ref.float()
        wc = torch.zeros(T, 16, dtype=torch.float64, device=dev)
        wc[:, :H] = torch.exp(s_c[:, :H] - ref32.double()[None, :H])
        wc32 = wc.float()
        null_sc = torch.zeros(1, 16, device=dev)
        null_sc[:, :H] = -1e30
        zrow = torch.zeros(1, D, device=dev)
        r = dict(wt_ext=wt_ext, cpos_tok=torch.cat([cpos32, zrow]), score_tok=torch.cat([sc32, null_sc]),
                 wconst_tok=torch.cat([wc32, torch.zeros(1, 16, device=dev)]), xnconst_tok=torch.cat([xnc32, zrow]),
                 z_all=(wc32.double().t() @ xnc32.double()).float(), l_all=wc32.double().sum(0).float(), score_msum=ms32, score_ref=ref32)
        # ---- cls-row tail: u_h = Wv'_h z_h;  x1 = Wo u + bias1;  hh = GELU(LN2(x1) W0'^T + b0');  x2 = x1 + b3 + W3 hh;  heads(LNf(x2)) ----
        Wv, bv = Wi[2 * D:].view(H, dh, D), bi[2 * D:]
        bvn = bv + (Wv @ b1).reshape(-1)
        Wo, bo = dd(b + "attn.out_proj.weight"), dd(b + "attn.out_proj.bias")
        g2, b2 = dd(b + "norm2.weight"), dd(b + "norm2.bias")
        W0, b0 = dd(b + "mlp.0.weight"), dd(b + "mlp.0.bias")
        gf, bf_ = dd("norm.weight"), dd("norm.bias")
        Wh = torch.zeros(256, D, dtype=torch.float64, device=dev)
        bh = torch.zeros(256, dtype=torch.float64, device=dev)
        Wh[:A], bh[:A] = dd("policy_head.weight"), dd("policy_head.bias")
        Wh[A], bh[A] = dd("value_head.weight")[0], dd("value_head.bias")[0]
        r.update(Wvn=(Wv * g1).float(),                       # sum_t a_t (g xn_t + b1) = g (sum_t a_t xn_t) + b1 since sum_t a_t = 1
                 Wo=Wo.float(), bias1=(x0 + bo + Wo @ bvn).float().contiguous(), W0G=(W0 * g2[None, :]).float(),
                 b0G=(W0 @ b2 + b0).float().contiguous(), W3=dd(b + "mlp.3.weight").float(), b3=dd(b + "mlp.3.bias").float().contiguous(),
                 WhG=(Wh * gf[None, :]).float(), bhG=(Wh @ bf_ + bh).float().contiguous())
        return r

    def _prepare_exact(self):
        """The fp32-accurate kernels' tables, or nothing: weights outside the fixed fp16 (hi, lo) scales of those tables (conv weight
        x 4096, tail weights x 256, constant-token softmax weights <= 1: the packers' range assertions) leave `_exact` None, exactly as a
        configuration exact_fold() does not cover - the torch float32 forward then runs, and an in-place promotion reports False."""
        try:
            self._prepare_exact_tables()
        except AssertionError:
            self._exact, self.fused_embed_pool = None, False

    def _prepare_exact_tables(self):
        import azk
        cfg = self.cfg
        self.fused_embed_pool = False
        r = self.exact_fold()
        if r is None:
            return
        H = cfg.num_heads
        e = dict(tables=azk.EmbedPoolXTables({k_: r[k_] for k_ in ("wt_ext", "cpos_tok", "score_tok", "wconst_tok", "xnconst_tok", "z_all", "l_all",
                                                                   "score_msum", "score_ref")}, H, cfg.patch_size, cfg.embed_dim, 1e-5))
        # the patch-pooling form of embedding + pooling + value projection in float32-accurate arithmetic (k_embed_fold<EX> + one batched
        # link of the (hi, lo)-plane tail); tables in float64 from the master weights
        rf = self.fold_u()
        e["foldu"] = azk.EmbedFoldTables(rf, H, cfg.patch_size, cfg.embed_dim, self.device, exact=True) if rf is not None else None
        e["WvX"] = torch.cat([azk.pack_linear_weight_x(r["Wvn"][h]).reshape(-1) for h in range(H)])
        for k_ in ("Wo", "W0G", "W3", "WhG"):
            e[k_ + "X"] = azk.pack_linear_weight_x(r[k_])
        for k_ in ("bias1", "b0G", "b3", "bhG"):
            e[k_] = r[k_]
        # the same weights as fp16 (hi, lo) planes for the fp16-pipe tail (azk_nnx_gemm_h) + the column sums its LayerNorm epilogue needs
        wvh = [azk.pack_linear_weight_h(r["Wvn"][h]) for h in range(H)]
        e["WvH"] = torch.cat([w_.reshape(-1) for w_, _ in wvh])
        for k_ in ("Wo", "W0G", "W3", "WhG"):
            e[k_ + "H"], e[k_ + "H_csum"] = azk.pack_linear_weight_h(r[k_])
        self.exact_tail = "h16"                                # "h16": fp16 (hi, lo) planes on the fp16 matrix pipe; "f32": v_mfma_f32_16x16x4_f32
        self._exact = e
        self.fused_embed_pool = True                          # the step graph may hand the engine's pending leaves straight to the kernel

    def forward_exact_emulated(self, x, r=None):
        """What k_embed_pool_x + k_gemm_x compute, step by step from the same folded tables, in float64 torch on any device: the
        checker of the kernels (tests) and of the fold itself (against the plain forward, on the CPU)."""
        cfg = self.cfg
        r = r or self.exact_fold(x.device)
        D, H, T, A, k = cfg.embed_dim, cfg.num_heads, cfg.tokens, cfg.action_dim, cfg.patch_size
        n = x.shape[0]
        f8 = lambda t: t.double()
        cols = F.unfold(x.double(), kernel_size=k, padding=k // 2).transpose(1, 2)               # [n, R*C, C*k*k] patch bits
        kreal = cols.shape[2]
        patch = torch.zeros(n, T, r["wt_ext"].shape[1], dtype=torch.float64, device=x.device)
        patch[:, 1:, :kreal] = cols                                                              # token 0 (cls) has no patch
        dirty = patch.abs().sum(2) > 0                                                           # [n, T]
        xx = patch @ r["wt_ext"][:D].t() + f8(r["cpos_tok"][:T])
        ext = patch @ r["wt_ext"][D:].t() + f8(r["score_tok"][:T])                               # [n, T, 16]
        mean = ext[..., 15]
        rstd = 1.0 / torch.sqrt(((xx * xx).sum(2) / D - mean * mean).clamp_min(0.0) + 1e-5)
        xn = (xx - mean[..., None]) * rstd[..., None]
        sco = rstd[..., None] * (ext - mean[..., None] * f8(r["score_msum"]))
        w = torch.exp(sco - f8(r["score_ref"]))[..., :H]                                         # [n, T, H]
        wc, xnc = f8(r["wconst_tok"][:T, :H]), f8(r["xnconst_tok"][:T])
        dm = dirty[..., None].double()
        Z = f8(r["z_all"][:H])[None] + torch.einsum("nth,ntd->nhd", w * dm, xn) - torch.einsum("nth,td->nhd", dm * wc[None], xnc)
        Lh = f8(r["l_all"][:H])[None] + ((w - wc[None]) * dm).sum(1)
        z = Z / Lh[..., None]                                                                    # [n, H, D]
        u = torch.einsum("nhd,hed->nhe", z, f8(r["Wvn"])).reshape(n, D)
        x1 = u @ f8(r["Wo"]).t() + f8(r["bias1"])
        ln = lambda t: (t - t.mean(1, keepdim=True)) / torch.sqrt(t.var(1, unbiased=False, keepdim=True) + 1e-5)
        hh = F.gelu(ln(x1) @ f8(r["W0G"]).t() + f8(r["b0G"]))
        x2 = x1 + hh @ f8(r["W3"]).t() + f8(r["b3"])
        out = ln(x2) @ f8(r["WhG"]).t() + f8(r["bhG"])
        return out[:, :A].float(), torch.tanh(out[:, A:A + 1]).float(), z.float()

    def fold_u(self, dev=None):
        """Operands of the patch-pooling form of the cls path (csrc/azk_nn.hip k_embed_fold), float64, from the float32 master weights.
        With x_t = Wc p_t + cpos_t (p_t the 0/1 patch of token t) everything LayerNorm1 and the cls attention need of a token is a
        function of its <= 64 patch bits, and the pooled row is LINEAR in x_t, so the D-wide token rows are never formed:
            x_t - mean(x_t) = Wt p_t + ct_t                 Wt = Wc - column means, ct_t = cpos_t - mean(cpos_t)
            D var_t        = p_t' G p_t + 2 U_t . p_t + n_t   G = Wt' Wt, U_t = Wt' ct_t, n_t = |ct_t|^2
            s_t[h]         = rstd_t (S_h . p_t + sc_t[h])     S_h = Wt' m'_h, sc_t[h] = m'_h . ct_t        (m'_h: exact_fold's m_n)
            u_h = Wv'_h z_h = (1 / L_h) sum_t a_t[h] (M_h p_t + D_t[h]),   a = w rstd,  M_h = Wv'_h Wt,  D_t[h] = Wv'_h ct_t
        u is the output of the tail's first link (nn.py:54-56 up to the value projection).  The kernel emits, per board and head,
        the token weights b_t = (a_t - ac_t) / L (ac: the token as an empty-patch constant; zero for tokens no stone reaches), 1 / L,
        and the pooled patch sum_t a_t p_t / L; one batched GEMM against [D_t[h]; U_all[h]; M_h] then gives u.  Returns None when the
        configuration is not covered."""
        cfg, m = self.cfg, self.master
        D, H, T = cfg.embed_dim, cfg.num_heads, cfg.tokens
        kreal = cfg.channels * cfg.patch_size ** 2
        if not (cfg.depth == 1 and D == 512 and H in (4, 8) and D // H == 64 and T + 3 <= 256 and kreal <= 64 and cfg.channels in (2, 3)
                and cfg.patch_size in (3, 5)):
            return None
        dev = self.device if dev is None else torch.device(dev)
        dh, eps, kp = D // H, 1e-5, 64
        dd = lambda k_: m[k_].to(dev, torch.float64)
        Wc = dd("embedding.patch_embed.patch_embed.weight").reshape(D, kreal)
        cpos = dd("embedding.pos_embedding")[0].clone()
        cpos[0] += dd("embedding.cls_token")[0, 0]
        cpos[1:] += dd("embedding.patch_embed.patch_embed.bias")
        b = "blocks.0."
        g1, b1 = dd(b + "norm1.weight"), dd(b + "norm1.bias")
        Wi, bi = dd(b + "attn.in_proj_weight"), dd(b + "attn.in_proj_bias")
        h0 = F.layer_norm(cpos[0], (D,), g1, b1, eps)
        q = (Wi[:D] @ h0 + bi[:D]).view(H, dh)
        m_n = torch.einsum("he,hed->hd", q, Wi[D:2 * D].view(H, dh, D)) * (1.0 / math.sqrt(dh)) * g1      # [H, D]
        if float((math.sqrt(D) * m_n.norm(dim=1)).max()) > 40.0:
            return None
        Wvn = (Wi[2 * D:].view(H, dh, D) * g1).reshape(D, D)                                            # row h*dh + j = Wv'_h[j]
        Wt = torch.zeros(D, kp, dtype=torch.float64, device=dev)
        Wt[:, :kreal] = Wc - Wc.mean(0, keepdim=True)
        ct = cpos - cpos.mean(1, keepdim=True)
        G = Wt.t() @ Wt                                                                                  # [kp, kp]
        U2 = 2.0 * (ct @ Wt)                                                                             # [T, kp]
        nt = (ct * ct).sum(1)                                                                            # [T]
        ext = torch.zeros(kp, 16, dtype=torch.float64, device=dev)
        ext[:, :H] = Wt.t() @ m_n.t()
        sct = ct @ m_n.t()                                                                               # [T, H]
        Dtab = ct @ Wvn.t()                                                                              # [T, D]
        M = Wvn @ Wt                                                                                     # [D, kp]
        rstdc = 1.0 / torch.sqrt(nt / D + eps)
        s_c = rstdc[:, None] * sct
        ref = s_c.max(0).values                                                                          # static softmax reference per head
        wc = torch.exp(s_c - ref[None])
        ac = wc * rstdc[:, None]
        uall = torch.einsum("th,thj->hj", ac, Dtab.view(T, H, dh)).reshape(D)
        return dict(G=G, ext=ext, U2=U2, nt=nt, sct=sct, Dtab=Dtab, M=M, rstdc=rstdc, ref=ref, wc=wc, uall=uall, lall=wc.sum(0), kreal=kreal)

    def forward_fold_u_emulated(self, x, r=None):
        """u (float64 [n, D]) by fold_u's formulas, step by step in torch on any device: checks the fold against the plain forward on
        the CPU and the kernel against the fold on the GPU.  Also returns the kernel's outputs before rounding: b / L [n, H, T],
        1 / L [n, H], pooled patch / L [n, H, kp]."""
        cfg = self.cfg
        r = r or self.fold_u(x.device)
        D, H, T, k = cfg.embed_dim, cfg.num_heads, cfg.tokens, cfg.patch_size
        n, kp = x.shape[0], r["G"].shape[0]
        cols = F.unfold(x.double(), kernel_size=k, padding=k // 2).transpose(1, 2)
        P = torch.zeros(n, T, kp, dtype=torch.float64, device=x.device)
        P[:, 1:, :cols.shape[2]] = cols
        var = (torch.einsum("ntk,kl,ntl->nt", P, r["G"], P) + (P * r["U2"][None]).sum(2) + r["nt"][None]) / D
        rstd = 1.0 / torch.sqrt(var.clamp_min(0.0) + 1e-5)
        s = rstd[..., None] * (P @ r["ext"][:, :H] + r["sct"][None])
        w = torch.exp(s - r["ref"][None, None])
        a = w * rstd[..., None]
        L = r["lall"][None] + (w - r["wc"][None]).sum(1)                                                 # [n, H]
        bw = (a - (r["wc"] * r["rstdc"][:, None])[None]) / L[:, None, :]                                 # [n, T, H]
        pw = torch.einsum("nth,ntk->nhk", a, P) / L[..., None]                                           # [n, H, kp]
        dh = D // H
        u = (torch.einsum("nth,thj->nhj", bw, r["Dtab"].view(T, H, dh)) + r["uall"].view(1, H, dh) / L[..., None]
             + torch.einsum("nhk,hjk->nhj", pw, r["M"].view(H, dh, kp))).reshape(n, D)
        return u, bw.transpose(1, 2), 1.0 / L, pw

    def forward_exact(self, x):
        """The fp32-accurate folded cls path: boards (or the engine's pending leaves) -> z float32 [n, H, D] -> logits, tanh(value)."""
        import azk
        cfg, e = self.cfg, self._exact
        if (e.get("foldu") is not None and self.use_fold_u and getattr(self, "exact_tail", "f32") == "h16"
                and (self.leaf_source is None or self.leaf_source.n_games <= azk.EMBED_FOLD_MAX_SLOTS)):
            # the token rows are never formed (k_embed_fold<EX>): float32 rows of token weights / L, 1 / L, pooled patch / L per head
            if self.leaf_source is not None:
                z = azk.nnx_embed_fold_leaves(self.leaf_source, e["foldu"], self._sched_for(self.leaf_source), timers=self.kernel_timers)
            else:
                if x.dtype not in (torch.bfloat16, torch.float32):
                    x = x.float()
                z = azk.nnx_embed_fold(x.contiguous(), e["foldu"], cfg.rows, cfg.cols, self._sched_for(None), count=self.live_count,
                                       timers=self.kernel_timers)
        elif self.leaf_source is not None:
            z = azk.nnx_embed_pool_leaves(self.leaf_source, e["tables"], self._sched_for(self.leaf_source), timers=self.kernel_timers)
        else:
            if x.dtype not in (torch.bfloat16, torch.float32):
                x = x.float()
            z = azk.nnx_embed_pool(x.contiguous(), e["tables"], cfg.rows, cfg.cols, self._sched_for(None), count=self.live_count,
                                   timers=self.kernel_timers)
        return self.tail_fast(z)

    def tail_exact_h(self, z):
        """The fp32-accurate cls-row tail on the fp16 matrix pipe (csrc/azk_nnx.hip k_gemm_h): every operand as two fp16 terms (22
        bits), activations handed from link to link as (hi, lo) planes, LayerNorm in the consuming epilogue; five launches, each
        honouring the device-side live count (nn.py:54-60, 78-83 for the row the heads read)."""
        import azk
        cfg, e = self.cfg, self._exact
        n, A, D, H = z.shape[0], cfg.action_dim, cfg.embed_dim, cfg.num_heads
        dev, cnt = z.device, self.live_count
        key = ("h", id(self.leaf_source) if self.leaf_source is not None else None)
        ws = self._tail_ws.get(key)
        if ws is None or ws["rows"] < n:
            rows = n if ws is None else max(n, 2 * ws["rows"])
            if ws is not None:
                self._tail_ws_retired.append(ws)                 # never freed: captured graphs may hold these addresses
            f16, f32 = dict(dtype=torch.float16, device=dev), dict(dtype=torch.float32, device=dev)
            ws = dict(rows=rows, u=torch.empty((2, rows, D), **f16), x1=torch.empty((2, rows, D), **f16), x1f=torch.empty((rows, D), **f32),
                      hh=torch.empty((2, rows, 4 * D), **f16), x2=torch.empty((2, rows, D), **f16),
                      st1=torch.empty((rows, D // 64, 2), **f32), st2=torch.empty((rows, D // 64, 2), **f32))
            self._tail_ws[key] = ws
        pl = lambda k_: (ws[k_][0, :n], ws[k_][1, :n])
        if getattr(self, "_exact_overflow", None) is None or self._exact_overflow.device != dev:
            self._exact_overflow = torch.zeros(1, dtype=torch.int32, device=dev)      # sticky: an activation left the fp16 planes' range (check_exact_range)
        of = self._exact_overflow
        G = lambda *a_, **k_: azk.nnx_gemm_h(*a_, overflow=of, **k_)
        if z.shape[-1] == azk.EMBED_FOLD_ROW:      # k_embed_fold's float32 rows against [D_t; U_all; M_h]: the value-projected row directly
            kf = azk.EMBED_FOLD_ROW
            self._launch(G, z.view(n, H * kf), e["foldu"].weight, D // H, kf, azk.TAIL_BF16, nbatch=H, a_batch_stride=kf, out=pl("u"), count=cnt)
        else:
            self._launch(G, z.view(n, H * D), e["WvH"], D // H, D, azk.TAIL_BF16, nbatch=H, a_batch_stride=D, out=pl("u"), count=cnt)
        self._launch(G, pl("u"), e["WoH"], D, D, azk.TAIL_BF16, bias=e["bias1"], out=pl("x1"), out_f32=ws["x1f"][:n], stats_out=ws["st1"][:n], count=cnt)
        lds = getattr(self, "use_lds_tail", True)        # the two wide links LDS-staged (azk_nnx_gemm_h_lds, csrc/azk_tail.hip): same chains, same epilogue
        self._launch(G, pl("x1"), e["W0GH"], 4 * D, D, azk.TAIL_GELU, bias=e["b0G"], col_sums=e["W0GH_csum"], out=pl("hh"), a_stats=ws["st1"][:n], count=cnt, lds=lds)
        self._launch(G, pl("hh"), e["W3H"], D, 4 * D, azk.TAIL_RESID, bias=e["b3"], resid=ws["x1f"][:n], out=pl("x2"), stats_out=ws["st2"][:n], count=cnt, lds=lds)
        if self.out_buffers is not None:
            lb, vb = self.out_buffers
        else:
            lb = torch.empty((n, A), dtype=torch.float32, device=dev)
            vb = torch.empty(n, dtype=torch.float32, device=dev)
        self._launch(G, pl("x2"), e["WhGH"], 256, D, azk.TAIL_HEADS, bias=e["bhG"], col_sums=e["WhGH_csum"], a_stats=ws["st2"][:n], logits=lb, values=vb,
                     action_dim=A, count=cnt)
        return lb, (vb if self.out_buffers is not None else vb[:, None])

    def check_exact_range(self):
        """The fp32-accurate tail hands its activations from link to link as fp16 (hi, lo) planes of x * 16: |x| >= 4094 does not fit.  The
        kernels set a sticky device flag when that happens (the outputs are then meaningless); this raises if it ever did.  Synchronises."""
        f = getattr(self, "_exact_overflow", None)
        if f is not None and int(f.item()) != 0:
            raise FloatingPointError("fp32-accurate tail: an activation left the range of its fp16 (hi, lo) planes (|x| >= 4094): use the torch "
                                     "float32 forward (path='cls') for this network")

    def tail_exact(self, z):
        """The cls-row tail as five float32 launches (csrc/azk_nnx.hip k_gemm_x), each honouring the device-side live count
        (nn.py:54-60, 78-83 for the row the heads read)."""
        import azk
        if getattr(self, "exact_tail", "f32") == "h16":
            return self.tail_exact_h(z)
        cfg, e = self.cfg, self._exact
        n, A, D, H = z.shape[0], cfg.action_dim, cfg.embed_dim, cfg.num_heads
        dev, cnt = z.device, self.live_count
        key = ("x", id(self.leaf_source) if self.leaf_source is not None else None)
        ws = self._tail_ws.get(key)
        if ws is None or ws["rows"] < n:
            rows = n if ws is None else max(n, 2 * ws["rows"])
            if ws is not None:
                self._tail_ws_retired.append(ws)                 # never freed: captured graphs may hold these addresses
            f32 = dict(dtype=torch.float32, device=dev)
            ws = dict(rows=rows, u=torch.empty((rows, D), **f32), x1=torch.empty((rows, D), **f32), hh=torch.empty((rows, 4 * D), **f32),
                      x2=torch.empty((rows, D), **f32), st1=torch.empty((rows, D // 64, 2), **f32), st2=torch.empty((rows, D // 64, 2), **f32))
            self._tail_ws[key] = ws
        ws = {k_: (v[:n] if k_ != "rows" else v) for k_, v in ws.items()}
        self._launch(azk.nnx_gemm, z.view(n, H * D), e["WvX"], D // H, D, azk.TAIL_BF16, nbatch=H, a_batch_stride=D, out=ws["u"], count=cnt)
        self._launch(azk.nnx_gemm, ws["u"], e["WoX"], D, D, azk.TAIL_BF16, bias=e["bias1"], out=ws["x1"], stats_out=ws["st1"], count=cnt)
        self._launch(azk.nnx_gemm, ws["x1"], e["W0GX"], 4 * D, D, azk.TAIL_GELU, bias=e["b0G"], out=ws["hh"], a_stats=ws["st1"], count=cnt)
        self._launch(azk.nnx_gemm, ws["hh"], e["W3X"], D, 4 * D, azk.TAIL_RESID, bias=e["b3"], resid=ws["x1"], out=ws["x2"], stats_out=ws["st2"], count=cnt)
        if self.out_buffers is not None:
            lb, vb = self.out_buffers
        else:
            lb = torch.empty((n, A), dtype=torch.float32, device=dev)
            vb = torch.empty(n, dtype=torch.float32, device=dev)
        self._launch(azk.nnx_gemm, ws["x2"], e["WhGX"], 256, D, azk.TAIL_HEADS, bias=e["bhG"], a_stats=ws["st2"], logits=lb, values=vb, action_dim=A, count=cnt)
        return lb, (vb if self.out_buffers is not None else vb[:, None])

    def _prepare_folded(self):
        """Operands of the folded cls-row attention of the LAST block (csrc/azk_nn.hip k_cls_attn):
        scores[h][t] = xhat_t . m_h + c_h with m_h = scale * Wk_h^T q_h, c_h = scale * q_h . bk_h; head output =
        Wv_h (sum_t a[h][t] xhat_t) + bv_h.  For depth 1 the cls query is input independent (x[:,0] = cls + pos[0]),
        so m and c are constants of the weights."""
        cfg, m = self.cfg, self.master
        D, H = cfg.embed_dim, cfg.num_heads
        dh = D // H
        if (D, H) not in ((512, 8), (512, 4), (256, 8), (256, 4), (128, 4), (128, 8)):
            return
        b = f"blocks.{cfg.depth - 1}."
        Wi, bi = m[b + "attn.in_proj_weight"], m[b + "attn.in_proj_bias"]
        dev = self.device
        f = dict(Wq=Wi[:D].to(dev), bq=bi[:D].to(dev), Wk=Wi[D:2 * D].reshape(H, dh, D).to(dev),
                 bk=bi[D:2 * D].reshape(H, dh).to(dev),
                 WvT=Wi[2 * D:].reshape(H, dh, D).transpose(1, 2).contiguous().to(dev, torch.bfloat16),   # [H, D, dh]
                 bv=bi[2 * D:].to(dev, torch.bfloat16))
        if cfg.depth == 1:
            x0 = self._hip["cpos"][0]
            h0 = F.layer_norm(x0, (D,), self._hip["ln_w"], self._hip["ln_b"], 1e-5)
            q = (F.linear(h0, f["Wq"], f["bq"])).view(H, dh)
            f["m"] = (torch.einsum("he,hed->hd", q, f["Wk"]) * self.scale).contiguous()
            f["c"] = ((q * f["bk"]).sum(1) * self.scale).contiguous()
            # LayerNorm affine folded out of the kernels: with xn = (x - mean) * rstd and xhat = gamma * xn + beta,
            #   xhat . m + c = xn . (gamma * m) + (beta . m + c);   sum_t a_t xhat_t = gamma * (sum_t a_t xn_t) + beta
            g, bta = self._hip["ln_w"], self._hip["ln_b"]
            f["m_n"] = (f["m"] * g).contiguous()
            f["c_n"] = (f["c"] + f["m"] @ bta).contiguous()
            # the scores ride on the conv GEMM as 16 extra output columns: x . m' = (m'^T Wconv) . patch + m' . cpos
            hp = self._hip
            kreal = cfg.channels * cfg.patch_size ** 2
            Wc = m["embedding.patch_embed.patch_embed.weight"].reshape(D, kreal).to(dev)
            wt_ext = torch.zeros(D + 16, hp["wt"].shape[1], device=dev)
            wt_ext[:D] = hp["wt"].float()
            wt_ext[D:D + H, :kreal] = f["m_n"] @ Wc
            wt_ext[D + 15] = hp["wt"].float().mean(0)            # column 15: the row mean (1/D) sum_d x[t][d] as a GEMM output
            f["wt_ext"] = wt_ext.to(torch.bfloat16).contiguous()
            sc = torch.zeros(cfg.tokens, 16, device=dev)
            sc[:, :H] = hp["cpos"] @ f["m_n"].t()
            sc[:, 15] = hp["cpos"].mean(1)
            f["score_cpos"] = sc.contiguous()
            ms = torch.zeros(16, device=dev)
            ms[:H] = f["m_n"].sum(1)
            f["score_msum"] = ms
            # operands of the fused kernel (azk_nn_embed_pool): per-token constants padded to whole 16-token tiles
            Tp = (cfg.tokens + 15) // 16 * 16
            cp = torch.zeros(Tp, D, device=dev)
            cp[:cfg.tokens] = hp["cpos"]
            scp = torch.zeros(Tp, 16, device=dev)
            scp[:cfg.tokens] = sc
            scp[cfg.tokens:, :H] = -1e30                           # padding tokens: softmax weight exactly 0
            # ... stored in the kernel's accumulator order (one 16-byte load per accumulator, 1 KB contiguous per wave):
            #   cpos_frag [tile][wave][q][lane = 16 l4 + l15][r] = cpos[16 tile + 4 l4 + r][128 wave + 8 l15 + q]
            #   score_frag [tile][lane][r]                        = score constants [16 tile + 4 l4 + r][l15]
            nt = Tp // 16
            f["cpos_frag"] = cp.view(nt, 4, 4, D // 128, 16, 8).permute(0, 3, 5, 1, 4, 2).contiguous()
            f["score_frag"] = scp.view(nt, 4, 4, 16).permute(0, 1, 3, 2).contiguous()
            # |xn . m'| <= |xn| |m'| <= sqrt(D) |m'|: a static softmax reference, usable while exp(-2 bound) is a normal float
            bound = math.sqrt(D) * f["m_n"].norm(dim=1)
            ref = torch.zeros(16, device=dev)
            ref[:H] = bound
            f["score_ref"] = ref if float(bound.max()) <= 40.0 else None
            self.fused_embed_pool = D == 512 and H in (4, 8) and hp["wt"].shape[1] <= 96
            self._compact = None
            if (self.fused_embed_pool and f["score_ref"] is not None and cfg.tokens <= 256 and cfg.channels in (2, 3)
                    and cfg.patch_size in (3, 5)):
                self._compact = self._prepare_compact(f, hp, sc, ms, ref)
            Wv = Wi[2 * D:].reshape(H, dh, D).to(dev)
            f["WvT_n"] = (Wv * g).transpose(1, 2).contiguous().to(torch.bfloat16)                      # [H, D, dh]
            f["bv_n"] = (bi[2 * D:].to(dev) + (Wv @ bta).reshape(-1)).to(torch.bfloat16)
            # cls-row tail as few GEMMs: value projection and output projection composed per head
            #   x1 = x0 + bo + sum_h Wo[:, h] (Wv'_h zn_h + bv'_h) = zn_flat @ Wcomb + bias1        (x0 = cls + pos[0], constant)
            Wo, bo = m[b + "attn.out_proj.weight"].to(dev), m[b + "attn.out_proj.bias"].to(dev)
            Wvn = (Wv * g)                                                                              # [H, dh, D]
            Wcomb = torch.einsum("ohe,hed->hdo", Wo.view(D, H, dh), Wvn).reshape(H * D, D)              # [H*D, D]
            bvn = bi[2 * D:].to(dev) + (Wv @ bta).reshape(-1)
            # the library GEMMs of the tail take the weights in nn.Linear layout [out, in] through a .t() view: measured 10-20 %
            # faster than a contiguous [in, out] operand at these skinny shapes (hipBLASLt picks its "NT" kernels)
            f["Wcomb"] = Wcomb.t().contiguous().to(torch.bfloat16).t()                                  # logical [H*D, D], stored [D, H*D]
            f["bias1"] = (x0 + bo + Wo @ bvn).to(torch.bfloat16)
            # policy and value heads as one GEMM ([A+1] outputs, padded to a multiple of 8 columns)
            A = cfg.action_dim
            Ap = (A + 1 + 7) // 8 * 8
            Wh = torch.zeros(Ap, D, device=dev)
            bh = torch.zeros(Ap, device=dev)
            Wh[:A], bh[:A] = m["policy_head.weight"].to(dev), m["policy_head.bias"].to(dev)
            Wh[A], bh[A] = m["value_head.weight"].to(dev)[0], m["value_head.bias"].to(dev)[0]
            f["Wh"], f["bh"] = Wh.to(torch.bfloat16), bh.to(torch.bfloat16)
            f["W0T"] = m["blocks.0.mlp.0.weight"].to(dev, torch.bfloat16).contiguous().t()             # logical [D, 4D], stored [4D, D]
            f["W3T"] = m["blocks.0.mlp.3.weight"].to(dev, torch.bfloat16).contiguous().t()             # logical [4D, D], stored [D, 4D]
            # the same tail for the hand-written small-M GEMM (azk_nn_gemm_rows): weights in MFMA fragment order, float32 biases
            import azk
            f["WcombP"] = azk.pack_linear_weight(Wcomb.t().contiguous())                               # [D, H*D] as an nn.Linear weight
            f["W0P"] = azk.pack_linear_weight(m[b + "mlp.0.weight"].to(dev))
            f["W3P"] = azk.pack_linear_weight(m[b + "mlp.3.weight"].to(dev))
            f["WhP"] = azk.pack_linear_weight(Wh)
            f["bias1_f"] = (x0 + bo + Wo @ bvn).float().contiguous()
            f["b0_f"] = m[b + "mlp.0.bias"].to(dev, torch.float32).contiguous()
            bhp = torch.zeros(f["WhP"].numel() // D, device=dev)
            bhp[:Ap] = bh
            f["bh_f"] = bhp
            # final LayerNorm's affine folded into the merged head: Wh (gamma * xn + beta) + bh = (Wh diag(gamma)) xn + (Wh beta + bh)
            gf, bf_ = m["norm.weight"].to(dev), m["norm.bias"].to(dev)
            f["WhGP"] = azk.pack_linear_weight(Wh * gf[None, :])
            bhg = torch.zeros_like(bhp)
            bhg[:Ap] = Wh @ bf_ + bh
            f["bhG_f"] = bhg
            self.hip_tail = D in (256, 512) and (H * D) % 256 == 0
            # the chain of latency-shaped GEMMs (azk_nn_tail_gemm): the composed [H*D -> D] projection is applied in its factors -
            # per-head value projection Wv'_h (dh x D, block diagonal over heads) then the output projection Wo - 4x fewer flops
            # and weight bytes than Wcomb; both LayerNorm affines folded into the consuming weights and biases
            self.chain_tail = D == 512 and dh == 64
            if self.chain_tail:
                g2, b2 = m[b + "norm2.weight"].to(dev), m[b + "norm2.bias"].to(dev)
                W0, b0_ = m[b + "mlp.0.weight"].to(dev), m[b + "mlp.0.bias"].to(dev)
                f["WvHP"] = torch.cat([azk.pack_linear_weight(Wvn[h]).reshape(-1) for h in range(H)])       # H blocks of [dh=64][D]
                f["WoP"] = azk.pack_linear_weight(Wo)
                f["W0GP"] = azk.pack_linear_weight(W0 * g2[None, :])
                # LDS-staged wide links (azk_nn_tail_gemm_lds): LayerNorm2 is applied in the epilogue, which needs the column sums of the
                # weight the matrix pipe really multiplies with (the bf16 values of W0GP)
                f["W0GP_csum"] = azk.packed_weight_col_sums(f["W0GP"], 4 * D, D)
                f["b0G_f"] = (W0 @ b2 + b0_).float().contiguous()
                f["b3_f"] = m[b + "mlp.3.bias"].to(dev, torch.float32).contiguous()
                if self.fused_embed_pool:
                    # the patch-pooling form of embedding + pooling + value projection (k_embed_fold + one batched GEMM), tables in float64
                    r = self.fold_u(dev)
                    if r is not None:
                        self._foldu = azk.EmbedFoldTables(r, H, cfg.patch_size, D, dev)
            for k_, src in (("ln2_w", b + "norm2.weight"), ("ln2_b", b + "norm2.bias"), ("lnf_w", "norm.weight"),
                            ("lnf_b", "norm.bias"), ("b3", b + "mlp.3.bias")):
                f[k_] = m[src].to(dev, torch.float32).contiguous()
            self._gelu_epilogue = False
            try:    # exact-erf GELU fused into the GEMM when the BLAS backend offers it; checked against the unfused op
                t = torch.randn(64, D, device=dev, dtype=torch.bfloat16)
                a_ = torch._addmm_activation(self.w["blocks.0.mlp.0.bias"], t, f["W0T"], use_gelu=True)
                b_ = F.gelu(F.linear(t, self.w["blocks.0.mlp.0.weight"], self.w["blocks.0.mlp.0.bias"]))
                self._gelu_epilogue = bool(torch.allclose(a_.float(), b_.float(), rtol=2e-2, atol=2e-2))
            except Exception:
                self._gelu_epilogue = False
        self._fold = f

    def _prepare_compact(self, f, hp, sc, ms, ref):
        """Tables of azk_nn_embed_pool_compact (csrc/azk_nn.hip k_embed_pool_c): a token whose patch is empty is a constant of
        the weights - x_t = cpos[t] - so with the static softmax reference its weight wc_t[h] = exp(s_t[h] - ref[h]) and its
        normalised row xnc_t are precomputed, together with their sums over ALL tokens (z_all, l_all); the kernel evaluates
        only the tokens a stone can reach and swaps their constant contribution for the real one.  The statistics follow the
        kernel's own arithmetic (mean, E[x^2] - mean^2, rsqrt) in float32."""
        import azk
        cfg = self.cfg
        D, H, T = cfg.embed_dim, cfg.num_heads, cfg.tokens
        dev = self.device
        cpos = hp["cpos"]                                                     # [T, D] float32
        mean = sc[:, 15]                                                      # the kernel's mean = score column 15 = cpos.mean(1)
        var = ((cpos * cpos).sum(1) / D - mean * mean).clamp_min(0.0)
        rstd = torch.rsqrt(var + 1e-5)
        xnc = cpos * rstd[:, None] + (-mean * rstd)[:, None]
        s_c = rstd[:, None] * (sc - mean[:, None] * ms[None, :])               # [T, 16]
        wc = torch.zeros(T, 16, device=dev)
        wc[:, :H] = torch.exp(s_c[:, :H] - ref[None, :H])
        xnc_b = xnc.to(torch.bfloat16)
        zall = (wc.to(torch.bfloat16).double().t() @ xnc_b.double())           # [16, D], exact products summed in float64
        lall = wc.double().sum(0)
        t = dict(wt_ext=f["wt_ext"], score_msum=ms, score_ref=ref)
        t["cpos_tok"] = torch.cat([cpos, torch.zeros(1, D, device=dev)])
        null_sc = torch.zeros(1, 16, device=dev)
        null_sc[:, :H] = -1e30
        t["score_tok"] = torch.cat([sc, null_sc])
        t["wconst_tok"] = torch.cat([wc, torch.zeros(1, 16, device=dev)])
        t["xnconst_tok"] = torch.cat([xnc_b, torch.zeros(1, D, device=dev, dtype=torch.bfloat16)])
        # accumulator order [w][q][lane = 16 l4 + l15][j]: head 4 l4 + j, column 128 w + 8 l15 + q
        t["z_all"] = zall.float().view(4, 4, 4, 16, 8).permute(2, 4, 0, 3, 1).reshape(4, 8, 64, 4).contiguous()
        t["l_all"] = lall.float()
        return azk.EmbedPoolTables(t, H, cfg.patch_size, D)

    def _prepare_blocks(self):
        """Operands of the hand-written path for networks deeper than one block (csrc/azk_block.hip; main.py:186-188 builds
        Net(embed_dim=256, num_heads=8, depth=2)): per full block the four linear maps in fragment packing (output padded to 128) with
        float32 biases and the LayerNorm affines; for the LAST block, whose only consumer is the cls row (nn.py:80), the folded cls
        path as plain GEMMs - the per-board attention query m_h = scale Wk_h^T (Wq_h LN1(x0) + bq_h) and offset c_h = scale bk_h . q_h as
        ONE linear map of LN1(x0) (composed in float64), the value and output projections composed per head into [H D -> D]."""
        import azk
        cfg, m, dev = self.cfg, self.master, self.device
        D, H, A = cfg.embed_dim, cfg.num_heads, cfg.action_dim
        dh = D // H
        if D % 128 or dh not in (32, 64) or cfg.tokens > 256 or (D, H) not in ((512, 8), (256, 8), (256, 4), (512, 4)):
            return
        f32 = lambda t: t.to(dev, torch.float32).contiguous()

        def lin(wt, bias):
            n_out = wt.shape[0]
            npad = (n_out + 127) // 128 * 128
            bp = torch.zeros(npad, dtype=torch.float32, device=dev)
            bp[:n_out] = bias.to(dev, torch.float32)
            return dict(w=azk.pack_linear_weight128(wt.to(dev)), b=bp, n=npad)
        blocks = []
        for i in range(cfg.depth - 1):
            b = f"blocks.{i}."
            blocks.append(dict(ln1=(f32(m[b + "norm1.weight"]), f32(m[b + "norm1.bias"])), ln2=(f32(m[b + "norm2.weight"]), f32(m[b + "norm2.bias"])),
                               qkv=lin(m[b + "attn.in_proj_weight"], m[b + "attn.in_proj_bias"]), out=lin(m[b + "attn.out_proj.weight"], m[b + "attn.out_proj.bias"]),
                               up=lin(m[b + "mlp.0.weight"], m[b + "mlp.0.bias"]), down=lin(m[b + "mlp.3.weight"], m[b + "mlp.3.bias"])))
        b = f"blocks.{cfg.depth - 1}."
        dd = lambda k_: m[k_].to(dev, torch.float64)
        Wi, bi = dd(b + "attn.in_proj_weight"), dd(b + "attn.in_proj_bias")
        Wq, bq = Wi[:D].view(H, dh, D), bi[:D].view(H, dh)
        Wk, bk = Wi[D:2 * D].view(H, dh, D), bi[D:2 * D].view(H, dh)
        Wv, bv = Wi[2 * D:].view(H, dh, D), bi[2 * D:]
        scale = 1.0 / math.sqrt(dh)
        Wm = torch.einsum("hed,hef->hdf", Wk, Wq).reshape(H * D, D) * scale            # m_h = Wm_h LN1(x0) + bm_h
        bm = torch.einsum("hed,he->hd", Wk, bq).reshape(H * D) * scale
        Wc = torch.einsum("he,hef->hf", bk, Wq) * scale                                 # c_h = Wc_h . LN1(x0) + bc_h
        bc = (bk * bq).sum(1) * scale
        Wo, bo = dd(b + "attn.out_proj.weight"), dd(b + "attn.out_proj.bias")
        Wcomb = torch.einsum("ohe,hed->ohd", Wo.view(D, H, dh), Wv).reshape(D, H * D)      # x1 = x0 + Wcomb z_flat + (bo + Wo bv)
        Wh = torch.zeros(A + 1, D, dtype=torch.float64, device=dev)
        Wh[:A], Wh[A] = dd("policy_head.weight"), dd("value_head.weight")[0]
        bh = torch.cat([dd("policy_head.bias"), dd("value_head.bias")])
        last = dict(ln1=(f32(m[b + "norm1.weight"]), f32(m[b + "norm1.bias"])), ln2=(f32(m[b + "norm2.weight"]), f32(m[b + "norm2.bias"])),
                    lnf=(f32(m["norm.weight"]), f32(m["norm.bias"])),
                    mc=lin(torch.cat([Wm, Wc]).float(), torch.cat([bm, bc]).float()), comb=lin(Wcomb.float(), (bo + Wo @ bv).float()),
                    up=lin(m[b + "mlp.0.weight"], m[b + "mlp.0.bias"]), down=lin(m[b + "mlp.3.weight"], m[b + "mlp.3.bias"]),
                    heads=lin(Wh.float(), bh.float()))
        self._blocks = dict(full=blocks, last=last)

    def forward_blocks_hip(self, x):
        """depth > 1 on hand-written kernels only (no F.linear, torch.bmm or scaled_dot_product_attention): token embedding
        (k_embed) -> depth - 1 full-token blocks (k_ln_rows, k_gemm_tok, k_attn_tok: nn.py:52-61) -> the last block for the cls row
        (k_ln_rows, the per-board folded query as one GEMM, k_cls_attn, four GEMMs) -> final LayerNorm + merged heads (nn.py:78-83)."""
        import azk
        cfg, bl = self.cfg, self._blocks
        n, T, D, H, A = x.shape[0], cfg.tokens, cfg.embed_dim, cfg.num_heads, cfg.action_dim
        if x.dtype not in (torch.bfloat16, torch.float32):
            x = x.float()
        t = self.embed_hip(x.contiguous(), want_x=True, want_xhat=False)[0].view(n * T, D)     # tokens, bf16
        # a device-side live board count (the step graph's leaf buffer is fixed-size): boards past it are skipped by every kernel
        cb = self.live_count
        ct = (cb * T) if cb is not None else None                                              # the same count in token rows
        G = azk.nn_gemm_tok
        for blk in bl["full"]:
            h = azk.nn_layernorm_rows(t, *blk["ln1"], count=ct)
            qkv = G(h, blk["qkv"]["w"], blk["qkv"]["n"], azk.TOK_BF16, bias=blk["qkv"]["b"], count=ct)
            o = azk.nn_attention_tok(qkv, n, T, D, H, count=cb)
            t = G(o, blk["out"]["w"], blk["out"]["n"], azk.TOK_RESID, bias=blk["out"]["b"], resid=t, count=ct)            # nn.py:54-56
            h = azk.nn_layernorm_rows(t, *blk["ln2"], count=ct)
            hh = G(h, blk["up"]["w"], blk["up"]["n"], azk.TOK_GELU, bias=blk["up"]["b"], count=ct)
            t = G(hh, blk["down"]["w"], blk["down"]["n"], azk.TOK_RESID, bias=blk["down"]["b"], resid=t, count=ct)        # nn.py:59-60
        la = bl["last"]
        xhat = azk.nn_layernorm_rows(t, *la["ln1"], count=ct)                                                 # LN1 of every token: keys / values of the cls query
        x0, xh0 = t.view(n, T, D)[:, 0], xhat.view(n, T, D)[:, 0]                                             # the cls rows (row stride T D)
        mc = G(xh0, la["mc"]["w"], la["mc"]["n"], azk.TOK_F32, bias=la["mc"]["b"], count=cb)                  # [n, H D + H (+ pad)] float32
        mm, cc = mc[:, :H * D].reshape(n, H, D), mc[:, H * D:H * D + H]
        z = azk.nn_cls_attention(xhat.view(n, T, D), mm, cc, H)                                               # [n, H, D] = sum_t a_t LN1(x)_t per head
        x1 = G(z.view(n, H * D), la["comb"]["w"], la["comb"]["n"], azk.TOK_RESID, bias=la["comb"]["b"], resid=x0, count=cb)
        h = azk.nn_layernorm_rows(x1, *la["ln2"], count=cb)
        hh = G(h, la["up"]["w"], la["up"]["n"], azk.TOK_GELU, bias=la["up"]["b"], count=cb)
        x2 = G(hh, la["down"]["w"], la["down"]["n"], azk.TOK_RESID, bias=la["down"]["b"], resid=x1, count=cb)
        y = azk.nn_layernorm_rows(x2, *la["lnf"], count=cb)
        out = G(y, la["heads"]["w"], la["heads"]["n"], azk.TOK_F32, bias=la["heads"]["b"], count=cb)
        self.last_forward_kernels = "hand-written"
        return out[:, :A].contiguous(), torch.tanh(out[:, A:A + 1])

    def tail_hip(self, z):
        """tail_fast on the hand-written kernels: every launch honours the device-side live count, split-K partial sums are
        added by the row-wise kernel that follows (LayerNorm / finalize)."""
        import azk
        cfg, f = self.cfg, self._fold
        n, A, D, H = z.shape[0], cfg.action_dim, cfg.embed_dim, cfg.num_heads
        dev, cnt = z.device, self.live_count
        bf = dict(dtype=torch.bfloat16, device=dev)
        k1 = 8 if (H * D // 32) % 8 == 0 else 1
        P1 = torch.empty((k1, n, D), dtype=torch.float32, device=dev)
        azk.nn_gemm_rows(z.view(n, H * D), f["WcombP"], D, ksplit=k1, partials=P1, count=cnt)              # nn.py:54-56
        h = torch.empty((n, D), **bf)
        x1b = torch.empty((n, D), **bf)
        azk.nn_layernorm_sum(P1, f["ln2_w"], f["ln2_b"], h, bias=f["bias1_f"], add_bias=f["b3"], x_out=x1b, count=cnt)
        hh = torch.empty((n, 4 * D), **bf)
        azk.nn_gemm_rows(h, f["W0P"], 4 * D, bias=f["b0_f"], gelu_out=hh, count=cnt)                       # nn.py:59 (Linear + GELU)
        P3 = torch.empty((4, n, D), dtype=torch.float32, device=dev)
        azk.nn_gemm_rows(hh, f["W3P"], D, ksplit=4, partials=P3, count=cnt)                                # nn.py:59-60
        y = torch.empty((n, D), **bf)
        azk.nn_layernorm_sum(P3, f["lnf_w"], f["lnf_b"], y, resid=x1b, count=cnt)                          # nn.py:78
        Np = f["bh_f"].numel()
        P4 = torch.empty((4, n, Np), dtype=torch.float32, device=dev)
        azk.nn_gemm_rows(y, f["WhP"], Np, ksplit=4, partials=P4, count=cnt)                                # nn.py:82-83
        if self.out_buffers is not None:
            lb, vb = self.out_buffers
        else:
            lb = torch.empty((n, A), dtype=torch.float32, device=dev)
            vb = torch.empty(n, dtype=torch.float32, device=dev)
        azk.nn_heads_finalize_sum(P4, f["bh_f"], A, lb, vb, count=cnt)
        return lb, (vb if self.out_buffers is not None else vb[:, None])

    def tail_chain(self, z):
        """The cls-row tail as five hand-written launches (csrc/azk_nn.hip k_tail_gemm), every one honouring the device-side live
        count:  u = blockdiag_h(Wv'_h) z_h  ->  x1 = u Wo^T + bias1 (+ row statistics)  ->  hh = GELU(LN2(x1) W0'^T + b0')  ->
        x2 = x1 + b3 + hh W3^T (+ row statistics)  ->  logits, tanh(value) = heads(LNf(x2))       (nn.py:54-60, 78-83)."""
        import azk
        cfg, f = self.cfg, self._fold
        n, A, D, H = z.shape[0], cfg.action_dim, cfg.embed_dim, cfg.num_heads
        dev, cnt = z.device, self.live_count
        # one workspace per stepped engine (= per stream), sized for the largest batch seen and sliced; a workspace is never freed:
        # a captured step graph has its addresses baked in (freeing would let the caching allocator hand the memory to someone else
        # while graph replays keep writing into it), so a larger batch retires the old buffers to a keep-alive list instead
        key = id(self.leaf_source) if self.leaf_source is not None else None
        ws = self._tail_ws.get(key)
        if ws is None or ws["rows"] < n:
            rows = n if ws is None else max(n, 2 * ws["rows"])
            if ws is not None:
                self._tail_ws_retired.append(ws)
            bf = dict(dtype=torch.bfloat16, device=dev)
            ws = dict(rows=rows, u=torch.empty((rows, D), **bf), x1=torch.empty((rows, D), **bf), hh=torch.empty((rows, 4 * D), **bf),
                      x2=torch.empty((rows, D), **bf), st1=torch.empty((rows, D // 64, 2), dtype=torch.float32, device=dev),
                      st2=torch.empty((rows, D // 64, 2), dtype=torch.float32, device=dev))
            self._tail_ws[key] = ws
        ws = {k_: (v[:n] if k_ != "rows" else v) for k_, v in ws.items()}
        if z.shape[-1] == azk.EMBED_FOLD_ROW:      # k_embed_fold's rows against [D_t; U_all; M_h]: the value-projected row directly
            kf = azk.EMBED_FOLD_ROW
            self._launch(azk.nn_tail_gemm, z.view(n, H * kf), self._foldu.weight, D // H, kf, azk.TAIL_BF16, nbatch=H, a_batch_stride=kf, out=ws["u"], count=cnt)
        else:
            self._launch(azk.nn_tail_gemm, z.view(n, H * D), f["WvHP"], D // H, D, azk.TAIL_BF16, nbatch=H, a_batch_stride=D, out=ws["u"], count=cnt)
        self._launch(azk.nn_tail_gemm, ws["u"], f["WoP"], D, D, azk.TAIL_BF16, bias=f["bias1_f"], out=ws["x1"], stats_out=ws["st1"], count=cnt)
        lds = getattr(self, "use_lds_tail", True)        # the two wide links LDS-staged (csrc/azk_tail.hip) or whole-K-in-registers (k_tail_gemm)
        self._launch(azk.nn_tail_gemm, ws["x1"], f["W0GP"], 4 * D, D, azk.TAIL_GELU, bias=f["b0G_f"], out=ws["hh"], a_stats=ws["st1"], count=cnt,
                     col_sums=f["W0GP_csum"] if lds else None, lds=lds)
        self._launch(azk.nn_tail_gemm, ws["hh"], f["W3P"], D, 4 * D, azk.TAIL_RESID, bias=f["b3_f"], resid=ws["x1"], out=ws["x2"], stats_out=ws["st2"], count=cnt,
                     lds=lds)
        if self.out_buffers is not None:
            lb, vb = self.out_buffers
        else:
            lb = torch.empty((n, A), dtype=torch.float32, device=dev)
            vb = torch.empty(n, dtype=torch.float32, device=dev)
        self._launch(azk.nn_tail_gemm, ws["x2"], f["WhGP"], f["bhG_f"].numel(), D, azk.TAIL_HEADS, bias=f["bhG_f"], a_stats=ws["st2"], logits=lb, values=vb,
                         action_dim=A, count=cnt)
        return lb, (vb if self.out_buffers is not None else vb[:, None])

    def tail_fast(self, z):
        """depth-1 cls row after the pooled tokens zn [n, H, D]: composed projection, MLP, final norm, merged heads."""
        # kernel_timers = (k_embed_pool, k_tail) on the fused path, (k_embed, k_cls_pool, k_tail) on the two-kernel path
        kt = self.kernel_timers[-1] if self.kernel_timers is not None and len(self.kernel_timers) >= (2 if self.fused_embed_pool else 3) else None
        self._tail_timer = kt                        # the hand-written chains bracket EACH launch with its own event pair (the host-side gap
        try:                                         # between two eager launches is longer than these kernels: one pair around all five would time the host)
            if kt is None or z.dtype == torch.float32 or (getattr(self, "chain_tail", False) and self.use_chain_tail):
                return self._tail_fast(z)
            kt.start()
            out = self._tail_fast(z)
            kt.stop()
            return out
        finally:
            self._tail_timer = None

    def _launch(self, fn, *args, **kw):
        """One tail launch, bracketed by the tail timer's own event pair when a sampled step is being timed."""
        kt = getattr(self, "_tail_timer", None)
        if kt is not None:
            kt.start()
        fn(*args, **kw)
        if kt is not None:
            kt.stop()

    def _tail_fast(self, z):
        if z.dtype == torch.float32:
            return self.tail_exact(z)
        if getattr(self, "chain_tail", False) and self.use_chain_tail:
            return self.tail_chain(z)
        if self.hip_tail and self.use_hip_tail:
            return self.tail_hip(z)
        w, cfg, f = self.w, self.cfg, self._fold
        n, A = z.shape[0], cfg.action_dim
        import azk
        x1 = torch.addmm(f["bias1"], z.view(n, -1), f["Wcomb"])                                        # nn.py:54-56
        # norm2(x1) and, in the same pass, x1 += mlp.3 bias (the residual the last MLP GEMM accumulates onto)
        h = azk.nn_layernorm_rows(x1, f["ln2_w"], f["ln2_b"], 1e-5, add_bias=f["b3"], count=self.live_count)
        if self._gelu_epilogue:                                                                         # GELU in the GEMM epilogue (hipBLASLt)
            h = torch._addmm_activation(w["blocks.0.mlp.0.bias"], h, f["W0T"], use_gelu=True)
        else:
            h = F.gelu(F.linear(h, w["blocks.0.mlp.0.weight"], w["blocks.0.mlp.0.bias"]))
        x2 = x1.addmm_(h, f["W3T"])                                                                    # nn.py:59-60 (in place: no copy)
        if self.hip_tail and self.fuse_ln_heads:
            # final LayerNorm + merged heads + finalize in one launch (azk_nn_ln_heads), nn.py:78-83
            if self.out_buffers is not None:
                lb, vb = self.out_buffers
            else:
                lb = torch.empty((n, A), dtype=torch.float32, device=z.device)
                vb = torch.empty(n, dtype=torch.float32, device=z.device)
            azk.nn_ln_heads(x2, None, None, f["WhGP"], f["bhG_f"], A, lb, vb, count=self.live_count)     # affine folded into WhGP / bhG_f
            return lb, (vb if self.out_buffers is not None else vb[:, None])
        out = F.linear(azk.nn_layernorm_rows(x2, f["lnf_w"], f["lnf_b"], 1e-5, count=self.live_count), f["Wh"], f["bh"])   # nn.py:78-83
        if self.out_buffers is not None:
            # the step graph's own float32 buffers: conversion, slicing and tanh in one launch (azk_nn_heads_finalize)
            lb, vb = self.out_buffers
            azk.nn_heads_finalize(out, A, lb, vb, count=self.live_count)
            return lb, vb
        logits = out[:, :A]
        return logits.float(), torch.tanh(out[:, A:A + 1].float())


    def _prepare_hip_embed(self):
        """Operands of azk_nn_patch_embed: conv weight [D, kp] bf16 (k padded to a multiple of 16) and the
        per-token additive term cpos [T, D] fp32 (row 0 = cls + pos[0]; row 1+j = conv bias + pos[1+j])."""
        cfg, m = self.cfg, self.master
        D, kreal = cfg.embed_dim, cfg.channels * cfg.patch_size ** 2
        kp = (kreal + 31) // 32 * 32
        if D not in (128, 256, 512) or kp // 32 not in (1, 2, 3) or cfg.channels * cfg.rows * cfg.cols > 62 * 32:
            return
        wt = torch.zeros(D, kp)
        wt[:, :kreal] = m["embedding.patch_embed.patch_embed.weight"].reshape(D, kreal)
        pos = m["embedding.pos_embedding"][0]
        cpos = pos.clone()
        cpos[0] += m["embedding.cls_token"][0, 0]
        cpos[1:] += m["embedding.patch_embed.patch_embed.bias"]
        dev = self.device
        self._hip = dict(wt=wt.to(dev, torch.bfloat16).contiguous(), cpos=cpos.to(dev, torch.float32).contiguous(),
                         ln_w=m["blocks.0.norm1.weight"].to(dev, torch.float32).contiguous(),
                         ln_b=m["blocks.0.norm1.bias"].to(dev, torch.float32).contiguous())

    def _sched_for(self, src):
        key = id(src) if src is not None else None
        if key not in self._scheds:
            import azk
            self._scheds[key] = azk.new_sched(self.device)
        return self._scheds[key]

    def eval(self):
        return self

    def __bool__(self):
        return True

    # ---- pieces ------------------------------------------------------------------------------------------
    def embed(self, x):
        """nn.py:13-19,30-36: stride-1 'same' conv as im2col + GEMM, + bias, cls token, + positional embedding.
        (MIOpen's bf16 conv for C=2 falls back to per-image im2col / naive kernels: 16k launches per call.)"""
        w, cfg = self.w, self.cfg
        n, k = x.shape[0], cfg.patch_size
        if self._hip is not None and x.is_cuda:
            return self.embed_hip(x, want_x=True, want_xhat=False)[0]
        cols = F.unfold(x.to(self.dtype), kernel_size=k, padding=k // 2)        # [n, C*k*k, R*C]
        wmat = w["embedding.patch_embed.patch_embed.weight"].reshape(cfg.embed_dim, -1)
        t = torch.matmul(cols.transpose(1, 2), wmat.t()) + w["embedding.patch_embed.patch_embed.bias"]
        t = torch.cat([w["embedding.cls_token"].expand(n, -1, -1), t], dim=1)  # nn.py:33-34
        return t + w["embedding.pos_embedding"]                                # nn.py:35

    def embed_hip(self, x, want_x, want_xhat):
        """(tokens, LayerNorm_block0(tokens)) from the hand-written im2col + MFMA kernel (csrc/azk_nn.hip)."""
        import azk
        h, cfg = self._hip, self.cfg
        if x.dtype not in (torch.bfloat16, torch.float32):
            x = x.float()
        return azk.nn_patch_embed(x.contiguous(), h["wt"], h["cpos"], h["ln_w"], h["ln_b"], cfg.rows, cfg.cols,
                                  cfg.patch_size, cfg.embed_dim, want_x=want_x, want_xhat=want_xhat)

    def _ln(self, x, name):
        return F.layer_norm(x, (self.cfg.embed_dim,), self.w[name + ".weight"], self.w[name + ".bias"], 1e-5)

    def block_full(self, x, i, dropout_p=0.0):
        """nn.py:52-61.  dropout_p > 0 is the block in model.train() mode: the three dropouts of the reference block in
        its order and on tensors of its shapes - attention probabilities [n*H, T, T] (nn.MultiheadAttention's need_weights
        path: softmax -> dropout -> bmm), after GELU [n, T, 4D], after mlp.3 [n, T, D] - so that on the CPU a caller
        seeded like the reference draws the reference's masks (tests/test_trainer.py)."""
        w, cfg = self.w, self.cfg
        b = f"blocks.{i}."
        n, T, D = x.shape
        H = cfg.num_heads
        h = self._ln(x, b + "norm1")
        qkv = F.linear(h, w[b + "attn.in_proj_weight"], w[b + "attn.in_proj_bias"])
        q, k, v = qkv.view(n, T, 3, H, D // H).permute(2, 0, 3, 1, 4)
        if dropout_p > 0.0:
            dh = D // H
            qs = (q * math.sqrt(1.0 / float(dh))).reshape(n * H, T, dh)
            aw = torch.softmax(torch.bmm(qs, k.reshape(n * H, T, dh).transpose(1, 2)), dim=-1)
            aw = F.dropout(aw, dropout_p, training=True)
            a = torch.bmm(aw, v.reshape(n * H, T, dh)).view(n, H, T, dh)
        else:
            a = F.scaled_dot_product_attention(q, k, v)                        # softmax(q k^T / sqrt(d)) v
        a = a.transpose(1, 2).reshape(n, T, D)
        x = x + F.linear(a, w[b + "attn.out_proj.weight"], w[b + "attn.out_proj.bias"])   # nn.py:54-56
        h = self._ln(x, b + "norm2")
        h = F.gelu(F.linear(h, w[b + "mlp.0.weight"], w[b + "mlp.0.bias"]))
        if dropout_p > 0.0:
            h = F.dropout(h, dropout_p, training=True)
        h = F.linear(h, w[b + "mlp.3.weight"], w[b + "mlp.3.bias"])
        if dropout_p > 0.0:
            h = F.dropout(h, dropout_p, training=True)
        return x + h                                                           # nn.py:59-60

    def block_cls(self, x, i, h=None, x0=None):
        """Last block restricted to the cls row: exact, because only x[:,0] is read afterwards (nn.py:80).
        h = LayerNorm1(x) may be supplied (fused into the embedding kernel); then only x0 = x[:,0] is needed."""
        w, cfg = self.w, self.cfg
        b = f"blocks.{i}."
        if h is None:
            h = self._ln(x, b + "norm1")
            x0 = x[:, 0]
        n, T, D = h.shape
        H, dh = cfg.num_heads, D // cfg.num_heads
        Wi, bi = w[b + "attn.in_proj_weight"], w[b + "attn.in_proj_bias"]
        q = F.linear(h[:, 0], Wi[:D], bi[:D]).view(n, H, 1, dh)
        kv = F.linear(h, Wi[D:], bi[D:]).view(n, T, 2, H, dh)
        k, v = kv[:, :, 0].transpose(1, 2), kv[:, :, 1].transpose(1, 2)        # [n,H,T,dh]
        a = F.scaled_dot_product_attention(q, k, v).reshape(n, D)
        x0 = x0 + F.linear(a, w[b + "attn.out_proj.weight"], w[b + "attn.out_proj.bias"])
        h = self._ln(x0, b + "norm2")
        h = F.linear(F.gelu(F.linear(h, w[b + "mlp.0.weight"], w[b + "mlp.0.bias"])), w[b + "mlp.3.weight"], w[b + "mlp.3.bias"])
        return x0 + h                                                          # [n, D]

    def block_cls_folded(self, xhat, x0, i, z=None, normalised=False):
        """Same function as block_cls, evaluated without forming K or V (see _prepare_folded)."""
        import azk
        w, cfg, f = self.w, self.cfg, self._fold
        b = f"blocks.{i}."
        D = cfg.embed_dim
        n = x0.shape[0]
        H, dh = cfg.num_heads, D // cfg.num_heads
        if z is None:
            if "m" in f:
                mm, cc = f["m"], f["c"]
            else:                                    # depth > 1: the cls query depends on the board
                q = F.linear(xhat[:, 0].float(), f["Wq"], f["bq"]).view(n, H, dh)
                mm = (torch.einsum("nhe,hed->nhd", q, f["Wk"]) * self.scale).contiguous()
                cc = ((q * f["bk"]).sum(2) * self.scale).contiguous()
            z = azk.nn_cls_attention(xhat, mm, cc, H)                            # [n, H, D] bf16
        WvT, bv = (f["WvT_n"], f["bv_n"]) if normalised else (f["WvT"], f["bv"])
        a = torch.bmm(z.transpose(0, 1), WvT).transpose(0, 1).reshape(n, D) + bv
        x0 = x0 + F.linear(a, w[b + "attn.out_proj.weight"], w[b + "attn.out_proj.bias"])
        h = self._ln(x0, b + "norm2")
        h = F.linear(F.gelu(F.linear(h, w[b + "mlp.0.weight"], w[b + "mlp.0.bias"])), w[b + "mlp.3.weight"], w[b + "mlp.3.bias"])
        return x0 + h

    def heads(self, x0):
        x0 = self._ln(x0, "norm")
        logits = F.linear(x0, self.w["policy_head.weight"], self.w["policy_head.bias"])
        value = torch.tanh(F.linear(x0, self.w["value_head.weight"], self.w["value_head.bias"]))   # nn.py:82-83
        return logits.float(), value.float()

    @torch.no_grad()
    def forward(self, x, path=None):
        return self.forward_impl(x, path)

    def forward_impl(self, x, path=None, dropout_p=0.0):
        """The forward without the no_grad guard (the training step differentiates the plain 'full' path; dropout_p > 0 =
        model.train() mode of a Net built with dropout, 'full' path only)."""
        path = path or self.path
        if dropout_p > 0.0 and path != "full":
            raise ValueError("training-mode dropout runs on the 'full' path only")
        x = x.to(self.device)
        self.last_value_pre_tanh = False     # set by the one path that hands back the raw value column (fast_outputs)
        depth = self.cfg.depth
        if path == "clsfold" and self._exact is not None:
            return self.forward_exact(x)
        if path == "clsfold" and self._fold is None and self.dtype == torch.float32:
            path = "cls"    # a float32 network the hand-written kernels do not cover (configuration, or weights outside the fp16-plane scales): the torch float32 forward, same function
        if path == "clsfold":
            if self._fold is None:
                raise RuntimeError("path 'clsfold' needs the HIP kernels (CUDA, bf16, supported embed_dim/heads)")
            last = depth - 1
            if (depth == 1 and not self.chain_tail and getattr(self, "_blocks", None) is not None and getattr(self, "use_hip_blocks", True)
                    and x.is_cuda and self.leaf_source is None):
                return self.forward_blocks_hip(x)             # depth 1 outside the benchmark shape (e.g. D = 256): no library GEMM in the tail either
            if depth == 1:
                import azk
                hp, f = self._hip, self._fold
                if self.cfg.num_heads in (4, 8):
                    if x.dtype not in (torch.bfloat16, torch.float32):
                        x = x.float()
                    compact_ok = self.leaf_source is None or self.leaf_source.n_games <= azk.EMBED_POOL_COMPACT_MAX_SLOTS
                    if (self._foldu is not None and self.use_fold_u and self.chain_tail and self.use_chain_tail
                            and (self.leaf_source is None or self.leaf_source.n_games <= azk.EMBED_FOLD_MAX_SLOTS)):
                        # the token rows are never formed: per head the token weights, 1 / L and the pooled patch (k_embed_fold);
                        # the tail's first GEMM turns them into the value-projected row
                        if self.leaf_source is not None:
                            z = azk.nn_embed_fold_leaves(self.leaf_source, self._foldu, self._sched_for(self.leaf_source), timers=self.kernel_timers)
                        else:
                            z = azk.nn_embed_fold(x.contiguous(), self._foldu, self.cfg.rows, self.cfg.cols, self._sched_for(None),
                                                  count=self.live_count, timers=self.kernel_timers)
                    elif self.fused_embed_pool and self._compact is not None and self.use_compact and compact_ok:
                        # only the tokens a stone can reach are evaluated (k_embed_pool_c); boards pulled from a device queue
                        if self.leaf_source is not None:
                            z = azk.nn_embed_pool_compact_leaves(self.leaf_source, self._compact, self._sched_for(self.leaf_source),
                                                                 timers=self.kernel_timers)
                        else:
                            z = azk.nn_embed_pool_compact(x.contiguous(), self._compact, self.cfg.rows, self.cfg.cols, self._sched_for(None),
                                                          count=self.live_count, timers=self.kernel_timers)
                    elif self.fused_embed_pool and self.leaf_source is not None:
                        # boards straight from the engine's pending leaves (no compaction launch, no evaluator batch)
                        z = azk.nn_embed_pool_leaves(self.leaf_source, f["wt_ext"], f["cpos_frag"], f["score_frag"], f["score_msum"],
                                                     f["score_ref"], self.cfg.patch_size, self.cfg.embed_dim, self.cfg.num_heads,
                                                     timers=self.kernel_timers)
                    elif self.fused_embed_pool:
                        # one launch: the normalised tokens never reach HBM (azk_nn_embed_pool)
                        z = azk.nn_embed_pool(x.contiguous(), f["wt_ext"], f["cpos_frag"], f["score_frag"], f["score_msum"],
                                              f["score_ref"], self.cfg.rows, self.cfg.cols, self.cfg.patch_size, self.cfg.embed_dim,
                                              self.cfg.num_heads, count=self.live_count, timers=self.kernel_timers)
                    else:
                        z = azk.nn_embed_scores_pool(x.contiguous(), f["wt_ext"], hp["cpos"], f["score_cpos"], f["score_msum"], f["c_n"],
                                                     self.cfg.rows, self.cfg.cols, self.cfg.patch_size, self.cfg.embed_dim,
                                                     self.cfg.num_heads, count=self.live_count, timers=self.kernel_timers)
                    return self.tail_fast(z)
                x0 = hp["cpos"][0].to(self.dtype).expand(x.shape[0], -1)
                _, xhat = self.embed_hip(x, want_x=False, want_xhat=True)
            else:
                if getattr(self, "_blocks", None) is not None and getattr(self, "use_hip_blocks", True) and x.is_cuda:
                    return self.forward_blocks_hip(x)
                t = self.embed(x)
                for i in range(last):
                    t = self.block_full(t, i)
                xhat = self._ln(t, f"blocks.{last}.norm1").contiguous()
                x0 = t[:, 0]
            return self.heads(self.block_cls_folded(xhat, x0, last))
        if path == "cls" and depth == 1 and self._hip is not None:
            # tokens are never materialised: the embedding kernel emits LayerNorm1(tokens) and the cls residual
            # row is the constant cls + pos[0]
            _, xhat = self.embed_hip(x, want_x=False, want_xhat=True)
            x0 = self._hip["cpos"][0].to(self.dtype).expand(x.shape[0], -1)
            return self.heads(self.block_cls(None, 0, h=xhat, x0=x0))
        x = self.embed(x)
        if path == "full":
            for i in range(depth):
                x = self.block_full(x, i, dropout_p)
            x0 = x[:, 0]
        elif path == "cls":
            for i in range(depth - 1):
                x = self.block_full(x, i)
            x0 = self.block_cls(x, depth - 1)
        else:
            raise ValueError(path)
        return self.heads(x0)

    __call__ = forward
