"""azk - ctypes binding of libazk.so (include/azk.h), the MI355X-native self-play engine.

Host orchestration stays Python (as in the reference); everything on the hot path is a HIP kernel
behind the C ABI.  There is NO CPU fallback: if libazk.so is missing or no GPU is visible the
functions here raise.  PyTorch is used for device memory and streams only.
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(os.path.dirname(_HERE), "csrc")
LIB_PATH = os.path.join(_HERE, "libazk.so")

ABI_VERSION = 4           # include/azk.h AZK_ABI_VERSION the structure layouts below were written for
GAME_ID = {"tictactoe": 0, "connect4": 1, "gomoku": 2}
LEAF_F32, LEAF_BF16 = 0, 1
EMBED_POOL_COMPACT_MAX_SLOTS = 65279       # AZK_EMBED_POOL_COMPACT_MAX_SLOTS (include/azk.h)

# every symbol include/azk.h declares (checked by tests/test_abi.py against the header text)
SYMBOLS = [
    "azk_abi_version", "azk_last_error", "azk_create", "azk_destroy", "azk_geometry", "azk_reset_games",
    "azk_set_positions", "azk_begin_search", "azk_step_select", "azk_step_expand_backup", "azk_step",
    "azk_root_stats", "azk_root_children", "azk_export_tree", "azk_advance", "azk_get_positions",
    "azk_get_counters", "azk_reset_counters", "azk_check_device_error", "azk_gen_noise",
    "azk_rules_legal_moves", "azk_rules_legal_mask", "azk_rules_apply_move", "azk_rules_undo_move",
    "azk_rules_check_winner", "azk_rules_canonical", "azk_softmax_rows",
    "azk_step_tree", "azk_step_gather", "azk_recycle_finished", "azk_nn_patch_embed", "azk_nn_cls_attention", "azk_nn_patch_embed_scores", "azk_nn_cls_pool", "azk_debug_stamps", "azk_debug_stamps_raw", "azk_emit_finished", "azk_clear_cache", "azk_nn_heads_finalize", "azk_nn_layernorm_rows",
    "azk_vanilla_set_rng", "azk_vanilla_get_rng", "azk_vanilla_search", "azk_nn_embed_pool",
    "azk_nn_gemm_rows", "azk_nn_layernorm_sum", "azk_nn_heads_finalize_sum", "azk_nn_ln_heads",
    "azk_leaf_source_of", "azk_nn_embed_pool_leaves", "azk_nn_embed_pool_compact", "azk_nn_embed_pool_compact_leaves",
    "azk_nn_embed_fold", "azk_nn_embed_fold_leaves", "azk_nnx_embed_fold", "azk_nnx_embed_fold_leaves",
    "azk_nn_tail_gemm", "azk_nn_tail_gemm_lds", "azk_nn_tail_lds_footprint", "azk_nn_embed_fold_grid", "azk_nn_gemm_tok", "azk_nn_attention_tok", "azk_begin_search_budget", "azk_search_unfinished",
    "azk_nnx_embed_pool", "azk_nnx_embed_pool_leaves", "azk_nnx_gemm", "azk_nnx_gemm_h", "azk_nnx_gemm_h_lds",
    "azk_async_begin", "azk_async_step", "azk_async_drain", "azk_async_set_budget",
]


class AzkError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [("game", C.c_int32), ("rows", C.c_int32), ("cols", C.c_int32), ("n_games", C.c_int32),
                ("max_sims", C.c_int32), ("leaf_dtype", C.c_int32), ("device", C.c_int32),
                ("arena_nodes", C.c_int32), ("cache_entries", C.c_int32), ("cache_shared", C.c_int32), ("leaves_per_step", C.c_int32),
                ("reserved", C.c_int32 * 5)]


class LeafSource(C.Structure):
    """azk_leaf_source (include/azk.h): where azk_nn_embed_pool_leaves finds the pending leaves of an engine."""
    _fields_ = [("leaf_flag", C.c_void_p), ("leaf_cells", C.c_void_p), ("to_move", C.c_void_p), ("leaf_depth", C.c_void_p),
                ("leaf_slot", C.c_void_p), ("n_leaf", C.c_void_p), ("n_games", C.c_int32), ("rows", C.c_int32), ("cols", C.c_int32),
                ("rc", C.c_int32), ("rc_pad", C.c_int32), ("planes", C.c_int32), ("flag_bytes", C.c_int32), ("cache_stamp", C.c_void_p)]


class EmbedPoolConsts(C.Structure):
    """azk_embed_pool_consts (include/azk.h): the per-token tables of the compacting embedding + pooling kernel."""
    _fields_ = [("wt_frag", C.c_void_p), ("cpos_tok", C.c_void_p), ("score_tok", C.c_void_p), ("wconst_tok", C.c_void_p),
                ("xnconst_tok", C.c_void_p), ("z_all", C.c_void_p), ("l_all", C.c_void_p), ("score_msum", C.c_void_p),
                ("score_ref", C.c_void_p), ("num_heads", C.c_int32), ("ksize", C.c_int32), ("kp", C.c_int32),
                ("embed_dim", C.c_int32), ("ln_eps", C.c_float), ("work_stats", C.c_void_p)]


class EmbedFoldConsts(C.Structure):
    """azk_embed_fold_consts (include/azk.h): tables of the patch-pooling embedding kernel."""
    _fields_ = [("g_frag", C.c_void_p), ("e_frag", C.c_void_p), ("u2_tok", C.c_void_p), ("score_tok", C.c_void_p),
                ("wconst_tok", C.c_void_p), ("l_all", C.c_void_p), ("score_ref", C.c_void_p), ("inv_scales", C.c_void_p),
                ("num_heads", C.c_int32), ("ksize", C.c_int32), ("embed_dim", C.c_int32), ("ln_eps", C.c_float),
                ("work_stats", C.c_void_p)]


class TailGemm(C.Structure):
    """azk_tail_gemm (include/azk.h): one link of the cls-row tail."""
    _fields_ = [("a_bf16", C.c_void_p), ("lda", C.c_int32), ("a_batch_stride", C.c_int32), ("w_packed", C.c_void_p),
                ("m", C.c_int32), ("n_out", C.c_int32), ("k", C.c_int32), ("nbatch", C.c_int32), ("n_valid", C.c_void_p),
                ("bias", C.c_void_p), ("layernorm_a", C.c_int32), ("epilogue", C.c_int32), ("ln_eps", C.c_float),
                ("a_stats", C.c_void_p), ("a_stats_groups", C.c_int32), ("stats_out", C.c_void_p),
                ("out_bf16", C.c_void_p), ("ldo", C.c_int32), ("resid_bf16", C.c_void_p), ("ldr", C.c_int32),
                ("logits_out", C.c_void_p), ("values_out", C.c_void_p), ("action_dim", C.c_int32), ("a_col_sums", C.c_void_p)]


class GemmTok(C.Structure):
    """azk_gemm_tok (include/azk.h): the LDS-staged GEMM of the full-token transformer block."""
    _fields_ = [("a_bf16", C.c_void_p), ("lda", C.c_int32), ("w_packed", C.c_void_p), ("m", C.c_int32), ("n_out", C.c_int32), ("k", C.c_int32),
                ("n_valid", C.c_void_p), ("bias", C.c_void_p), ("epilogue", C.c_int32), ("out", C.c_void_p), ("ldo", C.c_int32),
                ("resid_bf16", C.c_void_p), ("ldr", C.c_int32)]


class EmbedPoolXConsts(C.Structure):
    """azk_embed_pool_x_consts (include/azk.h): tables of the fp32-accurate embedding + pooling kernel."""
    _fields_ = [("wt_frag", C.c_void_p), ("cpos_tok", C.c_void_p), ("score_tok", C.c_void_p), ("wconst_tok", C.c_void_p),
                ("xnconst_tok", C.c_void_p), ("z_all", C.c_void_p), ("l_all", C.c_void_p), ("score_msum", C.c_void_p),
                ("score_ref", C.c_void_p), ("num_heads", C.c_int32), ("ksize", C.c_int32), ("kp", C.c_int32),
                ("embed_dim", C.c_int32), ("ln_eps", C.c_float), ("wt_scale", C.c_float), ("work_stats", C.c_void_p),
                ("wconst_h16_tok", C.c_void_p), ("pool_scale", C.c_float)]


class GemmX(C.Structure):
    """azk_gemm_x (include/azk.h): one link of the cls-row tail in float32."""
    _fields_ = [("a_f32", C.c_void_p), ("lda", C.c_int32), ("a_batch_stride", C.c_int32), ("w_packed", C.c_void_p),
                ("m", C.c_int32), ("n_out", C.c_int32), ("k", C.c_int32), ("nbatch", C.c_int32), ("n_valid", C.c_void_p),
                ("bias", C.c_void_p), ("layernorm_a", C.c_int32), ("epilogue", C.c_int32), ("ln_eps", C.c_float),
                ("a_stats", C.c_void_p), ("stats_out", C.c_void_p), ("out_f32", C.c_void_p), ("ldo", C.c_int32),
                ("resid_f32", C.c_void_p), ("ldr", C.c_int32), ("logits_out", C.c_void_p), ("values_out", C.c_void_p),
                ("action_dim", C.c_int32)]


class GemmH(C.Structure):
    """azk_gemm_h (include/azk.h): one link of the cls-row tail on fp16 (hi, lo) operand planes."""
    _fields_ = [("a_hi", C.c_void_p), ("a_lo", C.c_void_p), ("a_f32", C.c_void_p), ("lda", C.c_int32), ("a_batch_stride", C.c_int32),
                ("w_packed", C.c_void_p), ("m", C.c_int32), ("n_out", C.c_int32), ("k", C.c_int32), ("nbatch", C.c_int32),
                ("n_valid", C.c_void_p), ("bias", C.c_void_p), ("col_sums", C.c_void_p), ("layernorm_a", C.c_int32), ("epilogue", C.c_int32),
                ("ln_eps", C.c_float), ("a_scale", C.c_float), ("w_scale", C.c_float), ("a_stats", C.c_void_p), ("stats_out", C.c_void_p),
                ("out_hi", C.c_void_p), ("out_lo", C.c_void_p), ("out_f32", C.c_void_p), ("ldo", C.c_int32), ("resid_f32", C.c_void_p),
                ("ldr", C.c_int32), ("logits_out", C.c_void_p), ("values_out", C.c_void_p), ("action_dim", C.c_int32), ("overflow_flag", C.c_void_p)]


class AsyncConfig(C.Structure):
    """azk_async_config (include/azk.h)."""
    _fields_ = [("n_sims", C.c_int32), ("max_sims_per_launch", C.c_int32), ("sample_until_move", C.c_int32), ("dirichlet", C.c_int32),
                ("recycle", C.c_int32), ("young_launch_us", C.c_int32), ("seed", C.c_uint64), ("first_global_game", C.c_int64), ("alpha", C.c_double),
                ("stats_dev", C.c_void_p), ("record_capacity", C.c_int64), ("rec_meta_dev", C.c_void_p), ("rec_q_dev", C.c_void_p),
                ("rec_pi_dev", C.c_void_p)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("sims", "edges_scanned", "trace_nodes", "edges_created",
                                          "leaves_evaluated", "terminal_sims", "moves_played", "cache_hits")] + [("reserved", C.c_int64 * 8)]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_[:8]}


def build(force=False, verbose=False):
    """Compile libazk.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(os.path.dirname(os.path.dirname(_HERE)), "include", "azk.h"))
    if not force and os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in srcs):
        return LIB_PATH
    cmd = ["make", "-C", _CSRC] + (["-B"] if force else [])
    r = subprocess.run(cmd, capture_output=not verbose, text=True)
    if r.returncode != 0:
        raise AzkError("building libazk.so failed:\n" + (r.stdout or "") + (r.stderr or ""))
    return LIB_PATH


_LIB = None


def lib():
    """Load libazk.so (no GPU needed to load; compute entry points need one)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise AzkError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(the HIP extension is required; there is no CPU fallback)")
    # PyTorch ships its own HIP runtime (libamdhip64.so.7 + libhsa-runtime64); it must be the process's only
    # one, so torch is imported before libazk.so is mapped and the soname resolves to the copy already loaded.
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, u64, f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_double
    L.azk_abi_version.restype = i32
    if L.azk_abi_version() != ABI_VERSION:
        raise AzkError(f"{LIB_PATH} speaks ABI version {L.azk_abi_version()}, this binding was written for {ABI_VERSION} "
                       "(include/azk.h AZK_ABI_VERSION): rebuild with __graft_entry__.build()")
    L.azk_last_error.restype = C.c_char_p
    L.azk_last_error.argtypes = [vp]
    L.azk_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.azk_destroy.argtypes = [vp]
    L.azk_destroy.restype = None
    L.azk_geometry.argtypes = [vp] + [C.POINTER(i32)] * 5
    L.azk_reset_games.argtypes = [vp, i32, i32, vp]
    L.azk_set_positions.argtypes = [vp, i32, i32, vp, vp, vp, vp]
    L.azk_begin_search.argtypes = [vp, vp, vp]
    L.azk_step_select.argtypes = [vp, vp, vp, vp]
    L.azk_step_expand_backup.argtypes = [vp, vp, vp, vp]
    L.azk_step.argtypes = [vp, vp, vp, vp, vp, vp]
    L.azk_root_stats.argtypes = [vp, vp, vp, vp, vp]
    L.azk_root_children.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp]
    L.azk_export_tree.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp, vp]
    L.azk_advance.argtypes = [vp, vp, i32, vp, vp, vp, vp]
    L.azk_get_positions.argtypes = [vp, vp, vp, vp, vp]
    L.azk_get_counters.argtypes = [vp, C.POINTER(Counters), vp]
    L.azk_reset_counters.argtypes = [vp, vp]
    L.azk_clear_cache.argtypes = [vp, vp]
    L.azk_emit_finished.argtypes = [vp, vp, vp, vp, i64, vp, vp, vp]
    L.azk_debug_stamps.argtypes = [vp, vp]
    L.azk_check_device_error.argtypes = [vp, vp]
    L.azk_gen_noise.argtypes = [vp, u64, i64, i32, f64, vp, vp, vp]
    for name in ("azk_rules_legal_moves",):
        getattr(L, name).argtypes = [i32, i32, i32, vp, i32, vp, vp, vp]
    L.azk_rules_legal_mask.argtypes = [i32, i32, i32, vp, i32, vp, vp]
    L.azk_rules_apply_move.argtypes = [i32, i32, i32, vp, i32, vp, vp, vp, vp]
    L.azk_rules_undo_move.argtypes = [i32, i32, i32, vp, i32, vp, vp, vp]
    L.azk_rules_check_winner.argtypes = [i32, i32, i32, vp, i32, vp, vp, vp, vp]
    L.azk_rules_canonical.argtypes = [i32, i32, i32, vp, i32, vp, vp, vp]
    L.azk_softmax_rows.argtypes = [vp, i32, i32, vp, vp]
    L.azk_step_tree.argtypes = [vp, vp, vp, vp]
    L.azk_vanilla_set_rng.argtypes = [vp, i32, i32, vp, vp]
    L.azk_vanilla_get_rng.argtypes = [vp, i32, i32, vp, vp]
    L.azk_vanilla_search.argtypes = [vp, i32, vp]
    L.azk_step_gather.argtypes = [vp, vp, vp, vp]
    L.azk_recycle_finished.argtypes = [vp, vp, vp]
    L.azk_nn_patch_embed_scores.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, C.c_float, vp, vp]
    L.azk_nn_embed_pool.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, C.c_float, vp, vp]
    L.azk_leaf_source_of.argtypes = [vp, vp, C.POINTER(LeafSource)]
    L.azk_nn_embed_pool_leaves.argtypes = [C.POINTER(LeafSource), vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, C.c_float, vp]
    L.azk_nn_embed_pool_compact.argtypes = [vp, i32, C.POINTER(EmbedPoolConsts), vp, i32, i32, i32, i32, vp, vp, vp]
    L.azk_nn_embed_pool_compact_leaves.argtypes = [C.POINTER(LeafSource), C.POINTER(EmbedPoolConsts), vp, vp, vp]
    L.azk_nn_embed_fold.argtypes = [vp, i32, C.POINTER(EmbedFoldConsts), vp, i32, i32, i32, i32, vp, vp, vp]
    L.azk_nn_embed_fold_leaves.argtypes = [C.POINTER(LeafSource), C.POINTER(EmbedFoldConsts), vp, vp, vp]
    L.azk_nnx_embed_fold.argtypes = [vp, i32, C.POINTER(EmbedFoldConsts), vp, i32, i32, i32, i32, vp, vp, vp]
    L.azk_nnx_embed_fold_leaves.argtypes = [C.POINTER(LeafSource), C.POINTER(EmbedFoldConsts), vp, vp, vp]
    L.azk_begin_search_budget.argtypes = [vp, vp, i32, i32, vp]
    L.azk_search_unfinished.argtypes = [vp, vp, vp]
    L.azk_nn_tail_gemm.argtypes = [C.POINTER(TailGemm), vp]
    L.azk_nn_tail_gemm_lds.argtypes = [C.POINTER(TailGemm), vp]
    L.azk_nn_gemm_tok.argtypes = [C.POINTER(GemmTok), vp]
    L.azk_nn_attention_tok.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp]
    L.azk_nnx_embed_pool.argtypes = [vp, i32, C.POINTER(EmbedPoolXConsts), vp, i32, i32, i32, i32, vp, vp, vp]
    L.azk_nnx_embed_pool_leaves.argtypes = [C.POINTER(LeafSource), C.POINTER(EmbedPoolXConsts), vp, vp, vp]
    L.azk_nnx_gemm.argtypes = [C.POINTER(GemmX), vp]
    L.azk_nnx_gemm_h.argtypes = [C.POINTER(GemmH), vp]
    L.azk_nnx_gemm_h_lds.argtypes = [C.POINTER(GemmH), vp]
    L.azk_async_begin.argtypes = [vp, C.POINTER(AsyncConfig), vp]
    L.azk_async_step.argtypes = [vp, vp, vp, i32, vp]
    L.azk_async_set_budget.argtypes = [vp, i32, i32, vp]
    L.azk_async_drain.argtypes = [vp, vp, vp, vp, i64, vp, vp]
    L.azk_nn_ln_heads.argtypes = [vp, vp, vp, C.c_float, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp]
    L.azk_nn_gemm_rows.argtypes = [vp, i32, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp]
    L.azk_nn_layernorm_sum.argtypes = [vp, i32, i32, vp, vp, vp, vp, C.c_float, vp, vp, vp, i32, i32, vp, vp]
    L.azk_nn_heads_finalize_sum.argtypes = [vp, i32, i32, i32, vp, i32, i32, vp, vp, vp, vp]
    L.azk_nn_layernorm_rows.argtypes = [vp, vp, vp, C.c_float, vp, vp, i32, i32, vp, vp]
    L.azk_nn_heads_finalize.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp]
    L.azk_nn_cls_pool.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp, vp]
    L.azk_nn_cls_attention.argtypes = [vp, vp, vp, i32, vp, i32, i32, i32, i32, vp]
    L.azk_nn_patch_embed.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, C.c_float, vp]
    for name in SYMBOLS:
        f = getattr(L, name)
        if name not in ("azk_last_error", "azk_destroy"):
            f.restype = i32
    _LIB = L
    return L


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise AzkError("no GPU visible: the azk engine runs only on an MI355X (no CPU fallback)")
    return torch


def _stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _np(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Engine:
    """G concurrent games + their search trees resident on one GPU (one engine per process / GPU)."""

    def __init__(self, game, n_games, max_sims, size=None, device=0, leaf_dtype="float32", arena_nodes=0, cache_entries=0,
                 cache_shared=False, leaves_per_step=1):
        torch = _torch()
        self.torch = torch
        self.L = lib()
        self.game = game
        cfg = Config()
        cfg.game = GAME_ID[game]
        cfg.rows = cfg.cols = int(size or 0)
        cfg.n_games, cfg.max_sims, cfg.device, cfg.arena_nodes = int(n_games), int(max_sims), int(device), int(arena_nodes)
        cfg.cache_entries = int(cache_entries)
        cfg.cache_shared = 1 if (cache_shared and cache_entries) else 0      # one table for all games (the reference's process-global MCTS.cache)
        self.cache_shared = bool(cfg.cache_shared)
        # OPT-IN virtual-loss expansion: K leaves in flight per game (changes search results; 1 = the reference's sequential search)
        self.K = max(1, int(leaves_per_step))
        cfg.leaves_per_step = self.K
        cfg.leaf_dtype = LEAF_BF16 if leaf_dtype in ("bfloat16", "bf16", torch.bfloat16) else LEAF_F32
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        h = C.c_void_p()
        rc = self.L.azk_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise AzkError(f"azk_create failed ({rc}): {self.L.azk_last_error(None).decode()}")
        self.h = h
        vals = [C.c_int32() for _ in range(5)]
        self._chk(self.L.azk_geometry(self.h, *[C.byref(v) for v in vals]))
        self.planes, self.rows, self.cols, self.action_dim, self.state_dim = [v.value for v in vals]
        self.G, self.max_sims = int(n_games), int(max_sims)
        tdt = torch.bfloat16 if cfg.leaf_dtype == LEAF_BF16 else torch.float32
        dev = self.device
        self.slots = self.G * self.K                # pending-leaf slots = capacity of one evaluator batch
        self.leaf_boards = torch.zeros((self.slots, self.planes, self.rows, self.cols), dtype=tdt, device=dev)
        self.n_leaf = torch.zeros(1, dtype=torch.int32, device=dev)
        self.pi = torch.zeros((self.G, self.action_dim), dtype=torch.float64, device=dev)
        self.q = torch.zeros(self.G, dtype=torch.float64, device=dev)
        self.root_visit = torch.zeros(self.G, dtype=torch.int32, device=dev)
        self.chosen = torch.zeros(self.G, dtype=torch.int32, device=dev)
        self.winner = torch.zeros(self.G, dtype=torch.int32, device=dev)
        self.done = torch.zeros(self.G, dtype=torch.int32, device=dev)
        self._noise = None
        self.cache_entries = int(cache_entries)
        # with the eval cache a step can have pending (cached) leaves to expand although no leaf went to the evaluator
        need = cache_entries or self.K > 1
        self._no_logits = torch.zeros((1, self.action_dim), dtype=torch.float32, device=dev) if need else None
        self._no_values = torch.zeros(1, dtype=torch.float32, device=dev) if need else None

    def _chk(self, rc):
        if rc < 0:
            raise AzkError(f"libazk error {rc}: {self.L.azk_last_error(self.h).decode()}")
        return rc

    def close(self):
        if getattr(self, "h", None):
            self.L.azk_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- state ----------------------------------------------------------------------------------
    def reset_games(self, first=0, count=None):
        self._chk(self.L.azk_reset_games(self.h, first, self.G - first if count is None else count, _stream()))

    def set_positions(self, cells, to_move, move_count, first=0):
        cells = np.ascontiguousarray(cells, np.int8).reshape(-1, self.rows * self.cols)
        tm = np.ascontiguousarray(to_move, np.int32)
        mc = np.ascontiguousarray(move_count, np.int32)
        self._chk(self.L.azk_set_positions(self.h, first, len(cells), _np(cells), _np(tm), _np(mc), _stream()))

    def get_positions(self):
        cells = np.empty((self.G, self.rows * self.cols), np.int8)
        tm = np.empty(self.G, np.int32)
        mc = np.empty(self.G, np.int32)
        self._chk(self.L.azk_get_positions(self.h, _np(cells), _np(tm), _np(mc), _stream()))
        return cells, tm, mc

    # ---- search ---------------------------------------------------------------------------------
    def begin_search(self, noise=None):
        """noise: float64 CUDA tensor [G, A] (np.random.dirichlet draws) or None (dirichlet=False)."""
        if noise is not None:
            assert noise.dtype == self.torch.float64 and noise.is_cuda and noise.is_contiguous()
            assert tuple(noise.shape) == (self.G, self.action_dim)
        self._noise = noise     # keep alive for the whole search
        self._chk(self.L.azk_begin_search(self.h, _p(noise), _stream()))

    def begin_search_budget(self, noise, n_sims, per_launch=8):
        """begin_search + a simulation budget: afterwards every step lets a game run on inside the launch while its simulations
        need no evaluator (terminal leaves, eval-cache hits); step until unfinished() == 0, then step_expand_backup once."""
        if noise is not None:
            assert noise.dtype == self.torch.float64 and noise.is_cuda and noise.is_contiguous()
            assert tuple(noise.shape) == (self.G, self.action_dim)
        assert n_sims <= self.max_sims
        self._noise = noise
        self._chk(self.L.azk_begin_search_budget(self.h, _p(noise), int(n_sims), int(per_launch), _stream()))

    def unfinished(self):
        """Games that still owe simulations of their budget or whose last leaf awaits the evaluator (host sync)."""
        out = C.c_int32(0)
        self._chk(self.L.azk_search_unfinished(self.h, C.byref(out), _stream()))
        return int(out.value)

    def step(self, logits=None, values=None):
        """expand+backup the previous leaves (if logits given) and select the next; no host sync."""
        self._chk(self.L.azk_step(self.h, _p(logits), _p(values), _p(self.leaf_boards), _p(self.n_leaf), _stream()))

    def step_tree(self, logits=None, values=None):
        self._chk(self.L.azk_step_tree(self.h, _p(logits), _p(values), _stream()))

    def leaf_source(self):
        """The engine's pending-leaf arrays for azk_nn_embed_pool_leaves (the leaf count lands in self.n_leaf)."""
        if getattr(self, "_leaf_source", None) is None:
            src = LeafSource()
            self._chk(self.L.azk_leaf_source_of(self.h, _p(self.n_leaf), C.byref(src)))
            self._leaf_source = src
        return self._leaf_source

    def step_gather(self):
        self._chk(self.L.azk_step_gather(self.h, _p(self.leaf_boards), _p(self.n_leaf), _stream()))

    def recycle_finished(self, stats):
        """stats: int64 CUDA tensor [8] accumulating (games, plies, wins0, wins1, draws)."""
        assert stats.dtype == self.torch.int64 and stats.is_cuda and stats.numel() >= 8
        self._chk(self.L.azk_recycle_finished(self.h, _p(stats), _stream()))

    def emit_finished(self, replay):
        """Append the (state, pi, z) tuples of every just-finished game to a DeviceReplay; returns int64 [G] first indices (-1: not emitted)."""
        base = self.torch.empty(self.G, dtype=self.torch.int64, device=self.device)
        self._chk(self.L.azk_emit_finished(self.h, _p(replay.states), _p(replay.pis), _p(replay.zs), replay.capacity,
                                           _p(replay.cursor), _p(base), _stream()))
        return base

    def step_select(self):
        self._chk(self.L.azk_step_select(self.h, _p(self.leaf_boards), _p(self.n_leaf), _stream()))

    def step_expand_backup(self, logits, values):
        self._chk(self.L.azk_step_expand_backup(self.h, _p(logits), _p(values), _stream()))

    def search(self, evaluator, n_sims, noise=None):
        """MCTS.mcts for all G games: n_sims simulations each, one in flight per game.
        evaluator(boards[n,F,R,C]) -> (logits[n,A] float32, values[n] or [n,1] float32), on the GPU."""
        assert n_sims <= self.max_sims
        torch = self.torch
        self.begin_search(noise)
        logits = values = None
        for _ in range(n_sims):
            self.step(logits, values)
            n = int(self.n_leaf.item())
            if n > 0:
                logits, values = evaluator(self.leaf_boards[:n])
                logits = logits.to(torch.float32).contiguous()
                values = values.to(torch.float32).reshape(-1).contiguous()
                assert logits.shape == (n, self.action_dim) and values.shape[0] == n
            elif self.cache_entries:
                logits, values = self._no_logits, self._no_values      # cache hits still need their expand + backup
            else:
                logits = values = None
        if logits is not None:
            self.step_expand_backup(logits, values)

    def search_budget(self, evaluator, n_sims, noise=None, per_launch=8):
        """The same search as `search` - bit-identical trees - with budget stepping: a game runs on inside a launch while its
        simulations need no evaluator, so n_sims simulations take about (share of simulations that miss the cache) * n_sims
        launches, each with a fuller evaluator batch.  Returns the number of launches."""
        torch = self.torch
        self.begin_search_budget(noise, n_sims, per_launch)
        logits = values = None
        launches = 0
        while True:
            self.step(logits, values)
            launches += 1
            n = int(self.n_leaf.item())
            if n > 0:
                logits, values = evaluator(self.leaf_boards[:n])
                logits = logits.to(torch.float32).contiguous()
                values = values.to(torch.float32).reshape(-1).contiguous()
                assert logits.shape == (n, self.action_dim) and values.shape[0] == n
            else:
                logits, values = (self._no_logits, self._no_values) if self._no_logits is not None else (None, None)
                if self.unfinished() == 0:
                    break
            assert launches <= 2 * n_sims + 8, "budget stepping does not terminate"
        if self.cache_entries and self.K == 1:
            self.step_expand_backup(self._no_logits, self._no_values)       # leaves served by the cache in the last launch
        return launches

    # ---- asynchronous self-play (azk_async_*) -----------------------------------------------------
    def async_begin(self, n_sims, per_launch, sample_until, seed, first_global_game, alpha=0.03, dirichlet=True, recycle=True, record_capacity=0,
                    young_launch_us=0):
        """Every game starts its first search; from now on azk_async_step moves each game as soon as its own search is done.
        Returns (stats int64 [16] CUDA, records dict or None) - caller-visible tensors the engine writes (include/azk.h)."""
        torch = self.torch
        self.async_stats = torch.zeros(16, dtype=torch.int64, device=self.device)
        rec = None
        if record_capacity:
            rec = dict(meta=torch.zeros((record_capacity, 4), dtype=torch.int32, device=self.device),
                       q=torch.zeros(record_capacity, dtype=torch.float64, device=self.device),
                       pi=torch.zeros((record_capacity, self.action_dim), dtype=torch.float64, device=self.device))
        self.async_records = rec
        c = AsyncConfig()
        c.n_sims, c.max_sims_per_launch, c.sample_until_move = int(n_sims), int(per_launch), int(min(sample_until, 1 << 30))
        c.dirichlet, c.recycle, c.seed, c.first_global_game, c.alpha = int(bool(dirichlet)), int(bool(recycle)), int(seed), int(first_global_game), float(alpha)
        c.stats_dev, c.record_capacity = self.async_stats.data_ptr(), int(record_capacity)
        c.young_launch_us = int(young_launch_us)      # > 0: another simulation inside a launch only while the launch is younger than this
        if rec is not None:
            c.rec_meta_dev, c.rec_q_dev, c.rec_pi_dev = rec["meta"].data_ptr(), rec["q"].data_ptr(), rec["pi"].data_ptr()
        self._chk(self.L.azk_async_begin(self.h, C.byref(c), _stream()))
        return self.async_stats, rec

    def async_step(self, logits, values, phases=3):
        self._chk(self.L.azk_async_step(self.h, _p(logits), _p(values), int(phases), _stream()))

    def async_set_budget(self, n_sims, per_launch):
        self._chk(self.L.azk_async_set_budget(self.h, int(n_sims), int(per_launch), _stream()))

    def async_drain(self, replay=None):
        if replay is None:
            self._chk(self.L.azk_async_drain(self.h, None, None, None, 0, None, _stream()))
        else:
            self._chk(self.L.azk_async_drain(self.h, _p(replay.states), _p(replay.pis), _p(replay.zs), replay.capacity, _p(replay.cursor), _stream()))

    # ---- vanilla mode (model=None) ----------------------------------------------------------------
    def vanilla_set_rng(self, states, first=0):
        """states: uint32 [count, 625] = MT19937 key + position per game (np.random.get_state()[1:3])."""
        st = np.ascontiguousarray(states, np.uint32).reshape(-1, 625)
        self._chk(self.L.azk_vanilla_set_rng(self.h, first, len(st), _np(st), _stream()))

    def vanilla_get_rng(self, first=0, count=None):
        count = self.G - first if count is None else count
        st = np.empty((count, 625), np.uint32)
        self._chk(self.L.azk_vanilla_get_rng(self.h, first, count, _np(st), _stream()))
        return st

    def vanilla_search(self, n_sims, chunk=None):
        """MCTS.mcts(None, ...) for all G games: begin a search and run n_sims whole simulations (UCB1 walk, expansion,
        random rollout, backup) on the device; `chunk` bounds the simulations per launch."""
        assert n_sims <= self.max_sims
        self.begin_search(None)
        chunk = n_sims if not chunk else int(chunk)
        done = 0
        while done < n_sims:
            k = min(chunk, n_sims - done)
            self._chk(self.L.azk_vanilla_search(self.h, k, _stream()))
            done += k

    def root_stats(self):
        self._chk(self.L.azk_root_stats(self.h, _p(self.pi), _p(self.q), _p(self.root_visit), _stream()))
        return self.pi, self.q, self.root_visit

    def advance(self, uniforms=None, sample_until_move=0):
        if uniforms is not None:
            assert uniforms.dtype == self.torch.float64 and uniforms.is_cuda and uniforms.numel() == self.G
        self._chk(self.L.azk_advance(self.h, _p(uniforms), int(sample_until_move), _p(self.chosen), _p(self.winner),
                                     _p(self.done), _stream()))
        return self.chosen, self.winner, self.done

    def gen_noise(self, seed, first_global_game, move_index, alpha=0.03, want_noise=True, want_uniforms=True):
        torch = self.torch
        noise = torch.empty((self.G, self.action_dim), dtype=torch.float64, device=self.device) if want_noise else None
        uni = torch.empty(self.G, dtype=torch.float64, device=self.device) if want_uniforms else None
        self._chk(self.L.azk_gen_noise(self.h, int(seed), int(first_global_game), int(move_index), float(alpha),
                                       _p(noise), _p(uni), _stream()))
        return noise, uni

    # ---- inspection -----------------------------------------------------------------------------
    def root_children(self, game):
        cap = self.rows * self.cols
        cells = np.empty(cap, np.int32); visits = np.empty(cap, np.int32)
        values = np.empty(cap, np.float64); priors = np.empty(cap, np.float64)
        n = self._chk(self.L.azk_root_children(self.h, int(game), cap, _np(cells), _np(visits), _np(values), _np(priors), _stream()))
        return dict(cell=cells[:n].copy(), visit=visits[:n].astype(np.int64), value=values[:n].copy(), prior=priors[:n].copy())

    def export_tree(self, game, cap=None):
        cap = cap or (1 + self.max_sims * self.rows * self.cols)
        depth = np.empty(cap, np.int32); cell = np.empty(cap, np.int32); visit = np.empty(cap, np.int32)
        value = np.empty(cap, np.float64); prior = np.empty(cap, np.float64)
        n = self._chk(self.L.azk_export_tree(self.h, int(game), cap, _np(depth), _np(cell), _np(visit), _np(value), _np(prior), _stream()))
        n = min(n, cap)
        return dict(depth=depth[:n], cell=cell[:n], visit=visit[:n].astype(np.int64), value=value[:n], prior=prior[:n])

    def counters(self):
        c = Counters()
        self._chk(self.L.azk_get_counters(self.h, C.byref(c), _stream()))
        return c.as_dict()

    def reset_counters(self):
        self._chk(self.L.azk_reset_counters(self.h, _stream()))

    def clear_cache(self):
        """MCTS.cache.clear() - call when the evaluator's weights change."""
        self._chk(self.L.azk_clear_cache(self.h, _stream()))

    def check_error(self):
        rc = self.L.azk_check_device_error(self.h, _stream())
        if rc != 0:
            raise AzkError(f"device error {rc}: {self.L.azk_last_error(self.h).decode()}")


# ---- stateless board-rule kernels (Game statics) over float32 boards [n, F, R, C] on the GPU ------------
def _geom(game, size):
    gid = GAME_ID[game]
    s = int(size or 0)
    return gid, s, s


def _rules_chk(rc):
    if rc != 0:
        raise AzkError(f"libazk rules error {rc}: {lib().azk_last_error(None).decode()}")


def rules_legal_moves(game, boards, size=None):
    """boards: float32 CUDA tensor [n,F,R,C] -> (moves int16 [n, R*C] in the reference's list order, counts int32 [n])."""
    torch = _torch()
    assert boards.is_cuda and boards.dtype == torch.float32 and boards.is_contiguous()
    n, rc = boards.shape[0], boards.shape[2] * boards.shape[3]
    moves = torch.full((n, rc), -1, dtype=torch.int16, device=boards.device)
    counts = torch.zeros(n, dtype=torch.int32, device=boards.device)
    _rules_chk(lib().azk_rules_legal_moves(*_geom(game, size), _p(boards), n, _p(moves), _p(counts), _stream()))
    return moves, counts


def rules_legal_mask(game, boards, action_dim, size=None):
    torch = _torch()
    assert boards.is_cuda and boards.dtype == torch.float32 and boards.is_contiguous()
    n = boards.shape[0]
    mask = torch.zeros((n, action_dim), dtype=torch.uint8, device=boards.device)
    _rules_chk(lib().azk_rules_legal_mask(*_geom(game, size), _p(boards), n, _p(mask), _stream()))
    return mask


def rules_apply_move(game, boards, players, cells, size=None):
    """in-place make_move on each board; returns next player int32 [n]."""
    torch = _torch()
    assert boards.is_cuda and boards.dtype == torch.float32 and boards.is_contiguous()
    n = boards.shape[0]
    players = players.to(torch.int32).contiguous(); cells = cells.to(torch.int32).contiguous()
    nxt = torch.zeros(n, dtype=torch.int32, device=boards.device)
    _rules_chk(lib().azk_rules_apply_move(*_geom(game, size), _p(boards), n, _p(players), _p(cells), _p(nxt), _stream()))
    return nxt


def rules_undo_move(game, boards, current_players, cells, size=None):
    torch = _torch()
    assert boards.is_cuda and boards.dtype == torch.float32 and boards.is_contiguous()
    n = boards.shape[0]
    cp = current_players.to(torch.int32).contiguous(); cells = cells.to(torch.int32).contiguous()
    _rules_chk(lib().azk_rules_undo_move(*_geom(game, size), _p(boards), n, _p(cp), _p(cells), _stream()))


def rules_check_winner(game, boards, players, cells, size=None):
    torch = _torch()
    assert boards.is_cuda and boards.dtype == torch.float32 and boards.is_contiguous()
    n = boards.shape[0]
    players = players.to(torch.int32).contiguous(); cells = cells.to(torch.int32).contiguous()
    out = torch.zeros(n, dtype=torch.int32, device=boards.device)
    _rules_chk(lib().azk_rules_check_winner(*_geom(game, size), _p(boards), n, _p(players), _p(cells), _p(out), _stream()))
    return out


def rules_canonical(game, boards, players, size=None):
    torch = _torch()
    assert boards.is_cuda and boards.dtype == torch.float32 and boards.is_contiguous()
    players = players.to(torch.int32).contiguous()
    out = torch.empty_like(boards)
    _rules_chk(lib().azk_rules_canonical(*_geom(game, size), _p(boards), boards.shape[0], _p(players), _p(out), _stream()))
    return out


def softmax_rows(logits):
    torch = _torch()
    logits = logits.to(torch.float32).contiguous()
    out = torch.empty_like(logits)
    _rules_chk(lib().azk_softmax_rows(_p(logits), logits.shape[0], logits.shape[1], _p(out), _stream()))
    return out


def nn_patch_embed(boards, wt, cpos, ln_w, ln_b, rows, cols, ksize, embed_dim, want_x=True, want_xhat=False, eps=1e-5):
    """Token embedding on the matrix cores (azk_nn_patch_embed).  boards [n,C,R,Cc] bf16|f32 CUDA;
    wt [D,kp] bf16; cpos [T,D] f32.  Returns (x, xhat) bf16 [n,T,D] (None where not requested)."""
    torch = _torch()
    assert boards.is_cuda and boards.is_contiguous() and boards.dtype in (torch.bfloat16, torch.float32)
    n, C = boards.shape[0], boards.shape[1]
    T = rows * cols + 1
    x = torch.empty((n, T, embed_dim), dtype=torch.bfloat16, device=boards.device) if want_x else None
    xh = torch.empty((n, T, embed_dim), dtype=torch.bfloat16, device=boards.device) if want_xhat else None
    rc = lib().azk_nn_patch_embed(_p(boards), 1 if boards.dtype == torch.float32 else 0, _p(wt), _p(cpos), _p(ln_w), _p(ln_b),
                                  _p(x), _p(xh), n, C, rows, cols, ksize, wt.shape[1], embed_dim, float(eps), _stream())
    if rc != 0:
        raise AzkError(f"azk_nn_patch_embed failed ({rc})")
    return x, xh


def nn_cls_attention(xhat, m, c, num_heads):
    """z[b,h,:] = sum_t softmax_t(xhat[b,t,:] . m[.,h,:] + c[.,h]) * xhat[b,t,:]  (azk_nn_cls_attention).
    xhat bf16 [n,T,D]; m f32 [H,D] (shared) or [n,H,D]; c f32 [H] or [n,H].  Returns z bf16 [n,H,D]."""
    torch = _torch()
    assert xhat.is_cuda and xhat.dtype == torch.bfloat16 and xhat.is_contiguous()
    n, T, D = xhat.shape
    per_board = 1 if m.dim() == 3 else 0
    m = m.to(torch.float32).contiguous(); c = c.to(torch.float32).contiguous()
    z = torch.empty((n, num_heads, D), dtype=torch.bfloat16, device=xhat.device)
    rc = lib().azk_nn_cls_attention(_p(xhat), _p(m), _p(c), per_board, _p(z), n, T, D, num_heads, _stream())
    if rc != 0:
        raise AzkError(f"azk_nn_cls_attention failed ({rc})")
    return z


def nn_embed_scores_pool(boards, wt_ext, cpos, score_cpos, score_msum, c, rows, cols, ksize, embed_dim, num_heads, eps=1e-5,
                         count=None, timers=None):
    """Depth-1 folded cls attention in two launches: azk_nn_patch_embed_scores (normalised tokens + per-token head scores,
    the scores as 16 extra MFMA output columns: wt_ext [D+16, kp]) then azk_nn_cls_pool (softmax + weighted token sum).
    Returns z bf16 [n, H, D]."""
    torch = _torch()
    assert boards.is_cuda and boards.is_contiguous() and boards.dtype in (torch.bfloat16, torch.float32)
    n, C = boards.shape[0], boards.shape[1]
    T = rows * cols + 1
    Tp = (T + 15) // 16 * 16
    xh = torch.empty((n, T, embed_dim), dtype=torch.bfloat16, device=boards.device)
    sc = torch.empty((n, num_heads, Tp), dtype=torch.float32, device=boards.device)
    # with a device-side count the rows past it are never written; every later op is row-wise, so they cannot leak
    z = torch.empty((n, num_heads, embed_dim), dtype=torch.bfloat16, device=boards.device)
    L = lib()
    if timers is not None:
        timers[0].start()
    rc = L.azk_nn_patch_embed_scores(_p(boards), 1 if boards.dtype == torch.float32 else 0, _p(wt_ext), _p(cpos), None, None,
                                     _p(xh), _p(score_cpos), _p(score_msum), _p(sc), num_heads, n, C, rows, cols, ksize,
                                     wt_ext.shape[1], embed_dim, float(eps), _p(count), _stream())
    if timers is not None:
        timers[0].stop()
        timers[1].start()
    if rc != 0:
        raise AzkError(f"azk_nn_patch_embed_scores failed ({rc})")
    rc = L.azk_nn_cls_pool(_p(xh), _p(sc), _p(c), _p(z), n, T, embed_dim, num_heads, _p(count), _stream())
    if timers is not None:
        timers[1].stop()
    if rc != 0:
        raise AzkError(f"azk_nn_cls_pool failed ({rc})")
    return z


def nn_embed_pool(boards, wt_ext, cpos_frag, score_frag, score_msum, score_ref, rows, cols, ksize, embed_dim, num_heads,
                  eps=1e-5, count=None, timers=None):
    """Depth-1 folded cls attention in ONE launch (azk_nn_embed_pool): boards -> z bf16 [n, H, D]; tokens never reach HBM.
    cpos_frag / score_frag: per-token constants padded to whole 16-token tiles, in accumulator order (include/azk.h);
    score_ref: [16] static softmax reference per head or None (running maximum)."""
    torch = _torch()
    assert boards.is_cuda and boards.is_contiguous() and boards.dtype in (torch.bfloat16, torch.float32)
    Tp = (rows * cols + 1 + 15) // 16 * 16
    assert cpos_frag.numel() == Tp * embed_dim and score_frag.numel() == Tp * 16 and cpos_frag.is_contiguous() and score_frag.is_contiguous()
    n, C = boards.shape[0], boards.shape[1]
    z = torch.empty((n, num_heads, embed_dim), dtype=torch.bfloat16, device=boards.device)
    fn = lib().azk_nn_embed_pool
    args = (_p(boards), 1 if boards.dtype == torch.float32 else 0, _p(wt_ext), _p(cpos_frag), _p(score_frag), _p(score_msum),
            _p(score_ref), _p(z), num_heads, n, C, rows, cols, ksize, wt_ext.shape[1], embed_dim, float(eps), _p(count), _stream())
    if timers is not None:          # everything is marshalled already: the events bracket the launch alone
        timers[0].start()
    rc = fn(*args)
    if timers is not None:
        timers[0].stop()
    if rc != 0:
        raise AzkError(f"azk_nn_embed_pool failed ({rc})")
    return z


def nn_embed_pool_leaves(src, wt_ext, cpos_frag, score_frag, score_msum, score_ref, ksize, embed_dim, num_heads, eps=1e-5,
                         timers=None):
    """nn_embed_pool over an engine's pending leaves (LeafSource) instead of a compacted board batch: z bf16 [G, H, D],
    rows [0, n_leaf) valid; also writes the engine's leaf slots and the leaf count (no azk_step_gather needed)."""
    torch = _torch()
    z = torch.empty((src.n_games, num_heads, embed_dim), dtype=torch.bfloat16, device=wt_ext.device)
    fn = lib().azk_nn_embed_pool_leaves
    args = (C.byref(src), _p(wt_ext), _p(cpos_frag), _p(score_frag), _p(score_msum), _p(score_ref), _p(z), num_heads, ksize,
            wt_ext.shape[1], embed_dim, float(eps), _stream())
    if timers is not None:
        timers[0].start()
    rc = fn(*args)
    if timers is not None:
        timers[0].stop()
    if rc != 0:
        raise AzkError(f"azk_nn_embed_pool_leaves failed ({rc})")
    return z


class EmbedPoolTables:
    """The tables of azk_nn_embed_pool_compact, kept alive together with their ctypes descriptor.
    t: dict of CUDA tensors (wt_ext bf16 [D+16, kp]; cpos_tok f32 [T+1, D]; score_tok, wconst_tok f32 [T+1, 16]; xnconst_tok bf16
    [T+1, D]; z_all f32 [4, 8, 64, 4]; l_all, score_msum, score_ref f32 [16])."""

    def __init__(self, t, num_heads, ksize, embed_dim, eps=1e-5):
        torch = _torch()
        self.t = {k: v.contiguous() for k, v in t.items()}
        T1 = self.t["cpos_tok"].shape[0]
        for k, (shape, dt) in dict(cpos_tok=((T1, embed_dim), torch.float32), score_tok=((T1, 16), torch.float32),
                                   wconst_tok=((T1, 16), torch.float32), xnconst_tok=((T1, embed_dim), torch.bfloat16),
                                   z_all=((4, 8, 64, 4), torch.float32), l_all=((16,), torch.float32), score_msum=((16,), torch.float32),
                                   score_ref=((16,), torch.float32)).items():
            assert tuple(self.t[k].shape) == shape and self.t[k].dtype == dt and self.t[k].is_cuda, k
        assert self.t["wt_ext"].dtype == torch.bfloat16 and self.t["wt_ext"].shape[0] == embed_dim + 16 and embed_dim == 512
        # MFMA fragment order [33 column tiles][kp/32][64 lanes][8] (include/azk.h)
        w = self.t["wt_ext"]
        kp = w.shape[1]
        ct, l = torch.arange(33, device=w.device)[:, None], torch.arange(64, device=w.device)[None, :]
        col = torch.where(ct < 32, 128 * (ct >> 3) + 8 * (l & 15) + (ct & 7), 512 + (l & 15))           # [33, 64]
        kidx = (32 * torch.arange(kp // 32, device=w.device)[:, None, None] + 8 * (l[0] >> 4)[None, :, None]
                + torch.arange(8, device=w.device)[None, None, :])                                      # [KS, 64, 8]
        self.t["wt_frag"] = w[col[:, None, :, None], kidx[None]].contiguous()                           # [33, KS, 64, 8]
        self.tokens, self.num_heads, self.embed_dim = T1 - 1, num_heads, embed_dim
        self.c = EmbedPoolConsts(*[self.t[k].data_ptr() for k in ("wt_frag", "cpos_tok", "score_tok", "wconst_tok", "xnconst_tok", "z_all",
                                                                 "l_all", "score_msum", "score_ref")],
                                 num_heads, ksize, self.t["wt_ext"].shape[1], embed_dim, float(eps), None)
        self.work_stats = None

    def enable_work_stats(self):
        """Device counters [boards evaluated, 16-token tiles evaluated] (int64 [2]), bumped by every launch from now on."""
        if self.work_stats is None:
            self.work_stats = _torch().zeros(2, dtype=_torch().int64, device=self.t["cpos_tok"].device)
            self.c.work_stats = self.work_stats.data_ptr()
        return self.work_stats


def new_sched(device):
    """The work-queue words of the compacting kernel: int32 [2], zero; one buffer per stream that may run it concurrently."""
    return _torch().zeros(2, dtype=_torch().int32, device=device)


def nn_embed_pool_compact(boards, tables, rows, cols, sched, count=None, timers=None):
    """azk_nn_embed_pool_compact: boards [n, C, R, Cc] bf16 / f32 -> z bf16 [n, H, D], evaluating only the tokens a stone can reach."""
    torch = _torch()
    assert boards.is_cuda and boards.is_contiguous() and boards.dtype in (torch.bfloat16, torch.float32)
    assert rows * cols + 1 == tables.tokens and sched.dtype == torch.int32 and sched.numel() >= 2
    n, Cc = boards.shape[0], boards.shape[1]
    z = torch.empty((n, tables.num_heads, tables.embed_dim), dtype=torch.bfloat16, device=boards.device)
    fn = lib().azk_nn_embed_pool_compact
    args = (_p(boards), 1 if boards.dtype == torch.float32 else 0, C.byref(tables.c), _p(z), n, Cc, rows, cols, _p(count), _p(sched), _stream())
    if timers is not None:
        timers[0].start()
    rc = fn(*args)
    if timers is not None:
        timers[0].stop()
    if rc != 0:
        raise AzkError(f"azk_nn_embed_pool_compact failed ({rc})")
    return z


def nn_embed_pool_compact_leaves(src, tables, sched, timers=None):
    """azk_nn_embed_pool_compact over an engine's pending leaves (LeafSource): z bf16 [G, H, D], rows [0, n_leaf) valid."""
    torch = _torch()
    assert src.rows * src.cols + 1 == tables.tokens and sched.dtype == torch.int32 and sched.numel() >= 2
    z = torch.empty((src.n_games, tables.num_heads, tables.embed_dim), dtype=torch.bfloat16, device=sched.device)
    fn = lib().azk_nn_embed_pool_compact_leaves
    args = (C.byref(src), C.byref(tables.c), _p(z), _p(sched), _stream())
    if timers is not None:
        timers[0].start()
    rc = fn(*args)
    if timers is not None:
        timers[0].stop()
    if rc != 0:
        raise AzkError(f"azk_nn_embed_pool_compact_leaves failed ({rc})")
    return z


EMBED_FOLD_ROW = 384        # include/azk.h AZK_EMBED_FOLD_ROW
EMBED_FOLD_MAX_SLOTS = 8192 # include/azk.h AZK_EMBED_FOLD_MAX_SLOTS: pending-leaf slots azk_nn_embed_fold_leaves ranks in LDS


class EmbedFoldTables:
    """Tables of azk_nn_embed_fold and the weight of the batched GEMM that follows it, from PolicyValueNet.fold_u's float64 operands
    (r: G [64, 64], ext [64, 16], U2 [T, 64], nt [T], sct [T, H], Dtab [T, 512], M [512, 64], rstdc [T], ref [H], wc [T, H], uall [512],
    lall [H]); kept alive with the ctypes descriptor.  `weight`: H blocks of [64][EMBED_FOLD_ROW] in azk_nn_tail_gemm's packing."""

    def __init__(self, r, num_heads, ksize, embed_dim, device, eps=1e-5, exact=False):
        """exact: the tables of azk_nnx_embed_fold (float32 rows, 1 / L in one slot) and the weight in azk_nnx_gemm_h's (hi, lo) planes."""
        torch = _torch()
        self.exact = bool(exact)
        assert embed_dim == 512 and embed_dim // num_heads == 64
        H, D = num_heads, embed_dim
        dev = torch.device(device)
        T = r["U2"].shape[0]
        assert T + 3 <= 256 and r["G"].shape == (64, 64)
        # power-of-two scales that put the largest entry's hi term near 2^10: hi and lo both normal fp16 for entries down to ~1e-7 of it
        sc = lambda t: 2.0 ** (10 - math.ceil(math.log2(max(float(t.abs().max()), 1e-30))))
        gs, es = sc(r["G"]), sc(r["ext"])

        def frags(mat, scale, ncol):                    # mat [64 k, 16 ncol columns] -> [2 (hi, lo)][ncol][2 k-steps][64 lanes][8]
            hi, lo = split_fp16(mat.to(dev), scale)
            l = torch.arange(64, device=dev)
            col = 16 * torch.arange(ncol, device=dev)[:, None] + (l & 15)[None, :]                                      # [ncol, 64]
            kidx = 32 * torch.arange(2, device=dev)[:, None, None] + 8 * (l >> 4)[None, :, None] + torch.arange(8, device=dev)[None, None, :]   # [2, 64, 8]
            pick = lambda m_: m_[kidx[None], col[:, None, :, None]]                                                     # [ncol, 2, 64, 8]
            return torch.stack([pick(hi), pick(lo)]).contiguous()
        t = {}
        t["g_frag"] = frags(r["G"], gs, 4)              # element [term][q][s][l][i] = G[32 s + 8 (l>>4) + i][16 q + (l&15)] (G is symmetric)
        t["e_frag"] = frags(r["ext"], es, 1)
        f32 = lambda x: x.to(dev, torch.float32).contiguous()
        u2 = torch.zeros(T + 1, 64, dtype=torch.float64, device=dev)
        u2[:T] = r["U2"].to(dev)
        t["u2_tok"] = f32(u2)
        st = torch.zeros(T + 1, 16, dtype=torch.float64, device=dev)
        st[:T, :H], st[:T, 15] = r["sct"].to(dev), r["nt"].to(dev)
        st[T, :H], st[T, 15] = -1e30, float(D)
        t["score_tok"] = f32(st)
        wct = torch.zeros(T + 1, 16, dtype=torch.float64, device=dev)
        wct[:T, :H], wct[:T, 15] = r["wc"].to(dev), r["rstdc"].to(dev)
        wct[T, 15] = 1.0
        t["wconst_tok"] = f32(wct)
        la = torch.zeros(16, dtype=torch.float64, device=dev)
        la[:H] = r["lall"].to(dev)
        t["l_all"] = f32(la)
        ref = torch.full((16,), 1e30, dtype=torch.float64, device=dev)
        ref[:H] = r["ref"].to(dev)
        t["score_ref"] = f32(ref)
        t["inv_scales"] = torch.tensor([1.0 / gs, 1.0 / es], dtype=torch.float32, device=dev)
        self.t = t
        # the GEMM weight: per head [64 outputs][EMBED_FOLD_ROW]: D_t (t < T), U_all as bf16 hi at T and T + 1, its remainder at T + 2,
        # M_h at [256, 320)
        W = torch.zeros(H, 64, EMBED_FOLD_ROW, dtype=torch.float64, device=dev)
        W[:, :, :T] = r["Dtab"].to(dev).view(T, H, 64).permute(1, 2, 0)
        ua = r["uall"].to(dev).view(H, 64)
        ua_hi = ua.to(torch.bfloat16).double()
        if exact:
            W[:, :, T] = ua                                                  # (float32 rows carry 1 / L in one slot, the planes U_all in full)
        else:
            W[:, :, T], W[:, :, T + 1], W[:, :, T + 2] = ua_hi, ua_hi, ua - ua_hi
        W[:, :, 256:320] = r["M"].to(dev).view(H, 64, 64)
        self.weight_f64 = W
        if exact:
            t["weight"] = torch.stack([pack_linear_weight_h(W[h])[0] for h in range(H)]).contiguous()
        else:
            t["weight"] = torch.cat([pack_linear_weight(W[h].float()).reshape(-1) for h in range(H)])  # (in .t: promoted in place with the tables)
        self.weight = t["weight"]
        self.tokens, self.num_heads, self.embed_dim = T, H, D
        self.c = EmbedFoldConsts(*[t[k].data_ptr() for k in ("g_frag", "e_frag", "u2_tok", "score_tok", "wconst_tok", "l_all", "score_ref",
                                                             "inv_scales")], H, ksize, D, float(eps), None)
        self.work_stats = None

    def enable_work_stats(self):
        if self.work_stats is None:
            self.work_stats = _torch().zeros(2, dtype=_torch().int64, device=self.t["u2_tok"].device)
            self.c.work_stats = self.work_stats.data_ptr()
        return self.work_stats


def nn_embed_fold(boards, tables, rows, cols, sched, count=None, timers=None):
    """azk_nn_embed_fold: boards [n, C, R, Cc] bf16 / f32 -> bf16 [n, H, EMBED_FOLD_ROW] (token weights / L, 1 / L, pooled patch / L)."""
    torch = _torch()
    assert boards.is_cuda and boards.is_contiguous() and boards.dtype in (torch.bfloat16, torch.float32)
    assert rows * cols + 1 == tables.tokens and sched.dtype == torch.int32 and sched.numel() >= 2
    n, Cc = boards.shape[0], boards.shape[1]
    out = torch.empty((n, tables.num_heads, EMBED_FOLD_ROW), dtype=torch.bfloat16, device=boards.device)
    args = (_p(boards), 1 if boards.dtype == torch.float32 else 0, C.byref(tables.c), _p(out), n, Cc, rows, cols, _p(count), _p(sched), _stream())
    if timers is not None:
        timers[0].start()
    rc = lib().azk_nn_embed_fold(*args)
    if timers is not None:
        timers[0].stop()
    if rc != 0:
        raise AzkError(f"azk_nn_embed_fold failed ({rc})")
    return out


def nn_embed_fold_leaves(src, tables, sched, timers=None):
    """azk_nn_embed_fold over an engine's pending leaves (LeafSource): bf16 [G, H, EMBED_FOLD_ROW], rows [0, n_leaf) valid."""
    torch = _torch()
    assert src.rows * src.cols + 1 == tables.tokens and sched.dtype == torch.int32 and sched.numel() >= 2
    out = torch.empty((src.n_games, tables.num_heads, EMBED_FOLD_ROW), dtype=torch.bfloat16, device=sched.device)
    args = (C.byref(src), C.byref(tables.c), _p(out), _p(sched), _stream())
    if timers is not None:
        timers[0].start()
    rc = lib().azk_nn_embed_fold_leaves(*args)
    if timers is not None:
        timers[0].stop()
    if rc != 0:
        raise AzkError(f"azk_nn_embed_fold_leaves failed ({rc})")
    return out


def nnx_embed_fold(boards, tables, rows, cols, sched, count=None, timers=None):
    """azk_nnx_embed_fold: boards [n, C, R, Cc] bf16 / f32 -> float32 [n, H, EMBED_FOLD_ROW] (token weights / L, 1 / L, pooled patch / L)."""
    torch = _torch()
    assert tables.exact and boards.is_cuda and boards.is_contiguous() and boards.dtype in (torch.bfloat16, torch.float32)
    assert rows * cols + 1 == tables.tokens and sched.dtype == torch.int32 and sched.numel() >= 2
    n, Cc = boards.shape[0], boards.shape[1]
    out = torch.empty((n, tables.num_heads, EMBED_FOLD_ROW), dtype=torch.float32, device=boards.device)
    args = (_p(boards), 1 if boards.dtype == torch.float32 else 0, C.byref(tables.c), _p(out), n, Cc, rows, cols, _p(count), _p(sched), _stream())
    if timers is not None:
        timers[0].start()
    rc = lib().azk_nnx_embed_fold(*args)
    if timers is not None:
        timers[0].stop()
    if rc != 0:
        raise AzkError(f"azk_nnx_embed_fold failed ({rc})")
    return out


def nnx_embed_fold_leaves(src, tables, sched, timers=None):
    """azk_nnx_embed_fold over an engine's pending leaves (LeafSource): float32 [G, H, EMBED_FOLD_ROW], rows [0, n_leaf) valid."""
    torch = _torch()
    assert tables.exact and src.rows * src.cols + 1 == tables.tokens and sched.dtype == torch.int32 and sched.numel() >= 2
    out = torch.empty((src.n_games, tables.num_heads, EMBED_FOLD_ROW), dtype=torch.float32, device=sched.device)
    args = (C.byref(src), C.byref(tables.c), _p(out), _p(sched), _stream())
    if timers is not None:
        timers[0].start()
    rc = lib().azk_nnx_embed_fold_leaves(*args)
    if timers is not None:
        timers[0].stop()
    if rc != 0:
        raise AzkError(f"azk_nnx_embed_fold_leaves failed ({rc})")
    return out


TAIL_BF16, TAIL_GELU, TAIL_RESID, TAIL_HEADS = 0, 1, 2, 3


def nn_tail_gemm(a, w_packed, n_out, k, epilogue=TAIL_BF16, nbatch=1, a_batch_stride=0, bias=None, out=None, resid=None,
                 a_stats=None, stats_out=None, logits=None, values=None, action_dim=0, count=None, eps=1e-5, col_sums=None, lds=False):
    """One link of the cls-row tail (azk_nn_tail_gemm): a bf16 [m, lda] x packed weights -> out bf16 [m, nbatch * n_out] (or the
    heads' float32 logits / values).  a_stats [m, groups, 2]: A is LayerNorm(a) (affine folded by the caller), its row statistics
    coming from the producer's stats_out."""
    torch = _torch()
    assert a.dtype == torch.bfloat16 and a.stride(-1) == 1 and a.dim() == 2
    d = TailGemm()
    d.a_bf16, d.lda, d.a_batch_stride, d.w_packed = a.data_ptr(), a.stride(0), int(a_batch_stride), w_packed.data_ptr()
    d.m, d.n_out, d.k, d.nbatch = a.shape[0], int(n_out), int(k), int(nbatch)
    d.n_valid = count.data_ptr() if count is not None else None
    d.bias = bias.data_ptr() if bias is not None else None
    d.layernorm_a, d.epilogue, d.ln_eps = (1 if a_stats is not None else 0), int(epilogue), float(eps)
    if a_stats is not None:
        assert a_stats.dtype == torch.float32 and a_stats.dim() == 3 and a_stats.shape[2] == 2 and a_stats.is_contiguous()
        d.a_stats, d.a_stats_groups = a_stats.data_ptr(), a_stats.shape[1]
    if stats_out is not None:
        assert stats_out.dtype == torch.float32 and tuple(stats_out.shape) == (a.shape[0], nbatch * n_out // 64, 2) and stats_out.is_contiguous()
        d.stats_out = stats_out.data_ptr()
    if out is not None:
        assert out.dtype == torch.bfloat16 and out.stride(-1) == 1
        d.out_bf16, d.ldo = out.data_ptr(), out.stride(0)
    if resid is not None:
        d.resid_bf16, d.ldr = resid.data_ptr(), resid.stride(0)
    if logits is not None:
        assert logits.dtype == torch.float32 and values.dtype == torch.float32 and logits.is_contiguous()
        d.logits_out, d.values_out, d.action_dim = logits.data_ptr(), values.data_ptr(), int(action_dim)
    if col_sums is not None:
        assert col_sums.dtype == torch.float32 and col_sums.is_contiguous() and col_sums.numel() >= nbatch * n_out
        d.a_col_sums = col_sums.data_ptr()
    rc = (lib().azk_nn_tail_gemm_lds if lds else lib().azk_nn_tail_gemm)(C.byref(d), _stream())
    if rc != 0:
        raise AzkError(f"azk_nn_tail_gemm{'_lds' if lds else ''} failed ({rc})")


TOK_BF16, TOK_GELU, TOK_RESID, TOK_F32 = 0, 1, 2, 4


def pack_linear_weight128(w):
    """pack_linear_weight with the output dimension padded (zero rows) to a multiple of 128: the operand of nn_gemm_tok."""
    torch = _torch()
    n_out, k = w.shape
    npad = (n_out + 127) // 128 * 128
    wp = torch.zeros(npad, k, dtype=torch.float32, device=w.device)
    wp[:n_out] = w.float()
    return pack_linear_weight(wp)


def nn_gemm_tok(a, w_packed, n_out, epilogue=TOK_BF16, bias=None, out=None, resid=None, count=None):
    """out[m][n_out] = a[m][k] W^T (+ bias) through an epilogue (azk_nn_gemm_tok, csrc/azk_block.hip).  a: bf16 [m, k] (row stride
    a.stride(0)); w_packed: pack_linear_weight128(W); n_out: the padded output width (a multiple of 128); out: bf16 (float32 for
    TOK_F32) [m, >= n_out], allocated when None; resid: bf16 [m, >= n_out] for TOK_RESID."""
    torch = _torch()
    assert a.dtype == torch.bfloat16 and a.dim() == 2 and a.stride(1) == 1 and n_out % 128 == 0
    m, k = a.shape
    if out is None:
        out = torch.empty((m, n_out), dtype=torch.float32 if epilogue == TOK_F32 else torch.bfloat16, device=a.device)
    assert out.stride(1) == 1 and out.dtype == (torch.float32 if epilogue == TOK_F32 else torch.bfloat16)
    d = GemmTok()
    d.a_bf16, d.lda, d.w_packed, d.m, d.n_out, d.k = a.data_ptr(), a.stride(0), w_packed.data_ptr(), m, int(n_out), k
    d.n_valid = count.data_ptr() if count is not None else None
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() >= n_out
        d.bias = bias.data_ptr()
    d.epilogue, d.out, d.ldo = int(epilogue), out.data_ptr(), out.stride(0)
    if resid is not None:
        assert resid.dtype == torch.bfloat16 and resid.stride(1) == 1
        d.resid_bf16, d.ldr = resid.data_ptr(), resid.stride(0)
    rc = lib().azk_nn_gemm_tok(C.byref(d), _stream())
    if rc != 0:
        raise AzkError(f"azk_nn_gemm_tok failed ({rc})")
    return out


def nn_attention_tok(qkv, n_boards, tokens, embed_dim, num_heads, out=None, count=None):
    """softmax(q k^T / sqrt(dh)) v over all tokens of every board (azk_nn_attention_tok).  qkv: bf16 [n_boards * tokens, 3 embed_dim]."""
    torch = _torch()
    assert qkv.dtype == torch.bfloat16 and qkv.is_contiguous() and qkv.shape == (n_boards * tokens, 3 * embed_dim)
    if out is None:
        out = torch.empty((n_boards * tokens, embed_dim), dtype=torch.bfloat16, device=qkv.device)
    assert out.is_contiguous() and out.dtype == torch.bfloat16
    rc = lib().azk_nn_attention_tok(_p(qkv), _p(out), int(n_boards), int(tokens), int(embed_dim), int(num_heads), _p(count), _stream())
    if rc != 0:
        raise AzkError(f"azk_nn_attention_tok failed ({rc})")
    return out


def packed_weight_col_sums(w_packed, n_out, k):
    """Column sums sum_k W[j][k] of a pack_linear_weight() tensor's bf16 values (float64 sum, rounded once): the
    a_col_sums operand of azk_nn_tail_gemm_lds (LayerNorm applied in the epilogue)."""
    torch = _torch()
    npad = (n_out + 63) // 64 * 64
    # [g, s, c, l4, l15, i] -> [g, l15, c, s, l4, i]: column 64 g + 4 l15 + c, k = 32 s + 8 l4 + i
    w = w_packed.view(npad // 64, k // 32, 4, 4, 16, 8).permute(0, 4, 2, 1, 3, 5).reshape(npad, k)
    return w.double().sum(1).float().contiguous()


class _ReplayUnpickler(__import__("pickle").Unpickler):
    """pickle.Unpickler limited to the globals of replay_buffer.py's file format (replay_buffer.py:37-65)."""
    _ALLOWED = {("collections", "deque"), ("numpy", "ndarray"), ("numpy", "dtype"),
                ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
                ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
                ("numpy.core.numeric", "_frombuffer"), ("numpy._core.numeric", "_frombuffer")}

    def find_class(self, module, name):
        if (module, name) in self._ALLOWED:
            return super().find_class(module, name)
        import pickle
        raise pickle.UnpicklingError(f"replay file refers to {module}.{name}: not part of the replay format, refused")


class DeviceReplay:
    """HBM-resident ring of (state, pi, z) tuples: the device form of replay_buffer.ReplayBuffer (deque(maxlen),
    replay_buffer.py:7-13).  Filled by Engine.emit_finished; `sample` draws uniformly without replacement
    (replay_buffer.py:15-25) and returns float32 CUDA tensors ready for the training step."""

    def __init__(self, capacity, planes, rows, cols, action_dim, device=0):
        torch = _torch()
        self.torch, self.capacity = torch, int(capacity)
        dev = torch.device("cuda", device) if isinstance(device, int) else device
        self.states = torch.zeros((capacity, planes, rows, cols), dtype=torch.float32, device=dev)
        self.pis = torch.zeros((capacity, action_dim), dtype=torch.float64, device=dev)
        self.zs = torch.zeros(capacity, dtype=torch.float32, device=dev)
        self.cursor = torch.zeros(1, dtype=torch.int64, device=dev)

    def size(self):
        return min(int(self.cursor.item()), self.capacity)

    def sample(self, batch_size):
        n = self.size()
        if batch_size > n:          # np.random.choice(len, batch_size, replace=False) raises the same way (replay_buffer.py:16)
            raise ValueError(f"cannot sample {batch_size} tuples without replacement from a ring holding {n}")
        idx = self.torch.randperm(n, device=self.states.device)[:batch_size]
        return self.states[idx], self.pis[idx].float(), self.zs[idx][:, None]

    # ---- interchange with the reference's host-side ReplayBuffer (replay_buffer.py) ------------------------------
    def add(self, state, policy_distribution, reward):
        """ReplayBuffer.add (replay_buffer.py:12): append one tuple from the host (reward: float or [float])."""
        torch = self.torch
        i = int(self.cursor.item()) % self.capacity
        self.states[i] = torch.as_tensor(np.asarray(state, np.float32))
        self.pis[i] = torch.as_tensor(np.asarray(policy_distribution, np.float64))
        self.zs[i] = float(np.asarray(reward, np.float32).reshape(-1)[0])
        self.cursor += 1

    def to_reference_deque(self):
        """The ring as the reference keeps it: deque(maxlen=capacity) of (state float32 [F,R,C], pi float64 [A], [z]) tuples,
        oldest first (train.save_data_to_buffer's element format, train.py:30-49)."""
        from collections import deque
        n, cur = self.size(), int(self.cursor.item())
        order = [(cur - n + j) % self.capacity for j in range(n)]
        s, p, z = self.states.cpu().numpy(), self.pis.cpu().numpy(), self.zs.cpu().numpy()
        return deque(((s[i].copy(), p[i].copy(), [float(z[i])]) for i in order), maxlen=self.capacity)

    def save_pickle(self, filename):
        """ReplayBuffer.save_pickle's file format (replay_buffer.py:37-54): pickle.dump of the deque."""
        import os, pickle
        folder = os.path.dirname(filename)
        if folder:
            os.makedirs(folder, exist_ok=True)
        with open(filename, "wb") as fh:
            pickle.dump(self.to_reference_deque(), fh)

    def load_pickle(self, filename):
        """Refill the ring from a file in that format (one this class or the reference's ReplayBuffer wrote).  The file is
        read by a restricted unpickler that can only build what the format holds - a deque of (ndarray, ndarray, list of
        float) - and refuses every other global, so a crafted file cannot run code."""
        with open(filename, "rb") as fh:
            items = list(_ReplayUnpickler(fh).load())[-self.capacity:]
        self.cursor.zero_()
        if items:
            torch = self.torch
            n = len(items)
            self.states[:n] = torch.as_tensor(np.stack([np.asarray(t[0], np.float32) for t in items]))
            self.pis[:n] = torch.as_tensor(np.stack([np.asarray(t[1], np.float64) for t in items]))
            self.zs[:n] = torch.as_tensor(np.array([float(np.asarray(t[2], np.float32).reshape(-1)[0]) for t in items], np.float32))
            self.cursor += n


def nn_heads_finalize(heads, action_dim, logits_out, values_out, count=None):
    """heads bf16 [n, ld] (merged policy/value GEMM) -> logits_out f32 [n, A], values_out f32 [n] = tanh(raw), one launch."""
    torch = _torch()
    assert heads.dtype == torch.bfloat16 and heads.is_contiguous() and logits_out.dtype == torch.float32 and values_out.dtype == torch.float32
    rc = lib().azk_nn_heads_finalize(_p(heads), heads.shape[1], int(action_dim), heads.shape[0], _p(logits_out), _p(values_out),
                                     _p(count), _stream())
    if rc != 0:
        raise AzkError(f"azk_nn_heads_finalize failed ({rc})")


def nn_layernorm_rows(x, w, b, eps=1e-5, add_bias=None, count=None):
    """LayerNorm over the rows of bf16 x [n, D] -> new bf16 tensor; with add_bias, x becomes x + add_bias in place."""
    torch = _torch()
    assert x.dtype == torch.bfloat16 and x.is_contiguous() and w.dtype == torch.float32 and b.dtype == torch.float32
    y = torch.empty_like(x)
    rc = lib().azk_nn_layernorm_rows(_p(x), _p(w), _p(b), float(eps), _p(y), _p(add_bias), x.shape[0], x.shape[1], _p(count), _stream())
    if rc != 0:
        raise AzkError(f"azk_nn_layernorm_rows failed ({rc})")
    return y


def mt_state_from_numpy(state=None):
    """np.random.get_state() (or a RandomState's) -> the uint32[625] an engine game takes (key + position)."""
    st = np.random.get_state() if state is None else state
    assert st[0] == "MT19937"
    out = np.empty(625, np.uint32)
    out[:624] = st[1]
    out[624] = st[2]
    return out


def mt_state_to_numpy(words, template=None):
    """uint32[625] read back from the engine -> a tuple for np.random.set_state (Gaussian cache fields from `template`)."""
    t = np.random.get_state() if template is None else template
    return ("MT19937", np.asarray(words[:624], np.uint32), int(words[624]), t[3], t[4])


def pack_linear_weight(w):
    """nn.Linear weight [n_out, k] (any float dtype, any device) -> bf16 tensor in azk_nn_gemm_rows' fragment order
    (n_out padded with zero rows to a multiple of 64; k must be a multiple of 32)."""
    torch = _torch()
    n_out, k = w.shape
    assert k % 32 == 0
    npad = (n_out + 63) // 64 * 64
    wp = torch.zeros(npad, k, dtype=torch.float32, device=w.device)
    wp[:n_out] = w.float()
    # [g, l15, c, s, l4, i] -> [g, s, c, l4, l15, i]
    return wp.view(npad // 64, 16, 4, k // 32, 4, 8).permute(0, 3, 2, 4, 1, 5).contiguous().to(torch.bfloat16)


def nn_gemm_rows(a, w_packed, n_out, ksplit=1, partials=None, bias=None, gelu_out=None, count=None):
    """a bf16 [m, k] (row stride = a.stride(0)) times a packed weight: float32 partial planes [ksplit, m, n_out] or
    bf16 GELU(a W^T + bias)."""
    torch = _torch()
    assert a.dtype == torch.bfloat16 and a.stride(1) == 1
    m, k = a.shape
    rc = lib().azk_nn_gemm_rows(_p(a), a.stride(0), _p(w_packed), m, int(n_out), k, int(ksplit), _p(partials), _p(bias), _p(gelu_out),
                                _p(count), _stream())
    if rc != 0:
        raise AzkError(f"azk_nn_gemm_rows failed ({rc})")


def nn_layernorm_sum(partials, w, b, y, bias=None, resid=None, add_bias=None, x_out=None, eps=1e-5, count=None):
    nsplit, m, d = partials.shape
    rc = lib().azk_nn_layernorm_sum(_p(partials), nsplit, m, _p(bias), _p(resid), _p(w), _p(b), float(eps), _p(y), _p(add_bias), _p(x_out),
                                    y.shape[0], d, _p(count), _stream())
    if rc != 0:
        raise AzkError(f"azk_nn_layernorm_sum failed ({rc})")


def nn_heads_finalize_sum(partials, bias, action_dim, logits_out, values_out, count=None):
    nsplit, m, ld = partials.shape
    rc = lib().azk_nn_heads_finalize_sum(_p(partials), nsplit, m, ld, _p(bias), int(action_dim), logits_out.shape[0], _p(logits_out),
                                         _p(values_out), _p(count), _stream())
    if rc != 0:
        raise AzkError(f"azk_nn_heads_finalize_sum failed ({rc})")


def nn_ln_heads(x, ln_w, ln_b, w_packed, bias, action_dim, logits_out, values_out, eps=1e-5, count=None):
    """Final LayerNorm + merged policy/value head + finalize in one launch: x bf16 [n, D] -> logits f32 [n, A], values f32 [n]."""
    torch = _torch()
    assert x.dtype == torch.bfloat16 and x.is_contiguous()
    n, d = x.shape
    rc = lib().azk_nn_ln_heads(_p(x), _p(ln_w), _p(ln_b), float(eps), _p(w_packed), _p(bias), n, d, bias.numel(), int(action_dim),
                               _p(logits_out), _p(values_out), _p(count), _stream())
    if rc != 0:
        raise AzkError(f"azk_nn_ln_heads failed ({rc})")


# ---- fp32-accurate network path (csrc/azk_nnx.hip) ----------------------------------------------------------------------
def split_fp16(x64, scale):
    """float64 tensor -> (hi, lo) fp16 tensors with (hi + lo) / scale = x to 22 significant bits (two round-to-nearest steps)."""
    torch = _torch()
    xs = x64.double() * float(scale)
    hi = xs.to(torch.float16)
    lo = (xs - hi.double()).to(torch.float16)
    return hi, lo


class EmbedPoolXTables:
    """Tables of azk_nnx_embed_pool, kept alive with their ctypes descriptor.  t: dict of CUDA tensors - wt_ext float64 [D+16, kp] (conv
    weight, folded score rows, mean row); cpos_tok, xnconst_tok f32 [T+1, D]; score_tok, wconst_tok f32 [T+1, 16]; z_all f32 [16, D]
    (head-major, converted to accumulator order here); l_all, score_msum, score_ref f32 [16]."""

    WT_SCALE = 4096.0      # weights (|w| < 0.2 here) x 2^12: hi and lo fp16 terms both in the normal range down to |w| ~ 6e-5
    WC_SCALE, XNC_SCALE = 64.0, 16.0     # constant tokens' softmax weights (in (0, 1]) and normalised rows (|x| < 23) as two fp16 terms each

    def __init__(self, t, num_heads, ksize, embed_dim, eps=1e-5):
        torch = _torch()
        assert embed_dim == 512
        dev = t["cpos_tok"].device
        self.t = {k: v.contiguous() for k, v in t.items() if k not in ("wt_ext", "z_all")}      # (z_all is rebuilt below from the split constant terms)
        T1 = self.t["cpos_tok"].shape[0]
        for k, shape in dict(cpos_tok=(T1, embed_dim), xnconst_tok=(T1, embed_dim), score_tok=(T1, 16), wconst_tok=(T1, 16), l_all=(16,),
                             score_msum=(16,), score_ref=(16,)).items():
            assert tuple(self.t[k].shape) == shape and self.t[k].dtype == torch.float32 and self.t[k].is_cuda, k
        w = t["wt_ext"].double()
        assert w.shape[0] == embed_dim + 16 and w.shape[1] % 32 == 0 and float(w.abs().max()) * self.WT_SCALE < 60000.0
        kp = w.shape[1]
        hi, lo = split_fp16(w, self.WT_SCALE)
        ct, l = torch.arange(33, device=dev)[:, None], torch.arange(64, device=dev)[None, :]
        col = torch.where(ct < 32, 64 * (ct >> 2) + 4 * (l & 15) + (ct & 3), 512 + (l & 15))                 # [33, 64]
        kidx = (32 * torch.arange(kp // 32, device=dev)[:, None, None] + 8 * (l[0] >> 4)[None, :, None]
                + torch.arange(8, device=dev)[None, None, :])                                             # [KS, 64, 8]
        fh, fl = hi[col[:, None, :, None], kidx[None]], lo[col[:, None, :, None], kidx[None]]              # [33, KS, 64, 8]
        self.t["wt_frag"] = torch.stack([fh, fl], dim=2).contiguous()                                      # [33, KS, 2, 64, 8]
        # the constant tokens' part of the weighted token sum runs on v_mfma_f32_16x16x32_f16 with both operands as (hi, lo) fp16
        # pairs (all four partial products in two instructions): -wconst * WC_SCALE per (token, head) as one uint32 (hi | lo << 16),
        # xnconst * XNC_SCALE per (token, 4 columns) as 16 bytes (hi0..hi3, lo0..lo3); the accumulators then run in units of
        # POOL_SCALE = WC_SCALE * XNC_SCALE (the kernel scales the stone-touched tokens' weights by it, z_all comes pre-scaled)
        wc, xnc = self.t["wconst_tok"].double(), self.t["xnconst_tok"].double()
        assert float(wc.max()) <= 1.0 + 1e-6 and float(xnc.abs().max()) * self.XNC_SCALE < 60000.0
        wh, wl = split_fp16(-wc, self.WC_SCALE)
        self.t["wconst_h16"] = (wh.view(torch.int16).to(torch.int32) & 0xffff | (wl.view(torch.int16).to(torch.int32) << 16)).contiguous()
        xh, xl = split_fp16(xnc, self.XNC_SCALE)
        self.t["xnconst_h16"] = torch.cat([xh.view(T1, embed_dim // 4, 4), xl.view(T1, embed_dim // 4, 4)], dim=2).contiguous()   # [T+1, 128, 8] fp16
        self.pool_scale = self.WC_SCALE * self.XNC_SCALE
        wc_eff = -(wh.double() + wl.double()) / self.WC_SCALE                # what the kernel subtracts per dirty token ...
        xnc_eff = (xh.double() + xl.double()) / self.XNC_SCALE
        zall = (wc_eff.t() @ xnc_eff) * self.pool_scale                      # ... so the sum over ALL tokens uses the same terms
        # z_all [16 heads, D] -> accumulator order [g][q][lane = 16 l4 + l15][j]: head 4 l4 + j, column 64 g + 4 l15 + q
        za = zall.float().view(4, 4, 8, 16, 4)                                # [l4, j, g, l15, q]
        self.t["z_all"] = za.permute(2, 4, 0, 3, 1).reshape(8, 4, 64, 4).contiguous()
        self.tokens, self.num_heads, self.embed_dim = T1 - 1, num_heads, embed_dim
        self.c = EmbedPoolXConsts(*[self.t[k].data_ptr() for k in ("wt_frag", "cpos_tok", "score_tok", "wconst_tok", "xnconst_h16", "z_all",
                                                                  "l_all", "score_msum", "score_ref")],
                                  num_heads, ksize, kp, embed_dim, float(eps), float(self.WT_SCALE), None,
                                  self.t["wconst_h16"].data_ptr(), float(self.pool_scale))
        self.work_stats = None

    def enable_work_stats(self):
        if self.work_stats is None:
            self.work_stats = _torch().zeros(2, dtype=_torch().int64, device=self.t["cpos_tok"].device)
            self.c.work_stats = self.work_stats.data_ptr()
        return self.work_stats


def nnx_embed_pool(boards, tables, rows, cols, sched, count=None, timers=None):
    """azk_nnx_embed_pool: boards [n, C, R, Cc] bf16 / f32 (values 0 / 1) -> z float32 [n, H, 512]."""
    torch = _torch()
    assert boards.is_cuda and boards.is_contiguous() and boards.dtype in (torch.bfloat16, torch.float32)
    assert rows * cols + 1 == tables.tokens and sched.dtype == torch.int32 and sched.numel() >= 2
    n, Cc = boards.shape[0], boards.shape[1]
    z = torch.empty((n, tables.num_heads, tables.embed_dim), dtype=torch.float32, device=boards.device)
    args = (_p(boards), 1 if boards.dtype == torch.float32 else 0, C.byref(tables.c), _p(z), n, Cc, rows, cols, _p(count), _p(sched), _stream())
    if timers is not None:
        timers[0].start()
    rc = lib().azk_nnx_embed_pool(*args)
    if timers is not None:
        timers[0].stop()
    if rc != 0:
        raise AzkError(f"azk_nnx_embed_pool failed ({rc})")
    return z


def nnx_embed_pool_leaves(src, tables, sched, timers=None):
    """azk_nnx_embed_pool over an engine's pending leaves (LeafSource): z float32 [slots, H, 512], rows [0, n_leaf) valid."""
    torch = _torch()
    assert src.rows * src.cols + 1 == tables.tokens and sched.dtype == torch.int32 and sched.numel() >= 2
    z = torch.empty((src.n_games, tables.num_heads, tables.embed_dim), dtype=torch.float32, device=sched.device)
    args = (C.byref(src), C.byref(tables.c), _p(z), _p(sched), _stream())
    if timers is not None:
        timers[0].start()
    rc = lib().azk_nnx_embed_pool_leaves(*args)
    if timers is not None:
        timers[0].stop()
    if rc != 0:
        raise AzkError(f"azk_nnx_embed_pool_leaves failed ({rc})")
    return z


def pack_linear_weight_x(w):
    """nn.Linear weight [n_out, k] -> float32 tensor in azk_nnx_gemm's fragment order (n_out padded with zero rows to a multiple of 64)."""
    torch = _torch()
    n_out, k = w.shape
    assert k % 16 == 0
    npad = (n_out + 63) // 64 * 64
    wp = torch.zeros(npad, k, dtype=torch.float32, device=w.device)
    wp[:n_out] = w.float()
    # [g, l15, c, s, l4, i] -> [g, s, c, l4, l15, i]
    return wp.view(npad // 64, 16, 4, k // 16, 4, 4).permute(0, 3, 2, 4, 1, 5).contiguous()


def nnx_gemm(a, w_packed, n_out, k, epilogue=TAIL_BF16, nbatch=1, a_batch_stride=0, bias=None, out=None, resid=None,
             a_stats=None, stats_out=None, logits=None, values=None, action_dim=0, count=None, eps=1e-5):
    """One link of the fp32 cls-row tail (azk_nnx_gemm): a float32 [m, lda] x packed float32 weights -> out float32 [m, nbatch * n_out]
    (or logits / values for the heads epilogue).  epilogue: TAIL_BF16 (= plain) / TAIL_GELU / TAIL_RESID / TAIL_HEADS."""
    torch = _torch()
    assert a.dtype == torch.float32 and a.stride(1) == 1 and w_packed.dtype == torch.float32
    d = GemmX()
    d.a_f32, d.lda, d.a_batch_stride, d.w_packed = a.data_ptr(), a.stride(0), int(a_batch_stride), w_packed.data_ptr()
    d.m, d.n_out, d.k, d.nbatch = a.shape[0], int(n_out), int(k), int(nbatch)
    d.n_valid = count.data_ptr() if count is not None else None
    d.bias = bias.data_ptr() if bias is not None else None
    d.layernorm_a, d.epilogue, d.ln_eps = (1 if a_stats is not None else 0), int(epilogue), float(eps)
    d.a_stats = a_stats.data_ptr() if a_stats is not None else None
    d.stats_out = stats_out.data_ptr() if stats_out is not None else None
    if out is not None:
        assert out.dtype == torch.float32 and out.stride(1) == 1
        d.out_f32, d.ldo = out.data_ptr(), out.stride(0)
    if resid is not None:
        assert resid.dtype == torch.float32 and resid.stride(1) == 1
        d.resid_f32, d.ldr = resid.data_ptr(), resid.stride(0)
    if logits is not None:
        d.logits_out, d.values_out, d.action_dim = logits.data_ptr(), values.data_ptr(), int(action_dim)
    rc = lib().azk_nnx_gemm(C.byref(d), _stream())
    if rc != 0:
        raise AzkError(f"azk_nnx_gemm failed ({rc})")


GEMM_H_A_SCALE, GEMM_H_W_SCALE = 16.0, 256.0      # activations x 16, weights x 256 before the fp16 (hi, lo) split (azk_nnx_gemm_h)


def pack_linear_weight_h(w):
    """nn.Linear weight [n_out, k] (float64 / float32) -> (fp16 planes in azk_nnx_gemm_h's fragment order
    [n_out/64][k/32][4][2][64][8], float32 col_sums [n_out padded] = sum_k of the RECONSTRUCTED weights)."""
    torch = _torch()
    n_out, k = w.shape
    assert k % 32 == 0
    npad = (n_out + 63) // 64 * 64
    wp = torch.zeros(npad, k, dtype=torch.float64, device=w.device)
    wp[:n_out] = w.double()
    assert float(wp.abs().max()) * GEMM_H_W_SCALE < 60000.0
    hi, lo = split_fp16(wp, GEMM_H_W_SCALE)
    # [g, l15, c, s, l4, i] -> [g, s, c, plane, l4, l15, i]
    f = lambda t: t.view(npad // 64, 16, 4, k // 32, 4, 8).permute(0, 3, 2, 4, 1, 5)
    packed = torch.stack([f(hi), f(lo)], dim=3).contiguous()
    csum = ((hi.double() + lo.double()) / GEMM_H_W_SCALE).sum(1).float().contiguous()
    return packed, csum


def nnx_gemm_h(a, w_packed, n_out, k, epilogue=TAIL_BF16, nbatch=1, a_batch_stride=0, bias=None, col_sums=None, out=None, out_f32=None,
               resid=None, a_stats=None, stats_out=None, logits=None, values=None, action_dim=0, count=None, eps=1e-5, lds=False, overflow=None):
    """One link of the fp32-accurate tail on fp16 (hi, lo) planes (azk_nnx_gemm_h).  a: a float32 tensor [m, lda] (split on the fly)
    or a (hi, lo) pair of fp16 tensors; out: a (hi, lo) pair of fp16 tensors [m, nbatch * n_out] and / or out_f32."""
    torch = _torch()
    d = GemmH()
    if isinstance(a, (tuple, list)):
        ah, al = a
        assert ah.dtype == torch.float16 and al.dtype == torch.float16 and ah.stride(1) == 1 and ah.stride(0) == al.stride(0)
        d.a_hi, d.a_lo, d.lda, m = ah.data_ptr(), al.data_ptr(), ah.stride(0), ah.shape[0]
    else:
        assert a.dtype == torch.float32 and a.stride(1) == 1
        d.a_f32, d.lda, m = a.data_ptr(), a.stride(0), a.shape[0]
    d.a_batch_stride, d.w_packed = int(a_batch_stride), w_packed.data_ptr()
    d.m, d.n_out, d.k, d.nbatch = m, int(n_out), int(k), int(nbatch)
    d.n_valid = count.data_ptr() if count is not None else None
    d.bias = bias.data_ptr() if bias is not None else None
    d.col_sums = col_sums.data_ptr() if col_sums is not None else None
    d.layernorm_a, d.epilogue, d.ln_eps = (1 if a_stats is not None else 0), int(epilogue), float(eps)
    d.a_scale, d.w_scale = GEMM_H_A_SCALE, GEMM_H_W_SCALE
    d.a_stats = a_stats.data_ptr() if a_stats is not None else None
    d.stats_out = stats_out.data_ptr() if stats_out is not None else None
    if out is not None:
        oh, ol = out
        assert oh.dtype == torch.float16 and ol.dtype == torch.float16 and oh.stride(1) == 1 and oh.stride(0) == ol.stride(0)
        d.out_hi, d.out_lo, d.ldo = oh.data_ptr(), ol.data_ptr(), oh.stride(0)
    if out_f32 is not None:
        assert out_f32.dtype == torch.float32 and out_f32.stride(1) == 1 and (out is None or out_f32.stride(0) == out[0].stride(0))
        d.out_f32, d.ldo = out_f32.data_ptr(), out_f32.stride(0)
    if resid is not None:
        assert resid.dtype == torch.float32 and resid.stride(1) == 1
        d.resid_f32, d.ldr = resid.data_ptr(), resid.stride(0)
    if logits is not None:
        d.logits_out, d.values_out, d.action_dim = logits.data_ptr(), values.data_ptr(), int(action_dim)
    if overflow is not None:
        assert overflow.dtype == torch.int32 and overflow.numel() >= 1
        d.overflow_flag = overflow.data_ptr()
    rc = (lib().azk_nnx_gemm_h_lds if lds else lib().azk_nnx_gemm_h)(C.byref(d), _stream())
    if rc != 0:
        raise AzkError(f"azk_nnx_gemm_h{'_lds' if lds else ''} failed ({rc})")
