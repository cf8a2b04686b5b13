"""The drop-in boundary: libazk.so loads without a GPU and exports exactly what include/azk.h declares;
compute entry points fail loudly (no CPU fallback) when no device is visible.  CPU only."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def header_symbols():
    text = open(os.path.join(ROOT, "include", "azk.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(azk_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import azk
    assert os.path.exists(azk.LIB_PATH), "run __graft_entry__.build() first"
    L = ctypes.CDLL(azk.LIB_PATH)
    declared = header_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/azk.h but not exported"
    assert sorted(azk.SYMBOLS) == declared, "azk.SYMBOLS (python binding) and include/azk.h disagree"
    L.azk_abi_version.restype = ctypes.c_int32
    assert L.azk_abi_version() == azk.ABI_VERSION == 4


def test_no_torch_types_in_the_abi():
    import re
    text = open(os.path.join(ROOT, "include", "azk.h")).read()
    code = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)             # declarations only: comments may mention torch tensors as an example
    code = re.sub(r"//[^\n]*", " ", code)
    for banned in ("torch", "at::", "c10", "std::", "Tensor", "template", "class "):
        assert banned not in code, f"{banned!r} in the C ABI declarations"
    assert "#include <stdint.h>" in text and 'extern "C"' in text


def test_engine_creation_fails_loudly_without_gpu():
    import torch
    import azk
    if torch.cuda.is_available():
        pytest.skip("GPU visible")
    with pytest.raises(azk.AzkError):
        azk.Engine("gomoku", 2, 8, size=7)
    # the raw ABI also reports an error (no silent CPU path)
    L = azk.lib()
    cfg = azk.Config()
    cfg.game, cfg.rows, cfg.cols, cfg.n_games, cfg.max_sims = 2, 7, 7, 2, 8
    h = ctypes.c_void_p()
    rc = L.azk_create(ctypes.byref(cfg), ctypes.byref(h))
    assert rc < 0 and not h.value
    assert b"no HIP device" in L.azk_last_error(None) or b"HIP" in L.azk_last_error(None)
    # bad arguments are rejected before touching the device
    cfg.game = 9
    assert L.azk_create(ctypes.byref(cfg), ctypes.byref(h)) == -1


def test_facade_import_surface_matches_reference():
    """`from ai import Node, MCTS`, `from games import TicTacToe, Connect4, Gomoku` (ai/__init__.py, games/__init__.py)."""
    from ai import MCTS, Node
    from games import Connect4, Gomoku, TicTacToe
    assert (TicTacToe.rows, TicTacToe.cols, TicTacToe.action_dim, TicTacToe.state_dim) == (3, 3, 9, 9)
    assert (Connect4.rows, Connect4.cols, Connect4.action_dim, Connect4.state_dim) == (6, 7, 7, 42)
    assert (Gomoku.rows, Gomoku.cols, Gomoku.action_dim, Gomoku.feature_dim) == (7, 7, 49, 2)
    assert Gomoku.get_action_idx((2, 3)) == 17 and Connect4.get_action_idx((5, 3)) == 3
    n = Node(None, None, 0, 0)
    assert (n.visit, n.value, n.children, n.prior) == (0, 0, [], 0.0)
    assert MCTS.mcts_count >= 0 and isinstance(MCTS.cache, dict)
    for meth in ("display_board", "get_action_idx", "get_valid_moves", "make_move", "undo_move", "check_winner", "mcts"):
        assert hasattr(Gomoku, meth)


def test_binding_refuses_a_library_of_another_abi_version(monkeypatch):
    """azk.lib() compares azk_abi_version() with the version its structure layouts were written for (a stale libazk.so would
    otherwise overrun buffers silently: e.g. azk_emit_finished's game_base went from int32* to int64* between versions 1 and 2)."""
    import azk
    monkeypatch.setattr(azk, "_LIB", None)
    monkeypatch.setattr(azk, "ABI_VERSION", 999)
    with pytest.raises(azk.AzkError, match="ABI version"):
        azk.lib()
    monkeypatch.setattr(azk, "ABI_VERSION", 4)
    monkeypatch.setattr(azk, "_LIB", None)
    assert azk.lib().azk_abi_version() == 4
