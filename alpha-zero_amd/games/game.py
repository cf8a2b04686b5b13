"""Game - the static board-rule interface (games/game.py:4-38 + the members every caller uses),
implemented on the native rule kernels of libazk.so.

A board is the reference's own object: a caller-owned numpy float32 array [F, R, C] that make_move /
undo_move mutate IN PLACE.  Each static call ships the board to the GPU, runs the HIP kernel for that
rule through the C ABI (azk_rules_*) and writes the result back - slow per call, exact by
construction, and there is no Python re-implementation of any rule here.  Throughput comes from the
batched forms (`*_batch`, `selfplay.self_play_batch`), which keep boards resident in HBM.
"""
import numpy as np


def _t():
    import torch
    import azk
    azk.lib()
    if not torch.cuda.is_available():
        raise azk.AzkError("Game rules run on the GPU (libazk.so); no GPU is visible and there is no CPU fallback")
    return torch, azk


class Game:
    # subclasses set: engine_name, rows, cols, action_dim, state_dim, feature_dim
    engine_name = None
    rows = cols = action_dim = state_dim = feature_dim = 0

    def __init__(self):
        self.board = np.zeros((type(self).feature_dim, type(self).rows, type(self).cols), dtype=np.float32)

    # ---- helpers ---------------------------------------------------------------------------------
    @classmethod
    def _size(cls):
        return cls.rows if cls.engine_name == "gomoku" else None

    @classmethod
    def _dev(cls, board):
        torch, azk = _t()
        b = np.ascontiguousarray(board, dtype=np.float32)
        return torch, azk, torch.from_numpy(b)[None].cuda()

    # ---- static interface (games/game.py) ------------------------------------------------------------
    @classmethod
    def get_action_idx(cls, action):
        return action[1] if cls.engine_name == "connect4" else action[0] * cls.cols + action[1]

    @classmethod
    def get_valid_moves(cls, board):
        """List of (row, col) in the reference's list order (= child order of the search)."""
        torch, azk, b = cls._dev(board)
        moves, counts = azk.rules_legal_moves(cls.engine_name, b, cls._size())
        n = int(counts[0].item())
        return [(int(c) // cls.cols, int(c) % cls.cols) for c in moves[0, :n].cpu().numpy()]

    @classmethod
    def make_move(cls, board, current_player, action):
        row, col = action
        if row is None:                                  # connect4.py:57-60: full column
            print("Invalid move: column is full.")
            return current_player
        torch, azk, b = cls._dev(board)
        nxt = azk.rules_apply_move(cls.engine_name, b, torch.tensor([current_player], device=b.device),
                                   torch.tensor([row * cls.cols + col], device=b.device), cls._size())
        nxt = int(nxt[0].item())
        if nxt == current_player:
            print("Invalid move. Try again.")            # tictactoe.py:44 / gomoku.py:57
            return current_player
        board[...] = b[0].cpu().numpy()
        return nxt

    @classmethod
    def undo_move(cls, board, current_player, action):
        row, col = action
        torch, azk, b = cls._dev(board)
        azk.rules_undo_move(cls.engine_name, b, torch.tensor([current_player], device=b.device),
                            torch.tensor([row * cls.cols + col], device=b.device), cls._size())
        board[...] = b[0].cpu().numpy()

    @classmethod
    def check_winner(cls, board, player, action):
        row, col = action
        torch, azk, b = cls._dev(board)
        w = azk.rules_check_winner(cls.engine_name, b, torch.tensor([player], device=b.device),
                                   torch.tensor([row * cls.cols + col], device=b.device), cls._size())
        return int(w[0].item())

    @classmethod
    def get_canonical_board(cls, board, current_player):
        if current_player == 0:
            return board                                 # same object, no copy (gomoku.py:35-36)
        torch, azk, b = cls._dev(board)
        out = azk.rules_canonical(cls.engine_name, b, torch.tensor([current_player], device=b.device), cls._size())
        return out[0].cpu().numpy()

    @classmethod
    def display_board(cls, board):
        sym = np.full((cls.rows, cls.cols), ' ')
        sym[np.asarray(board[0]) == 1] = 'O'
        sym[np.asarray(board[1]) == 1] = 'X'
        print("\n  " + ' '.join(str(c % 10) for c in range(cls.cols)))
        for i, row in enumerate(sym):
            print(i % 10, ' '.join(row))
        print()

    @classmethod
    def mcts(cls, model, board, root, mcts_iterations, dirichlet=True):
        from ai import MCTS
        MCTS.mcts(model, board, root, cls, mcts_iterations, dirichlet)

    def _self_play_like_the_reference(self, model, mcts_iter, display):
        """The reference's own loop (gomoku.py:123-164 / tictactoe.py:99-133 / connect4.py:117-151) over the facade's
        primitives, one search per move: np.random.dirichlet is drawn once per search and np.random.choice once per sampled
        move from the GLOBAL stream, exactly where the reference draws them - a caller that seeds np.random gets the
        reference's game."""
        from ai import Node
        cls = type(self)
        gomoku = cls.engine_name == "gomoku"
        player, move_count = 0, 0
        boards, actions, pis, qs = [], [(-1, -1)], [], []
        while True:
            root = Node(None, None, player, move_count)
            cls.mcts(model, self.board, root, mcts_iter)
            pis.append(root.visit_distribution(cls))                    # utils.get_probablity_distribution_of_children
            boards.append(self.board.copy())
            qs.append(root.value / root.visit)
            sample = move_count < 8 if gomoku else True                 # gomoku.py:144; tictactoe.py:117 / connect4.py:135 (model given)
            child = root.sample_child(cls) if sample else root.max_visit_child()
            player = cls.make_move(self.board, player, child.prevAction)
            move_count += 1
            actions.append(child.prevAction)
            winner = cls.check_winner(self.board, root.currentPlayer, child.prevAction)
            if winner == -1 and move_count == cls.state_dim:
                break
            if winner != -1:
                break
        if display:
            cls.display_board(self.board)
        if gomoku:
            return boards, actions, pis, qs, winner
        return boards, pis, winner

    # ---- one self-play game (the reference's per-game entry point) -------------------------------------
    def self_play(self, model, mcts_iter, display=False):
        """One game on the engine (G = 1).  Gomoku returns the reference's 5-tuple (boards, actions, pis, qs, winner)
        (gomoku.py:164); TicTacToe / Connect4 their 3-tuple (boards, pis, winner) (tictactoe.py:133, connect4.py:151).
        The global np.random stream is consumed exactly as the reference consumes it.  For throughput call
        selfplay.self_play_batch instead - thousands of games per call."""
        from selfplay import self_play_batch
        cls = type(self)
        if model is None:
            # vanilla MCTS: the rollouts consume the global np.random stream exactly as the reference does (the MT19937
            # state travels to the device generator and back), moves are max_visit_child (tictactoe.py:117, gomoku.py:146)
            import azk
            eng = azk.Engine(cls.engine_name, 1, mcts_iter, size=cls._size())
            st = np.random.get_state()
            res = self_play_batch(cls.engine_name, None, 1, mcts_iter, size=cls._size(), engine=eng,
                                  vanilla_rng=azk.mt_state_from_numpy(st)[None])[0]
            np.random.set_state(azk.mt_state_to_numpy(eng.vanilla_get_rng()[0], st))
            eng.close()
        else:
            return self._self_play_like_the_reference(model, mcts_iter, display)
        self.board = res.boards[-1].copy()
        last = res.actions[-1]
        cls.make_move(self.board, (len(res.boards) - 1) % 2, last)
        if display:
            cls.display_board(self.board)
        if cls.engine_name == "gomoku":
            return res.as_reference_tuple()
        return res.boards, res.pis, res.winner
