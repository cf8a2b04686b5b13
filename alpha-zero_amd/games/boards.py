"""The three games of the reference as thin descriptors over the native rule kernels.

Geometry, plane count and run length are the only per-game facts kept on the host; the rules
themselves (legal-move order, apply, undo, k-in-a-row) are HIP kernels (csrc/azk_device.h).
"""
from .game import Game


class TicTacToe(Game):
    # games/tictactoe.py:10-12,17: 3x3, 3 planes (player 0, player 1, side to move), three in a row
    engine_name = "tictactoe"
    rows, cols = 3, 3
    action_dim = 9
    state_dim = 9
    feature_dim = 3


class Connect4(Game):
    # games/connect4.py:8-10,14: 6x7, action = column, four in a row
    engine_name = "connect4"
    rows, cols = 6, 7
    action_dim = 7
    state_dim = 42
    feature_dim = 3

    @staticmethod
    def get_drop_row(board, col):
        for row in range(Connect4.rows - 1, -1, -1):
            if board[0, row, col] == 0 and board[1, row, col] == 0:
                return row
        return None


class Gomoku(Game):
    # games/gomoku.py:10-14: ships 7x7; any square size by overriding rows/cols/action_dim/state_dim
    #  or with Gomoku.set_size(15).  Two planes, five in a row (overlines win).
    engine_name = "gomoku"
    rows, cols = 7, 7
    action_dim = 49
    state_dim = 49
    feature_dim = 2

    @classmethod
    def set_size(cls, n):
        cls.rows = cls.cols = n
        cls.action_dim = cls.state_dim = n * n
