"""`from games import TicTacToe, Connect4, Gomoku` - same import surface as the reference (games/__init__.py:1-3)."""
from .game import Game
from .boards import TicTacToe, Connect4, Gomoku

__all__ = ["Game", "TicTacToe", "Connect4", "Gomoku"]
