"""`from ai import Node, MCTS` - same import surface as the reference's ai package (ai/__init__.py:1-2)."""
from .node import Node
from .mcts import MCTS

__all__ = ["Node", "MCTS"]
