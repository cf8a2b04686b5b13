"""Node - host-side VIEW of one tree edge, with the reference's field names (ai/node.py:21-40).

The tree itself lives in the engine's structure-of-arrays arena in HBM; after a search the root and
its children are materialised into Node objects because that is what callers read
(train.py / test.py: root.children, child.visit, child.prevAction, root.value / root.visit,
root.max_visit_child(), root.sample_child(Game), child.to_string(Game)).
"""
import numpy as np


class Node:
    def __init__(self, parent, prevAction, currentPlayer, move_count, prior=0.0):
        self.parent = parent
        self.visit = 0
        self.value = 0
        self.ucb = np.inf
        self.prior = prior
        self.move_count = move_count
        self.prevAction = prevAction
        self.currentPlayer = currentPlayer
        self.children = []

    # -- the three read-side methods callers use ------------------------------------------------------
    def max_visit_child(self):
        """First child with the most visits (Python max keeps the first maximum; node.py:76-81)."""
        best = None
        for ch in self.children:
            if best is None or ch.visit > best.visit:
                best = ch
        return best

    def visit_distribution(self, Game):
        """utils.get_probablity_distribution_of_children (utils.py:46-55)."""
        counts = np.zeros(Game.action_dim)
        for ch in self.children:
            counts[Game.get_action_idx(ch.prevAction)] = ch.visit
        return counts / np.sum(counts)

    def sample_child(self, Game):
        """np.random.choice over children by visit share (node.py:83-93); consumes one global uniform."""
        p = self.visit_distribution(Game)
        slots = [None] * Game.action_dim
        for ch in self.children:
            slots[Game.get_action_idx(ch.prevAction)] = ch
        return np.random.choice(slots, p=p)

    def to_string(self, Game):
        p = self.parent.visit_distribution(Game)
        return (f"Node: {self.prevAction}, Value: {self.value}, Visit: {self.visit}, "
                f"P(Visit): {p[Game.get_action_idx(self.prevAction)]}, UCB: {self.ucb}")

    # -- write-side methods: the engine owns select / expand / backup ----------------------------------
    def select(self, mode):
        raise NotImplementedError("selection runs inside the HIP engine (azk_step_select); a Node is a read-only view")

    def expand(self, valid_moves, policy_distribution, Game):
        raise NotImplementedError("expansion runs inside the HIP engine (azk_step_expand_backup)")

    def backup(self, trace, value, board, Game):
        raise NotImplementedError("backup runs inside the HIP engine (azk_step_expand_backup)")
