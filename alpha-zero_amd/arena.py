"""Arena on the batched engine: test.compete / test.compare (test.py:60-140).

compete: two agents alternate by side to move (model1 plays player 0); every move is a fresh-root search with the
mover's model and iteration count (Dirichlet noise on, as Game.mcts's default); the move is sampled from the visit
distribution while move_count < 20 when `sampling`, else the first most-visited child; returns the winner.
compare: `iterations` games, the best model plays first in the first half and second in the second half; draws score
0.5 each; optional early stopping at 55 %.  All games of a half run as one lock-step batch; the early-stopping rule
is then applied to the outcomes in game order, which yields exactly what the sequential loop would return.
"""
import numpy as np

from selfplay import self_play_batch


def _game_name(game, size):
    """`game`: an engine name ("gomoku", ...) or one of the facade's Game classes (games.Gomoku, ...), as test.py passes it."""
    if isinstance(game, str):
        return game, size
    return game.engine_name, (game._size() if size is None else size)


def compete_batch(game, model1, model2, n_games, model1_mcts_iter=50, model2_mcts_iter=50, sampling=False, size=None,
                  seed=0, first_global_game=0, device=0, leaf_dtype="float32", noise_fn=None, uniform_fn=None, cache_entries=0):
    """n_games independent test.compete games at once -> (winners int array in {0, 1, -1}, final boards).

    cache_entries > 0 turns on the per-game eval cache, which - like the reference's process-global MCTS.cache (mcts.py:7,
    38-44: keyed by the position alone) - is shared by BOTH models: whichever model evaluated a position first also answers
    for the other.  That reproduces the reference's compete games exactly as long as no entry is evicted (direct-mapped
    table: size it well above the positions a game visits); with 0 every model evaluates its own positions."""
    game, size = _game_name(game, size)
    res = self_play_batch(game, (model1, model2), n_games, (model1_mcts_iter, model2_mcts_iter), size=size, seed=seed,
                          first_global_game=first_global_game, device=device, leaf_dtype=leaf_dtype,
                          noise_fn=noise_fn, uniform_fn=uniform_fn, sample_until=20 if sampling else 0, cache_entries=cache_entries)
    winners = np.array([r.winner for r in res], np.int64)
    return winners, res


def compete(Game, model1, model2, model1_mcts_iter=50, model2_mcts_iter=50, sampling=False, display=False):
    """test.compete (test.py:60-105) for ONE game, as the reference runs it: the facade's primitives, one search per move,
    np.random.dirichlet / np.random.choice / np.random.randint from the global stream where the reference draws them, and
    one eval cache shared by both models (MCTS.cache).  Returns (winner, final board).  Use compete_batch for throughput."""
    from ai import Node
    game = Game()
    player, move_count = 0, 0
    while True:
        if display:
            Game.display_board(game.board)
        root = Node(None, None, player, move_count)
        if player == 0:
            Game.mcts(model1, game.board, root, model1_mcts_iter)
        else:
            Game.mcts(model2, game.board, root, model2_mcts_iter)
        child = root.sample_child(Game) if (sampling and move_count < 20) else root.max_visit_child()
        player = Game.make_move(game.board, player, child.prevAction)
        move_count += 1
        winner = Game.check_winner(game.board, root.currentPlayer, child.prevAction)
        if winner != -1:
            return winner, game.board
        if move_count == Game.state_dim:
            return -1, game.board


def score_like_reference(winners_first_half, winners_second_half, iterations, early_stopping):
    """test.compare's bookkeeping (test.py:107-140) over outcomes given in game order.
    Returns 1 / 0 on an early stop (accepted / rejected) else the contender's win rate."""
    win_count = [0, 0, 0]                                      # best model's, contender's, draws
    outcomes = list(winners_first_half) + list(winners_second_half)
    for i, winner in enumerate(outcomes):
        first = i < iterations // 2
        if winner == 0:
            win_count[0 if first else 1] += 1
        elif winner == 1:
            win_count[1 if first else 0] += 1
        else:
            win_count[0] += 0.5
            win_count[1] += 0.5
            win_count[2] += 1
        if early_stopping:
            if win_count[1] >= int(iterations * 0.55):
                return 1
            if win_count[1] + (iterations - (i + 1)) < int(iterations * 0.55):
                return 0
    return win_count[1] / iterations


def compare_sequential(Game, best_model, contender_model, best_model_mcts_iter, contender_model_mcts_iter, iterations, sampling,
                       early_stopping, winners_out=None):
    """test.compare exactly as the reference runs it (test.py:107-140): the games one after the other through `compete` (global
    np.random stream, MCTS.cache kept across the games and shared by both models), the models swap sides at half time, the
    early-stopping rule is applied after every game (so it also decides how many games are played).  Returns 1 / 0 on an
    early stop, else the contender's win rate.  `winners_out` (a list) receives each game's winner."""
    win_count = [0, 0, 0]                                      # best model's, contender's, draws
    for i in range(iterations):
        first = i < iterations // 2
        winner, _ = compete(Game, best_model if first else contender_model, contender_model if first else best_model,
                            best_model_mcts_iter if first else contender_model_mcts_iter,
                            contender_model_mcts_iter if first else best_model_mcts_iter, sampling=sampling)
        if winners_out is not None:
            winners_out.append(int(winner))
        if winner == 0:
            win_count[0 if first else 1] += 1
        elif winner == 1:
            win_count[1 if first else 0] += 1
        else:
            win_count[0] += 0.5
            win_count[1] += 0.5
            win_count[2] += 1
        if early_stopping:
            if win_count[1] >= int(iterations * 0.55):
                return 1
            if win_count[1] + (iterations - (i + 1)) < int(iterations * 0.55):
                return 0
    return win_count[1] / iterations


def compare(game, best_model, contender_model, best_model_mcts_iter, contender_model_mcts_iter, iterations, sampling,
            early_stopping, size=None, seed=0, device=0, leaf_dtype="float32"):
    """test.compare(Game, best_model, contender_model, best_iter, contender_iter, iterations, sampling, early_stopping)
    (test.py:107-140) with all games of a half played at once; a model of None plays vanilla MCTS (main.py:76).
    (compare_sequential is the game-by-game form that reproduces the reference's RNG stream and cache history.)"""
    game, size = _game_name(game, size)
    half = iterations // 2
    w1, _ = compete_batch(game, best_model, contender_model, half, best_model_mcts_iter, contender_model_mcts_iter,
                          sampling, size, seed, 0, device, leaf_dtype) if half else (np.zeros(0, np.int64), None)
    w2, _ = compete_batch(game, contender_model, best_model, iterations - half, contender_model_mcts_iter,
                          best_model_mcts_iter, sampling, size, seed, half, device, leaf_dtype)
    return score_like_reference(w1, w2, iterations, early_stopping)
