/*
 * azk.h - C ABI of libazk.so: the MI355X-native batched self-play engine (HIP, gfx950).
 *
 * This is the drop-in boundary for the reference's self-play hot path
 *     train.collect_data -> <Game>.self_play -> ai.mcts.MCTS.mcts -> Game statics + model
 * (reference files cited per entry point as file:line, relative to the reference root).
 * The reference has no native layer: everything below replaces interpreted Python.  The ABI is
 * plain C: no C++ / torch types, raw pointers + sizes, int32 status returns (0 = OK, <0 = error,
 * text via azk_last_error).  Pointers named *_dev are device (HBM) pointers - e.g. a torch
 * tensor's data_ptr() - valid until the stream reaches the call; *_host are host pointers.
 * `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream); every
 * kernel is enqueued on it and nothing here synchronises unless stated, so calls are capturable
 * in a hipGraph.  One host thread drives one engine; one engine per GPU / process.
 *
 * Data model (DESIGN.md "HBM layout"): G concurrent games; per game a structure-of-arrays tree
 * arena (the reference's Node fields, ai/node.py:21-40, one row per child edge): N int32, W float64,
 * P float32 (+ a float64 root-prior row after Dirichlet mixing), cell int16, first_child int32,
 * n_children int16.  One 64-lane wavefront owns one game; child blocks are contiguous, so a PUCT
 * scan is a coalesced read of the N/W/P columns.
 */
#ifndef AZK_H
#define AZK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: azk_emit_finished's game_base_dev became int64*, azk_leaf_source gained cache_stamp, azk_config gained cache_shared /
 * leaves_per_step (round 2); callers compare azk_abi_version() with the header they were built against */
#define AZK_ABI_VERSION 4

/* games (games/tictactoe.py, games/connect4.py, games/gomoku.py) */
#define AZK_TICTACTOE 0
#define AZK_CONNECT4 1
#define AZK_GOMOKU 2

/* leaf batch element type handed to the evaluator */
#define AZK_LEAF_F32 0
#define AZK_LEAF_BF16 1

/* status codes */
#define AZK_OK 0
#define AZK_ERR_ARG (-1)
#define AZK_ERR_HIP (-2)
#define AZK_ERR_ARENA_FULL (-3)
#define AZK_ERR_STATE (-4)

typedef struct azk_engine azk_engine;

typedef struct {
    int32_t game;          /* AZK_* */
    int32_t rows, cols;    /* Gomoku only (reference ships 7x7, gomoku.py:10; 15x15 by attribute override); 1..400 cells, at most 30 columns */
    int32_t n_games;       /* G: games resident on this GPU */
    int32_t max_sims;      /* largest mcts_iterations a search will run (arena sizing) */
    int32_t leaf_dtype;    /* AZK_LEAF_F32 | AZK_LEAF_BF16 */
    int32_t device;        /* HIP device ordinal */
    int32_t arena_nodes;   /* nodes per game; 0 = worst case 1 + max_sims * max_children */
    int32_t cache_entries; /* eval cache (MCTS.cache, ai/mcts.py:7,38-51): entries per game, power of two, 0 = off.  Exact keys
                            * (the whole canonical position), so a hit returns precisely what the evaluator returned. */
    int32_t cache_shared;  /* 0: one table per game (single writer; exact MCTS.matched bookkeeping of a one-game run).
                            * 1: ONE table of n_games * cache_entries entries shared by every game of the engine - the reference's
                            * MCTS.cache is process-global (mcts.py:7): positions reached by several games (openings) are evaluated
                            * once.  Entries are written at expansion under a per-entry claim word and become readable at the next
                            * tree launch; a hit is copied into the game's own buffers at once.  Results are identical in both
                            * modes (the cache is transparent); only the hit COUNT of the shared mode depends on timing. */
    int32_t leaves_per_step; /* 0 / 1: the reference's sequential search, one leaf in flight per game (parity mode, the default).
                              * K > 1 (OPT-IN, changes search results): virtual-loss expansion - K leaves in flight per game; a
                              * selection leaves a visit and a lost game on its path until the leaf's value is backed up, so the
                              * next selections of the same game go elsewhere; the evaluator batch holds up to n_games * K boards.
                              * Drive it with azk_begin_search_budget (n_sims completed simulations per game). */
    int32_t reserved[5];
} azk_config;

/* device-side work counters; sums over all games since the last azk_reset_counters */
typedef struct {
    int64_t sims;             /* iterations of ai/mcts.py:16 */
    int64_t edges_scanned;    /* children read by PUCT scans (utils.py:29-44) */
    int64_t trace_nodes;      /* nodes updated by backups (node.py:62-74) */
    int64_t edges_created;    /* children appended by expansions (node.py:50-59) */
    int64_t leaves_evaluated; /* leaf boards handed to the evaluator (mcts.py:46) */
    int64_t terminal_sims;    /* simulations that ended in mcts.py:25-32 */
    int64_t moves_played;
    int64_t cache_hits;       /* MCTS.matched (mcts.py:9,44): leaves served by the eval cache; leaves_evaluated counts the misses */
    int64_t reserved[8];
} azk_counters;

int32_t azk_abi_version(void);
/* e may be NULL: returns the message of the last failed azk_create on this thread */
const char *azk_last_error(const azk_engine *e);

/* ---- engine life cycle ---------------------------------------------------------------------- */
int32_t azk_create(const azk_config *cfg, azk_engine **out);
void azk_destroy(azk_engine *e);
/* geometry the caller needs to size buffers: planes F, rows, cols, action_dim A, state_dim */
int32_t azk_geometry(const azk_engine *e, int32_t *planes, int32_t *rows, int32_t *cols,
                     int32_t *action_dim, int32_t *state_dim);

/* Game(): empty boards, player 0 to move, move_count 0 (gomoku.py:16-17,125-126) for games [first, first+count) */
int32_t azk_reset_games(azk_engine *e, int32_t first, int32_t count, void *stream);
/* load caller positions: cells int8 [count][rows*cols] (0 empty, 1 player-0, 2 player-1), side to move, plies played.
 * This is how MCTS.mcts(model, board, root, ...) (ai/mcts.py:11) receives the caller's board and root. */
int32_t azk_set_positions(azk_engine *e, int32_t first, int32_t count, const int8_t *cells_host,
                          const int32_t *to_move_host, const int32_t *move_count_host, void *stream);

/* ---- one search = Node(None, None, player, move_count) + MCTS.mcts(...) (gomoku.py:134-136) ---- */
/* Fresh root for every game.  noise_dev: float64 [G][A] Dirichlet draws (utils.py:24) or NULL for
 * dirichlet=False.  The pointer is read by later steps: keep it alive until the search ends. */
int32_t azk_begin_search(azk_engine *e, const double *noise_dev, void *stream);

/* Budget stepping: like azk_begin_search, plus a simulation budget per game.  After it, every azk_step / azk_step_tree lets a game
 * run on inside the launch while its simulations need no evaluator (terminal leaves, leaves served by the eval cache) and stop at
 * the first leaf that does, at n_sims simulations in all, or after max_sims_per_launch in one launch.  A game's simulations stay
 * strictly sequential, so trees and games are bit-identical to one-simulation-per-call stepping; the caller steps until
 * azk_search_unfinished reports 0 (games still owing simulations or an evaluation), then calls azk_step_expand_backup once. */
int32_t azk_begin_search_budget(azk_engine *e, const double *noise_dev, int32_t n_sims, int32_t max_sims_per_launch, void *stream);
int32_t azk_search_unfinished(azk_engine *e, int32_t *count_host, void *stream);

/* ---- asynchronous self-play: a game moves as soon as ITS search is done (games/gomoku.py:132-162 runs one game at a time; in
 * the batched engine that means no game waits for the slowest search of its batch).  Drive it as
 *     azk_async_begin;  loop { azk_async_step(logits, values);  evaluator over the pending leaves;  every few steps: azk_async_drain }
 * azk_async_step = one budget-stepped tree launch (as azk_step_tree after azk_begin_search_budget: a game's simulations stay strictly
 * sequential, so every tree is bit-identical to one-simulation-per-call stepping) followed by the per-game move kernel: a game whose
 * n_sims simulations are complete gets root statistics, pi recorded, the move chosen (sampled while move_count < sample_until_move with
 * the uniform of (seed, global game, slot move counter), else the first most-visited child), make_move / check_winner / draw, a record
 * into the caller's ring, and - unless it ended - its next search at once (fresh root, the Dirichlet row of its next move key).
 * azk_async_drain handles the games that ended: (state, pi, z) emission into the caller's replay ring (as azk_emit_finished; pass
 * states_dev = NULL to skip), statistics, and - with recycle - Game() + first search of the next game in the slot.
 * The random keys are those of the lock-step drivers (azk_gen_noise with move_index = the slot's move counter), so a slot plays the
 * same games move for move.  Both calls only enqueue kernels (capturable).
 *   stats_dev  int64 [16], zeroed by azk_async_begin: [0] games finished, [1] their plies, [2] wins of player 0, [3] of player 1,
 *              [4] draws, [5] moves played, [6] records written (ring cursor), [7] searches begun
 *   record ring (optional, record_capacity entries; entry r % capacity): rec_meta int32 [cap][4] = slot, the slot's move counter,
 *              chosen cell, winner (-2 running, -1 draw, 0 / 1); rec_q float64 [cap] = root.value / root.visit; rec_pi float64 [cap][A] */
typedef struct azk_async_config {
    int32_t n_sims, max_sims_per_launch, sample_until_move, dirichlet, recycle;
    int32_t young_launch_us;   /* > 0: a game starts another simulation inside a launch only while the launch is younger than this
                                * (a launch lasts as long as its slowest wave); 0: up to max_sims_per_launch whatever the time */
    uint64_t seed;
    int64_t first_global_game;
    double alpha;
    int64_t *stats_dev;
    int64_t record_capacity;
    int32_t *rec_meta_dev;
    double *rec_q_dev, *rec_pi_dev;
} azk_async_config;
int32_t azk_async_begin(azk_engine *e, const azk_async_config *cfg, void *stream);
/* phases: bit 0 = the tree launch, bit 1 = the move kernel (3 = both; separately for per-kernel timing) */
int32_t azk_async_step(azk_engine *e, const float *logits_dev, const float *values_dev, int32_t phases, void *stream);
/* change the simulation budget of every later launch (it lives in device memory, so captured step graphs pick it up); synchronises */
int32_t azk_async_set_budget(azk_engine *e, int32_t n_sims, int32_t max_sims_per_launch, void *stream);
int32_t azk_async_drain(azk_engine *e, float *states_dev, double *pis_dev, float *zs_dev, int64_t capacity, int64_t *cursor_dev, void *stream);

/* One simulation per active game (ai/mcts.py:16-60), split around the evaluator:
 *   azk_step_select   - mcts.py:18-37: PUCT walk (node.py:42-47, utils.py:29-44), make_move along the
 *                       path, terminal test + immediate backup, get_valid_moves, canonical board.
 *                       Non-terminal leaves are compacted (ascending game index) into
 *                       leaf_boards_dev [n_leaf][F][R][C]; *n_leaf_dev (int32) gets the count.
 *   azk_step_expand_backup - mcts.py:46-60: float32 softmax without max subtraction, root noise
 *                       mixing (utils.py:12-27), Node.expand (node.py:50-59), Node.backup (node.py:62-74)
 *                       with value = -v.  logits_dev float32 [n_leaf][A], values_dev float32 [n_leaf],
 *                       rows in the slot order azk_step_select produced.
 * azk_step fuses "expand+backup of the previous step's leaves" with "select of the next" in one launch
 * (pass logits_dev = NULL on the first step of a search). */
int32_t azk_step_select(azk_engine *e, void *leaf_boards_dev, int32_t *n_leaf_dev, void *stream);
int32_t azk_step_expand_backup(azk_engine *e, const float *logits_dev, const float *values_dev, void *stream);
int32_t azk_step(azk_engine *e, const float *logits_dev, const float *values_dev,
                 void *leaf_boards_dev, int32_t *n_leaf_dev, void *stream);

/* The two launches azk_step is made of, separately (per-kernel timing; also lets the caller overlap):
 *   azk_step_tree   - k_tree: expand+backup of the pending leaves (logits_dev != NULL) then PUCT select
 *   azk_step_gather - k_gather: leaf compaction + canonical boards into the evaluator batch */
int32_t azk_step_tree(azk_engine *e, const float *logits_dev, const float *values_dev, void *stream);
int32_t azk_step_gather(azk_engine *e, void *leaf_boards_dev, int32_t *n_leaf_dev, void *stream);

/* Continuous self-play (the batched form of train.collect_data's `for iter in range(iterations): game = Game()`,
 * train.py:63-65): every finished game's slot restarts from an empty board.  stats_dev int64[8] accumulates
 * [0] games finished, [1] their total plies, [2] wins of player 0, [3] wins of player 1, [4] draws. */
int32_t azk_recycle_finished(azk_engine *e, int64_t *stats_dev, void *stream);

/* (state, pi, z) emission for every game that has just finished (train.save_data_to_buffer, train.py:30-49, with the D4
 * augmentation of rotate_data / flip_data, train.py:8-27).  Call after azk_advance and before azk_recycle_finished.
 * Position i of the game (side to move i & 1): state = canonical board float32 [F][R][C], pi float64 [A] as recorded by
 * azk_advance, z = +1 / -1 / 0; positions 0 and 1 once, the rest 8 times in the reference's order (rot0, lr, tb, rot90,
 * lr, tb, rot180, rot270).  The engine appends at *cursor_dev (uint64, monotonically increasing) and writes tuple t to
 * slot t % capacity of the caller's ring buffers - the device-resident form of ReplayBuffer's deque(maxlen)
 * (replay_buffer.py:7-13).  game_base_dev (optional, int64 [G]) receives each emitted game's first tuple index, -1 for
 * the others.  Square boards with one action per cell only (Gomoku, TicTacToe); others return AZK_ERR_ARG. */
int32_t azk_emit_finished(azk_engine *e, float *states_dev, double *pis_dev, float *zs_dev, int64_t capacity,
                          int64_t *cursor_dev, int64_t *game_base_dev, void *stream);

/* MCTS.cache.clear() (main.py:55-57): must be called whenever the evaluator's weights change. */
int32_t azk_clear_cache(azk_engine *e, void *stream);

/* Root statistics after a search, for all G games (device outputs, any may be NULL):
 *   pi_dev float64 [G][A]   utils.get_probablity_distribution_of_children (utils.py:46-55)
 *   q_dev  float64 [G]      root.value / root.visit (gomoku.py:140)
 *   root_visit_dev int32 [G] */
int32_t azk_root_stats(azk_engine *e, double *pi_dev, double *q_dev, int32_t *root_visit_dev, void *stream);

/* Root children of ONE game, in the reference's list order (what callers read from root.children:
 * child.prevAction / visit / value / prior, test.py:46-47, node.py:76-97).  Host outputs of
 * capacity `cap` each; returns the child count (>= 0) or an error.  Synchronises the stream. */
int32_t azk_root_children(azk_engine *e, int32_t game, int32_t cap, int32_t *cells_host,
                          int32_t *visits_host, double *values_host, double *priors_host, void *stream);
/* Whole tree of one game, DFS pre-order with children in list order (for digests / Node views).
 * Returns the node count (may exceed cap: only the first cap rows are written).  Synchronises. */
int32_t azk_export_tree(azk_engine *e, int32_t game, int32_t cap, int32_t *depth_host, int32_t *cell_host,
                        int32_t *visit_host, double *value_host, double *prior_host, void *stream);

/* Move selection + state advance for all active games (gomoku.py:143-162):
 *   game g samples ~ visits when move_count[g] < sample_until_move (Node.sample_child, node.py:83-93:
 *   legacy np.random.choice == searchsorted(cumsum(pi)/sum, u, 'right') with u = uniforms_dev[g]),
 *   else takes the first child with the most visits (Node.max_visit_child, node.py:76-81);
 *   then make_move, check_winner for the mover, draw when move_count == state_dim.
 * Outputs (device, may be NULL): chosen_cell_dev int32 [G] (r*cols+c, -1 for finished games),
 * winner_dev int32 [G] (-2 still running, -1 draw, 0/1 winner), done_dev int32 [G]. */
int32_t azk_advance(azk_engine *e, const double *uniforms_dev, int32_t sample_until_move,
                    int32_t *chosen_cell_dev, int32_t *winner_dev, int32_t *done_dev, void *stream);

/* current boards as int8 cells [G][rows*cols], side to move [G], plies [G] (host outputs; synchronises) */
int32_t azk_get_positions(azk_engine *e, int8_t *cells_host, int32_t *to_move_host, int32_t *move_count_host, void *stream);

int32_t azk_get_counters(azk_engine *e, azk_counters *out, void *stream);   /* synchronises */
int32_t azk_reset_counters(azk_engine *e, void *stream);
/* debug only: per-phase shader-clock sums of k_tree (collected when AZK_TREE_ABLATE has bit 16): expand, board+root
 * load, walk, terminal test, valid moves, writes (cycles), then summed depth and sample count.  Synchronises. */
int32_t azk_debug_stamps(azk_engine *e, int64_t *out8_host);
/* debug only: the raw [n_games][8] stamp records (AZK_TREE_ABLATE bit 8192: one record per wave of the LAST k_tree launch - cycles of
 * expansion, walk, terminal test, legal moves, leaf writes + cache probe; depth; kind + 1 | legal moves << 8 | children created << 20;
 * the wave's whole life).  Synchronises. */
int32_t azk_debug_stamps_raw(azk_engine *e, int64_t *out_host, int32_t n_games);
/* sticky device-side error word (arena overflow etc.); synchronises; returns AZK_OK or the error */
int32_t azk_check_device_error(azk_engine *e, void *stream);

/* Dirichlet(alpha) rows and uniforms from a counter-based generator keyed by
 * (seed, global game index, move index) - results do not depend on how games are sharded over GPUs.
 * noise_dev float64 [count][A]; uniforms_dev float64 [count] (either may be NULL). */
int32_t azk_gen_noise(azk_engine *e, uint64_t seed, int64_t first_global_game, int32_t move_index, double alpha,
                      double *noise_dev, double *uniforms_dev, void *stream);

/* ---- stateless board-rule kernels over caller boards (float32 [n][F][R][C], the reference's layout) ----
 * game/rows/cols as in azk_config.  These replace the Game statics (games/game.py:4-38):
 *   azk_rules_legal_moves : get_valid_moves - cells in the reference's LIST ORDER (tictactoe.py:82-83,
 *                           connect4.py:44-53, gomoku.py:93-106 incl. CPython set order), moves_dev int16 [n][R*C],
 *                           counts_dev int32 [n]
 *   azk_rules_legal_mask  : same set as a uint8 mask over ACTIONS [n][A]
 *   azk_rules_apply_move  : make_move (returns next player; occupied cell => unchanged player, tictactoe.py:37-45,
 *                           gomoku.py:51-58; connect4.py:56-63 never checks)
 *   azk_rules_undo_move   : undo_move (tictactoe.py:48-51, connect4.py:67-70, gomoku.py:61-63)
 *   azk_rules_check_winner: check_winner (k-in-a-row through the cell; player or -1)
 *   azk_rules_canonical   : get_canonical_board (gomoku.py:34-40; 3-plane: mcts.py:126-137) */
int32_t azk_rules_legal_moves(int32_t game, int32_t rows, int32_t cols, const float *boards_dev, int32_t n,
                              int16_t *moves_dev, int32_t *counts_dev, void *stream);
int32_t azk_rules_legal_mask(int32_t game, int32_t rows, int32_t cols, const float *boards_dev, int32_t n,
                             uint8_t *mask_dev, void *stream);
int32_t azk_rules_apply_move(int32_t game, int32_t rows, int32_t cols, float *boards_dev, int32_t n,
                             const int32_t *players_dev, const int32_t *cells_dev, int32_t *next_player_dev, void *stream);
int32_t azk_rules_undo_move(int32_t game, int32_t rows, int32_t cols, float *boards_dev, int32_t n,
                            const int32_t *current_players_dev, const int32_t *cells_dev, void *stream);
int32_t azk_rules_check_winner(int32_t game, int32_t rows, int32_t cols, const float *boards_dev, int32_t n,
                               const int32_t *players_dev, const int32_t *cells_dev, int32_t *winners_dev, void *stream);
int32_t azk_rules_canonical(int32_t game, int32_t rows, int32_t cols, const float *boards_dev, int32_t n,
                            const int32_t *players_dev, float *out_dev, void *stream);

/* ---- policy-value network: token embedding on the matrix cores (ai/nn.py:5-36) ------------------------------
 * tokens[b,0,:] = cls + pos[0];  tokens[b,1+j,:] = Conv2d(C->D, k x k, stride 1, pad k/2)(board)[:, j] + pos[1+j]
 * as an im2col GEMM on the matrix cores (fp32 accumulate), epilogue fused: + bias + positional embedding,
 * and optionally the first block's LayerNorm (nn.py:53, eps 1e-5) so the block's Q/K/V GEMM reads xhat directly.
 *   boards_dev        [n][C][R][Cc] bf16 (boards_are_f32 = 0) or float32 (1), values 0/1 (the engine's leaf batch)
 *   wt_bf16_dev       [D][kp] bf16: conv weight reshaped [D][C*k*k], zero padded to kp (multiple of 32)
 *   cpos_dev          [T][D] float32: row 0 = cls + pos[0], row 1+j = conv bias + pos[1+j]   (T = R*Cc + 1)
 *   ln_w_dev/ln_b_dev [D] float32 (required when xhat_out is given)
 *   x_out / xhat_out  [n][T][D] bf16, either may be NULL
 * Supported: embed_dim in {128, 256, 512}, kp in {32, 64, 96}, C*R*Cc <= 1984; anything else returns AZK_ERR_ARG
 * (the caller keeps a generic path).  v_mfma_f32_16x16x32_bf16; one wavefront per 16-token x D tile. */
int32_t azk_nn_patch_embed(const void *boards_dev, int32_t boards_are_f32, const void *wt_bf16_dev,
                           const float *cpos_dev, const float *ln_w_dev, const float *ln_b_dev,
                           void *x_out_bf16_dev, void *xhat_out_bf16_dev, int32_t n, int32_t channels,
                           int32_t rows, int32_t cols, int32_t ksize, int32_t kp, int32_t embed_dim,
                           float ln_eps, void *stream);

/* cls-row attention of the last block, folded (ai/nn.py:52-56 restricted to the row nn.py:80 reads): streams
 * xhat = LayerNorm1(tokens) [n][T][D] bf16 once and writes z[b][h][:] = sum_t softmax_t(xhat_t . m[b][h] + c[b][h])[t] * xhat_t
 * ([n][H][D] bf16) with an online softmax, so K and V are never materialised.  m_dev float32 [n or 1][H][D] =
 * scale * Wk_h^T q_h, c_dev float32 [n or 1][H] = scale * q_h . bk_h (per_board_m = 0: one shared row set, the depth-1
 * case where q is input independent).  Supported (embed_dim, heads): (512,8) (512,4) (256,8) (256,4) (128,4) (128,8). */
int32_t azk_nn_cls_attention(const void *xhat_bf16_dev, const float *m_dev, const float *c_dev, int32_t per_board_m,
                             void *z_out_bf16_dev, int32_t n, int32_t tokens, int32_t embed_dim, int32_t num_heads,
                             void *stream);

/* Depth-1 fast path of the folded cls attention (the cls query is a constant of the weights, so m is shared):
 *   azk_nn_patch_embed_scores - writes xn = (tokens - mean) * rstd (LayerNorm WITHOUT the affine; the caller folds gamma/beta
 *       into m', c and the value projection; ln_w/ln_b are ignored) and scores_out[b][h][t] = xn[b][t][:] . m'[h][:]
 *       (float32 [n][H][Tp], Tp = 16*ceil(T/16)).  The scores ride on the matrix cores: wt_bf16_dev has 16 extra rows
 *       [D .. D+16) holding m'_h^T Wconv (heads >= H zero), so x . m' comes out of the same MFMA chain as 16 extra output
 *       columns; score_cpos_dev float32 [T][16] = m'_h . cpos[t] is their additive term, score_msum_dev float32 [16] =
 *       sum_d m'_h[d], and xn . m' = rstd * (x . m' - mean * sum(m')).
 *   azk_nn_cls_pool - a = softmax_t(scores + c[h]);  z[b][h][:] = sum_t a[h][t] * xhat[b][t][:]  ([n][H][D] bf16):
 *       one streaming pass over xhat with no cross-lane reductions (HBM-read-bound).
 * n_valid_dev (optional, device int32): only the first min(n, *n_valid_dev) boards are processed - lets a captured
 * hipGraph with fixed launch sizes do work proportional to the live leaf count (azk_step_select's n_leaf_dev).
 * Heads: 8 or 4 (and 2*H <= 16). */
int32_t azk_nn_patch_embed_scores(const void *boards_dev, int32_t boards_are_f32, const void *wt_bf16_dev,
                                  const float *cpos_dev, const float *ln_w_dev, const float *ln_b_dev,
                                  void *xhat_out_bf16_dev, const float *score_cpos_dev, const float *score_msum_dev,
                                  float *scores_out_dev, int32_t num_heads, int32_t n, int32_t channels, int32_t rows, int32_t cols,
                                  int32_t ksize, int32_t kp, int32_t embed_dim, float ln_eps,
                                  const int32_t *n_valid_dev, void *stream);
int32_t azk_nn_cls_pool(const void *xhat_bf16_dev, const float *scores_dev, const float *c_dev, void *z_out_bf16_dev,
                        int32_t n, int32_t tokens, int32_t embed_dim, int32_t num_heads, const int32_t *n_valid_dev,
                        void *stream);

/* azk_nn_embed_pool - the two calls above fused: the normalised tokens stay on the CU (one workgroup per board, the
 * LayerNorm statistics cross its four waves through LDS, softmax over the tokens, the weighted token sum as a second
 * MFMA on the tile while it is still in registers).  Operands as azk_nn_patch_embed_scores, with the per-token constants
 * padded to whole 16-token tiles (Tp = 16 ceil(T/16) rows; cpos padding rows 0, score-constant padding rows -1e30 in the
 * head columns and 0 elsewhere) and stored in the kernel's accumulator order:
 *   cpos_frag_dev  [Tp/16][4][8][64][4]: [tile][w][q][lane][r] = cpos[16 tile + 4 (lane>>4) + r][128 w + 8 (lane&15) + q]
 *   score_frag_dev [Tp/16][64][4]:       [tile][lane][r]       = score_cpos[16 tile + 4 (lane>>4) + r][lane&15]
 * Column 15 of the extra weight rows / score constants is the row mean ((1/D) sum_d W[d][k], (1/D) sum_d cpos[t][d]).  score_ref_dev [16] (optional): per-head upper bound of
 * the scores (sqrt(D) |m'_h|) used as the softmax reference when 2*bound cannot underflow exp(); NULL = running maximum.
 * The constant c[h] of azk_nn_cls_pool drops out of a softmax over tokens.  embed_dim 512, heads 8 or 4.
 * z_out [n][H][512] bf16. */
int32_t azk_nn_embed_pool(const void *boards_dev, int32_t boards_are_f32, const void *wt_ext_bf16_dev,
                          const float *cpos_frag_dev, const float *score_frag_dev, const float *score_msum_dev,
                          const float *score_ref_dev, void *z_out_bf16_dev, int32_t num_heads, int32_t n, int32_t channels,
                          int32_t rows, int32_t cols, int32_t ksize, int32_t kp, int32_t embed_dim, float ln_eps,
                          const int32_t *n_valid_dev, void *stream);

/* Board source for the fused step: azk_nn_embed_pool_leaves takes the pending leaves of the last azk_step_tree straight from
 * the engine - it builds the prefix over the leaf flags itself (board j = the j-th flagged game in ascending game order,
 * exactly azk_step_gather's order), reads that game's cell codes, stores the slot j the next expansion will read its
 * logits row from, and writes the leaf count to n_leaf - so a simulation step is azk_step_tree -> azk_nn_embed_pool_leaves
 * -> tail, without azk_step_gather and without an evaluator batch.  Pointers are device pointers owned by the engine
 * (valid until azk_destroy) except n_leaf, which is the caller's. */
typedef struct azk_leaf_source {
    const uint8_t *leaf_flag;     /* [flag_bytes] 1 = the game has a pending leaf that needs the evaluator */
    const uint8_t *leaf_cells;    /* [n_games][rc_pad] cell codes at the leaf (bit 0 / 1 = player 0 / 1 stone) */
    const int32_t *to_move;       /* [n_games] side to move at the root */
    const int32_t *leaf_depth;    /* [n_games] */
    int32_t *leaf_slot;           /* [n_games] out: row of the evaluator outputs */
    int32_t *n_leaf;              /* [1] out */
    int32_t n_games, rows, cols, rc, rc_pad, planes, flag_bytes;
    uint32_t *cache_stamp;        /* shared eval cache: launch stamp the hand-off kernel bumps (NULL otherwise) */
} azk_leaf_source;
int32_t azk_leaf_source_of(azk_engine *e, int32_t *n_leaf_dev, azk_leaf_source *out);
int32_t azk_nn_embed_pool_leaves(const azk_leaf_source *src, const void *wt_ext_bf16_dev, const float *cpos_frag_dev,
                                 const float *score_frag_dev, const float *score_msum_dev, const float *score_ref_dev,
                                 void *z_out_bf16_dev, int32_t num_heads, int32_t ksize, int32_t kp, int32_t embed_dim,
                                 float ln_eps, void *stream);

/* azk_nn_embed_pool_compact(_leaves) - azk_nn_embed_pool restricted to the tokens a stone can reach (static softmax
 * reference only).  A token whose k x k patch is empty is a constant of the weights, so its contribution to the weighted
 * token sum and to the softmax denominator is precomputed over ALL tokens (z_all, l_all) and the kernel only evaluates the
 * "dirty" tokens - adding their real contribution and subtracting their constant one in the same MFMA.  The per-token
 * tables are indexed by token (T = rows*cols + 1 tokens; row T is the null token that pads the last tile):
 *   cpos_tok    f32  [T+1][512]  bias + positional term (row 0: cls + pos[0]); row T = 0
 *   score_tok   f32  [T+1][16]   score constants (column 15: row mean of cpos); row T = -1e30 in the head columns, 0 elsewhere
 *   wconst_tok  f32  [T+1][16]   exp(score - ref) of the token taken as an empty-patch token; 0 beyond num_heads and in row T
 *   xnconst_tok bf16 [T+1][512]  LayerNorm (no affine) of cpos_tok; row T = 0
 *   z_all       f32  [4][8][64][4] sum_t bf16(wconst)[t][h] * xnconst[t][col] at [w][q][lane][j]: h = 4 (lane>>4) + j, col = 128 w + 8 (lane&15) + q
 *   l_all       f32  [16]        sum_t wconst[t][h]
 *   wt_frag     bf16 the conv weight with its 16 extra rows (azk_nn_embed_pool's wt_ext [512+16][kp]) in MFMA fragment order:
 *               [33 column tiles][kp/32][64 lanes][8]: element [ct][s][l][i] = wt_ext[col(ct, l)][32 s + 8 (l>>4) + i],
 *               col(ct, l) = 128 (ct>>3) + 8 (l&15) + (ct&7) for ct < 32, 512 + (l&15) for ct = 32
 * sched_dev: int32 [2] work-queue words, zero before the first launch (each launch leaves them zero); one buffer per
 * stream that may run the kernel concurrently.  Boards differ in cost, so workgroups pull them from that queue.
 * Channels x ksize in {2, 3} x {3, 5}; kp = 32 ceil(channels ksize^2 / 32). */
typedef struct azk_embed_pool_consts {
    const void *wt_frag;
    const float *cpos_tok, *score_tok, *wconst_tok;
    const void *xnconst_tok;
    const float *z_all, *l_all, *score_msum, *score_ref;
    int32_t num_heads, ksize, kp, embed_dim;
    float ln_eps;
    uint64_t *work_stats;   /* optional device uint64 [2]: += boards evaluated, += 16-token tiles evaluated (bench accounting) */
} azk_embed_pool_consts;
int32_t azk_nn_embed_pool_compact(const void *boards_dev, int32_t boards_are_f32, const azk_embed_pool_consts *consts,
                                  void *z_out_bf16_dev, int32_t n, int32_t channels, int32_t rows, int32_t cols,
                                  const int32_t *n_valid_dev, int32_t *sched_dev, void *stream);
/* azk_nn_embed_pool_compact_leaves ranks the pending leaves with 16-bit counters: engines with more pending-leaf slots
 * (n_games * leaves_per_step) than this get AZK_ERR_ARG and keep azk_nn_embed_pool_leaves */
#define AZK_EMBED_POOL_COMPACT_MAX_SLOTS 65279
int32_t azk_nn_embed_pool_compact_leaves(const azk_leaf_source *src, const azk_embed_pool_consts *consts, void *z_out_bf16_dev,
                                         int32_t *sched_dev, void *stream);

/* ---- cls-row tail (nn.py:54-60, 78-83 for the row the heads read): small-M GEMMs with a device-side row count.
 * azk_nn_gemm_rows: C = A W^T for A bf16 [m][lda] (first k columns), W = an nn.Linear weight [n_out][k] packed in MFMA
 *   B-fragment order: Wp[n_out/64][k/32][4][64][8] with element [g][s][c][lane][i] = W[64 g + 4 (lane&15) + c][32 s + 8 (lane>>4) + i]
 *   (n_out a multiple of 64, k of 32).  Either partials_out_dev != NULL: the K range is split over `ksplit` waves and
 *   float32 partial sums are written to [ksplit][m][n_out] planes (the row-wise kernel that follows adds them: no
 *   atomics); or gelu_out_bf16_dev != NULL (ksplit 1): bf16 [m][n_out] = GELU(A W^T + bias) (nn.GELU, exact erf).
 * azk_nn_layernorm_sum: x = sum of nsplit partial planes (plane stride m_stride rows) (+ bias) (+ resid bf16);
 *   y = LayerNorm(x) bf16; optional x_out = x + add_bias bf16 (the residual the next product is added to).
 * azk_nn_heads_finalize_sum: logits / tanh(value) from the partial planes of the merged head GEMM (+ bias).
 * n_valid_dev as above: rows at or beyond it are neither read nor written. */
int32_t azk_nn_gemm_rows(const void *a_bf16_dev, int32_t lda, const void *w_packed_dev, int32_t m, int32_t n_out, int32_t k,
                         int32_t ksplit, float *partials_out_dev, const float *bias_dev, void *gelu_out_bf16_dev,
                         const int32_t *n_valid_dev, void *stream);
/* azk_nn_tail_gemm - one link of the cls-row tail as a latency-shaped small GEMM (every load of a K chunk in flight before
 * the first MFMA, no LDS):  C[m][nbatch * n_out] = op(A) W^T (+ bias) through one of four epilogues.
 *   a_bf16 [m][lda] row-major; batch b reads A columns [b * a_batch_stride, b * a_batch_stride + k), multiplies them with
 *   weight block b and writes output columns [b * n_out, (b + 1) * n_out) - nbatch = 1 is a plain GEMM, nbatch = heads is the
 *   block-diagonal per-head value projection.  w_packed: nbatch consecutive nn.Linear weights [n_out][k] in
 *   azk_nn_gemm_rows' fragment packing.  k = 512 or 2048 (or 384 = AZK_EMBED_FOLD_ROW, plain bf16 epilogue only); n_out a multiple of 64.
 *   layernorm_a (k = 512): A = LayerNorm(rows) without affine (fold it into weight / bias: W diag(gamma), W beta + b); the row
 *            statistics are read from a_stats [m][a_stats_groups][2] = per 64-column group (sum, sum of squares) of the row, as
 *            left by the GEMM that produced A (its stats_out); a_stats_groups must be 8 (= k / 64; 64-byte rows, 16-byte aligned).
 *   stats_out (optional, epilogues 0-2): float32 [m][nbatch * n_out / 64][2], the same partials of the rows written here.
 *   epilogue 0: out_bf16 = acc + bias;  1: GELU(acc + bias) (exact erf);  2: acc + bias + resid_bf16;
 *            3: merged heads - logits_out float32 [m][action_dim], values_out[m] = tanh(column action_dim)   (nn.py:82-83).
 *   n_valid (optional, device): rows at or beyond it are neither read nor written. */
typedef struct azk_tail_gemm {
    const void *a_bf16; int32_t lda, a_batch_stride;
    const void *w_packed;
    int32_t m, n_out, k, nbatch;
    const int32_t *n_valid;
    const float *bias;
    int32_t layernorm_a, epilogue;
    float ln_eps;
    const float *a_stats; int32_t a_stats_groups;
    float *stats_out;
    void *out_bf16; int32_t ldo;
    const void *resid_bf16; int32_t ldr;
    float *logits_out, *values_out; int32_t action_dim;
    const float *a_col_sums;          /* azk_nn_tail_gemm_lds with layernorm_a: float32 [n_out] column sums of the bf16 weight (ABI 4) */
} azk_tail_gemm;
int32_t azk_nn_tail_gemm(const azk_tail_gemm *desc, void *stream);
/* azk_nn_tail_gemm_lds - the same descriptor, the two WIDE links of the tail (nn.py:58-60) as LDS-staged GEMMs (csrc/azk_tail.hip:
 * activation and weight tiles to LDS by LDS-DMA in full 128-byte lines, a ring of K stages retired by counted waits, eight waves per
 * workgroup at two waves per SIMD):
 *   k = 512,  epilogue 1 (GELU), layernorm_a = 1, n_out a multiple of 128: LayerNorm applied in the epilogue,
 *             out = GELU(rstd (A W^T - mean a_col_sums) + bias) - A enters the matrix pipe as stored (no re-rounded normalised copy);
 *             a_stats as for azk_nn_tail_gemm, a_col_sums required, no stats_out;
 *   k = 2048, epilogue 2 (residual), layernorm_a = 0, n_out a multiple of 64: K split over four wave groups whose chains are added in
 *             the fixed order 0..3 - bit for bit azk_nn_tail_gemm's result on the same inputs; optional stats_out.
 * Anything else: AZK_ERR_ARG (use azk_nn_tail_gemm). */
int32_t azk_nn_tail_gemm_lds(const azk_tail_gemm *desc, void *stream);
/* process-wide launch-shape knobs for callers that step several game groups on separate streams (selfplay.SelfPlayRunner n_split > 1):
 * azk_nn_tail_lds_footprint(1): the K = 2048 link keeps two LDS ring buffers (96 KB) instead of three (144 KB);
 * azk_nn_embed_fold_grid(n): azk_nn_embed_fold(_leaves) launches at most n workgroups (0 = default, two per CU).  Results do not change. */
int32_t azk_nn_tail_lds_footprint(int32_t small);
int32_t azk_nn_embed_fold_grid(int32_t max_workgroups);

/* azk_nn_ln_heads: final LayerNorm + merged policy/value head + finalize in one launch (nn.py:78-83 for the cls row):
 *   logits[n][A] = LN(x) Wh^T + bh (float32), values[n] = tanh(column A).  w_packed_dev: the merged head weight
 *   [n_out_padded][embed_dim] in azk_nn_gemm_rows' packing; each wave reads whole rows and takes their statistics itself.
 *   ln_w_dev = ln_b_dev = NULL: LayerNorm's affine is already folded into the operands (weight W diag(gamma), bias
 *   W beta + b); the kernel then fetches its 16-row slab once and normalises from registers (embed_dim 256 or 512). */
int32_t azk_nn_ln_heads(const void *x_bf16_dev, const float *ln_w_dev, const float *ln_b_dev, float eps, const void *w_packed_dev,
                        const float *bias_dev, int32_t n, int32_t embed_dim, int32_t n_out_padded, int32_t action_dim,
                        float *logits_out_dev, float *values_out_dev, const int32_t *n_valid_dev, void *stream);
int32_t azk_nn_layernorm_sum(const float *partials_dev, int32_t nsplit, int32_t m_stride, const float *bias_dev,
                             const void *resid_bf16_dev, const float *w_dev, const float *b_dev, float eps, void *y_bf16_dev,
                             const float *add_bias_dev, void *x_out_bf16_dev, int32_t n, int32_t embed_dim,
                             const int32_t *n_valid_dev, void *stream);
int32_t azk_nn_heads_finalize_sum(const float *partials_dev, int32_t nsplit, int32_t m_stride, int32_t ld, const float *bias_dev,
                                  int32_t action_dim, int32_t n, float *logits_out_dev, float *values_out_dev,
                                  const int32_t *n_valid_dev, void *stream);

/* azk_nn_embed_fold(_leaves) - the embedding + cls pooling WITHOUT forming the token rows (round 3; ai/nn.py:7-27, 36-56 for the
 * cls row, as azk_nn_embed_pool_compact + the first azk_nn_tail_gemm link).  With x_t = Wc p_t + cpos_t (p_t: the 0/1 patch of token
 * t, <= 64 bits) LayerNorm1's statistics and the cls scores of a token are functions of its patch bits alone:
 *     x_t - mean(x_t) = Wt p_t + ct_t        D var_t = p_t' G p_t + u2_t . p_t + n_t        s_t[h] = rstd_t (S_h . p_t + sc_t[h])
 * (Wt = Wc minus its column means, ct_t = cpos_t minus its mean, G = Wt' Wt, u2_t = 2 Wt' ct_t, n_t = |ct_t|^2, S_h = Wt' m'_h,
 * sc_t[h] = m'_h . ct_t), and the value-projected pooled row is linear in x_t:
 *     u_h = Wv'_h z_h = (1 / L_h) sum_t a_t[h] (M_h p_t + D_t[h]),   a = w rstd,   M_h = Wv'_h Wt,   D_t[h] = Wv'_h ct_t.
 * The kernel evaluates the tokens a stone can reach (the rest are constants of the weights, summed ahead: U_all, l_all) and writes,
 * per board and head, one bf16 row of AZK_EMBED_FOLD_ROW entries:
 *     [0, T)       (a_t - aconst_t) / L      (0 for tokens no stone reaches)
 *     [T, T+3)     1 / L as bf16 hi, lo, hi  (against U_all hi, U_all hi, U_all lo in the weight)
 *     [256, 320)   sum_t a_t p_t / L         (the pooled patch; entries >= channels ksize^2 are 0)
 *     elsewhere 0
 * so that ONE batched azk_nn_tail_gemm (nbatch = heads, k = AZK_EMBED_FOLD_ROW, weight rows [D_t[h]; U_all; ...; M_h]) yields u, the
 * input of the output projection.  Tables (T = rows cols + 1 tokens, row T = the null token that pads the last tile):
 *   g_frag      f16 [2 (hi, lo)][4 column tiles][2 k-steps][64 lanes][8]: G x g_scale as two terms, MFMA B-fragment order
 *               (element [t][q][s][l][i] = term t of G[16 q + (l & 15)][32 s + 8 (l >> 4) + i])
 *   e_frag      f16 [2][2][64][8]: the score columns S (column h < heads; others 0) x e_scale, same order
 *   u2_tok      f32 [T+1][64];   score_tok f32 [T+1][16]: sc_t[h], column 15 = n_t (row T: -1e30 in the head columns, column 15 = 512)
 *   wconst_tok  f32 [T+1][16]: exp(rstdc_t sc_t[h] - ref[h]), column 15 = rstdc_t = rsqrt(n_t / 512 + eps)
 *   l_all, score_ref f32 [16] (score_ref >= 1e30 beyond the heads)
 * Covered: embed_dim 512, head dim 64, 4 / 8 heads, channels ksize^2 <= 64, T + 3 <= 256. */
#define AZK_EMBED_FOLD_ROW 384
/* azk_nn_embed_fold_leaves keeps the rank -> game table of the launch in LDS: engines with more pending-leaf slots than this get
 * AZK_ERR_ARG and keep azk_nn_embed_pool_compact_leaves */
#define AZK_EMBED_FOLD_MAX_SLOTS 8192
typedef struct azk_embed_fold_consts {
    const void *g_frag, *e_frag;
    const float *u2_tok, *score_tok, *wconst_tok, *l_all, *score_ref;
    const float *inv_scales;    /* device [2]: 1 / g_scale, 1 / e_scale (device data, so that new weights can be copied in under a captured graph) */
    int32_t num_heads, ksize, embed_dim;
    float ln_eps;
    uint64_t *work_stats;   /* optional device uint64 [2]: += boards evaluated, += 16-token tiles evaluated */
} azk_embed_fold_consts;
int32_t azk_nn_embed_fold(const void *boards_dev, int32_t boards_are_f32, const azk_embed_fold_consts *consts, void *rows_out_bf16_dev,
                          int32_t n, int32_t channels, int32_t rows, int32_t cols, const int32_t *n_valid_dev, int32_t *sched_dev,
                          void *stream);
int32_t azk_nn_embed_fold_leaves(const azk_leaf_source *src, const azk_embed_fold_consts *consts, void *rows_out_bf16_dev,
                                 int32_t *sched_dev, void *stream);
/* azk_nnx_embed_fold(_leaves): the same function in float32-accurate arithmetic for the fp32 line (ai/nn.py:74-84 at ai/mcts.py:46 in the
 * reference's own precision): correctly rounded rsqrt, exp with an extended-precision argument, the pooled patch on v_mfma_f32_16x16x4_f32,
 * float32 rows out ([T] = 1 / L, [T+1], [T+2] = 0) for azk_nnx_gemm_h's float32-A link (k = AZK_EMBED_FOLD_ROW). */
int32_t azk_nnx_embed_fold(const void *boards_dev, int32_t boards_are_f32, const azk_embed_fold_consts *consts, float *rows_out_f32_dev,
                           int32_t n, int32_t channels, int32_t rows, int32_t cols, const int32_t *n_valid_dev, int32_t *sched_dev,
                           void *stream);
int32_t azk_nnx_embed_fold_leaves(const azk_leaf_source *src, const azk_embed_fold_consts *consts, float *rows_out_f32_dev,
                                  int32_t *sched_dev, void *stream);

/* ---- fp32-accurate network path (csrc/azk_nnx.hip): the reference evaluates its network in float32 (ai/nn.py:74-84 called at
 * ai/mcts.py:46), and north_star asks for visit-count policies within 1e-5 of it.  The same two stages as above - boards -> pooled
 * tokens (azk_nn_embed_pool_compact's function) -> cls-row tail (azk_nn_tail_gemm's function) - in arithmetic that keeps float32
 * accuracy end to end: the 0/1 board against the conv weight (and the folded score / mean columns) split into two fp16 terms
 * (exact products, float32 accumulation on v_mfma_f32_16x16x32_f16), everything with two run-time operands on v_mfma_f32_16x16x4_f32,
 * statistics / softmax / GELU(erf) / tanh in float32.  Tables as azk_embed_pool_consts, all float32, with
 *   wt_frag     fp16 [33 column tiles][kp/32][2: hi, lo][64 lanes][8]: element [ct][s][p][l][i] = the p-th fp16 term of
 *               wt_scale * wt_ext[col(ct, l)][32 s + 8 (l>>4) + i], col(ct, l) = 64 (ct>>2) + 4 (l&15) + (ct&3) for ct < 32, 512 + (l&15) for ct = 32
 *   xnconst_tok fp16 [T+1][128][8]: the constant tokens' normalised rows x 16 as two fp16 terms, per 4 columns (hi0..hi3, lo0..lo3)
 *   score_ref   per head the largest score of a constant token (their weights wconst are in (0, 1]); score bound <= 40 required
 *   z_all       f32 [8][4][64][4]: pool_scale * sum_t wconst[t][h] * xnconst[t][col] (from the fp16 terms) at [g][q][lane][j]: h = 4 (lane>>4) + j, col = 64 g + 4 (lane&15) + q
 * kp <= 64 (the hi / lo image must fit one CU's LDS), tokens <= 256.  z_out float32 [n][H][512]. */
typedef struct azk_embed_pool_x_consts {
    const void *wt_frag;
    const float *cpos_tok, *score_tok, *wconst_tok;
    const void *xnconst_tok;
    const float *z_all, *l_all, *score_msum, *score_ref;
    int32_t num_heads, ksize, kp, embed_dim;
    float ln_eps, wt_scale;
    uint64_t *work_stats;   /* optional device uint64 [2]: += boards evaluated, += 16-token tiles evaluated */
    const uint32_t *wconst_h16_tok;   /* [T+1][16]: -wconst * 64 as two fp16 terms, hi | lo << 16 */
    float pool_scale;                 /* 64 * 16: the unit of z_all and of the kernel's accumulators */
} azk_embed_pool_x_consts;
int32_t azk_nnx_embed_pool(const void *boards_dev, int32_t boards_are_f32, const azk_embed_pool_x_consts *consts,
                           float *z_out_f32_dev, int32_t n, int32_t channels, int32_t rows, int32_t cols,
                           const int32_t *n_valid_dev, int32_t *sched_dev, void *stream);
int32_t azk_nnx_embed_pool_leaves(const azk_leaf_source *src, const azk_embed_pool_x_consts *consts, float *z_out_f32_dev,
                                  int32_t *sched_dev, void *stream);
/* azk_nnx_gemm - azk_nn_tail_gemm in float32: a_f32 [m][lda], out_f32 / resid_f32 float32, w_packed = nbatch consecutive nn.Linear
 * weights [n_out][k] as float32 in fragment order Wp[n_out/64][k/16][4][64 lanes][4]: element [g][s][c][lane][i] =
 * W[64 g + 4 (lane&15) + c][16 s + 4 (lane>>4) + i].  k = 512 or 2048.  layernorm_a (k = 512): a_stats [m][8][2] as left by the producing
 * call's stats_out.  Epilogues 0-3 as azk_nn_tail_gemm (GELU = exact erf form via erff). */
typedef struct azk_gemm_x {
    const float *a_f32; int32_t lda, a_batch_stride;
    const float *w_packed;
    int32_t m, n_out, k, nbatch;
    const int32_t *n_valid;
    const float *bias;
    int32_t layernorm_a, epilogue;
    float ln_eps;
    const float *a_stats;
    float *stats_out;
    float *out_f32; int32_t ldo;
    const float *resid_f32; int32_t ldr;
    float *logits_out, *values_out; int32_t action_dim;
} azk_gemm_x;
int32_t azk_nnx_gemm(const azk_gemm_x *desc, void *stream);

/* azk_nnx_gemm_h - the same link with every operand as TWO fp16 terms (x scale = hi + lo, 22 significant bits; products hi*hi + hi*lo +
 * lo*hi exact in the float32 accumulator) on v_mfma_f32_16x16x32_f16: a fifth of the matrix-pipe time of the float32-input MFMA.
 *   A: either (a_hi, a_lo) fp16 planes [m][lda] holding a * a_scale (written by the producing call's out_hi / out_lo), or a_f32
 *      float32 [m][lda] split on the fly (first link only: k = 512, epilogue 0);
 *   w_packed: nbatch weights [n_out][k] x w_scale as fp16 (hi, lo) in fragment order Wp[n_out/64][k/32][4][2][64 lanes][8]:
 *      element [g][s][c][p][lane][i] = term p of W[64 g + 4 (lane&15) + c][32 s + 8 (lane>>4) + i];
 *   layernorm_a: LayerNorm moves into the epilogue - out = rstd (acc - mean col_sums[n]) + bias, col_sums[n] = sum_k W[n][k] (of the
 *      reconstructed terms), mean / rstd from a_stats as in azk_nnx_gemm;
 *   outputs: out_f32 and / or (out_hi, out_lo) planes (x a_scale) [m][ldo]; stats_out / heads as azk_nnx_gemm. */
typedef struct azk_gemm_h {
    const void *a_hi, *a_lo; const float *a_f32; int32_t lda, a_batch_stride;
    const void *w_packed;
    int32_t m, n_out, k, nbatch;
    const int32_t *n_valid;
    const float *bias, *col_sums;
    int32_t layernorm_a, epilogue;
    float ln_eps, a_scale, w_scale;
    const float *a_stats;
    float *stats_out;
    void *out_hi, *out_lo; float *out_f32; int32_t ldo;
    const float *resid_f32; int32_t ldr;
    float *logits_out, *values_out; int32_t action_dim;
    int32_t *overflow_flag;       /* optional, device (ABI 4): set to 1 when a value written to (or split into) the fp16 planes leaves fp16's range
                                   * (|x a_scale| >= 65504): the planes then hold inf and the caller must not trust the outputs */
} azk_gemm_h;
int32_t azk_nnx_gemm_h(const azk_gemm_h *desc, void *stream);
/* azk_nnx_gemm_h_lds - the same descriptor, the two WIDE links (k = 512 -> n_out, LayerNorm + GELU; k = 2048, residual) with both
 * operand planes staged through LDS by LDS-DMA (csrc/azk_tail.hip, as azk_nn_tail_gemm_lds): the same accumulation chains in the same
 * order and the same epilogue arithmetic as azk_nnx_gemm_h.  (a_hi, a_lo) planes only, nbatch = 1; anything else: AZK_ERR_ARG. */
int32_t azk_nnx_gemm_h_lds(const azk_gemm_h *desc, void *stream);

/* ---- vanilla mode: MCTS.mcts(model=None, ...) (mcts.py:57-59), MCTS.simulate (mcts.py:62-79), UCB1 of
 * utils.py:29-44 mode 'normal'.  A search is azk_begin_search(e, NULL) followed by azk_vanilla_search calls summing to
 * n simulations (each launch runs its simulations - select, expand, random rollout, backup - entirely on the device);
 * results are read with azk_root_stats / azk_root_children / azk_export_tree / azk_advance as in network mode.
 * Random numbers: np.random.randint of the legacy global RandomState, i.e. MT19937 + numpy's masked rejection, on one
 * MT19937 state per game: 625 uint32 = the 624 key words and the position, the layout of np.random.get_state()[1:3].
 * Seeding a game with the caller's np.random state makes the search consume exactly the reference's stream; the state
 * read back is what np.random.set_state() needs afterwards.  Default states: init_genrand(5489 + game). */
int32_t azk_vanilla_set_rng(azk_engine *e, int32_t first, int32_t count, const uint32_t *mt_states_host, void *stream);
int32_t azk_vanilla_get_rng(azk_engine *e, int32_t first, int32_t count, uint32_t *mt_states_host, void *stream);  /* synchronises */
int32_t azk_vanilla_search(azk_engine *e, int32_t n_sims, void *stream);

/* ---- the full-token transformer block (ai/nn.py:38-61) for networks with depth > 1 (csrc/azk_block.hip) ----
 * azk_nn_gemm_tok: out[m][n_out] = A[m][k] W^T (+ bias) through an epilogue, LDS-staged (LDS-DMA in full lines, counted-wait ring):
 *   a_bf16 [m][lda] row-major; w_packed = an nn.Linear weight [n_out][k] in azk_nn_gemm_rows' fragment packing (n_out a multiple of
 *   128 - pad with zero rows -, k a multiple of 64, k >= 128); epilogue 0: bf16 out; 1: bf16 GELU(.) (erf form); 2: bf16 out = . +
 *   resid_bf16[m][ldr]; 4: float32 out.  n_valid (optional, device): rows at or beyond it are neither read nor written. */
typedef struct azk_gemm_tok {
    const void *a_bf16; int32_t lda;
    const void *w_packed;
    int32_t m, n_out, k;
    const int32_t *n_valid;
    const float *bias;
    int32_t epilogue;
    void *out; int32_t ldo;
    const void *resid_bf16; int32_t ldr;
} azk_gemm_tok;
int32_t azk_nn_gemm_tok(const azk_gemm_tok *desc, void *stream);
/* azk_nn_attention_tok: nn.MultiheadAttention (eval mode) over ALL tokens of every board: out[b][t][h dh ..] = softmax_t'(q k^T /
 * sqrt(dh)) v per board b and head h.  qkv_bf16 [n_boards][tokens][3 embed_dim] = the in-projection's output (q | k | v);
 * tokens <= 256; head dimension 32 or 64.  n_valid (optional, device): boards at or beyond it are skipped. */
int32_t azk_nn_attention_tok(const void *qkv_bf16_dev, void *out_bf16_dev, int32_t n_boards, int32_t tokens, int32_t embed_dim,
                             int32_t num_heads, const int32_t *n_valid_dev, void *stream);

/* nn.LayerNorm over the rows of a bf16 matrix [n][embed_dim] (norm2 / norm of nn.py:41-42,78; fp32 statistics) -> y;
 * with add_bias_dev != NULL the rows of x are also replaced by x + add_bias (the residual the next GEMM accumulates
 * onto, nn.py:59-60).  embed_dim in {128, 256, 512}. */
int32_t azk_nn_layernorm_rows(void *x_bf16_dev, const float *w_dev, const float *b_dev, float eps, void *y_bf16_dev,
                              const float *add_bias_dev, int32_t n, int32_t embed_dim, const int32_t *n_valid_dev,
                              void *stream);

/* Merged policy/value head output (bf16 [n][ld]: columns [0, A) logits, column A the raw value) -> float32 logits [n][A]
 * and values [n] = tanh(raw) (nn.py:82-83) in one launch; n_valid_dev as above. */
int32_t azk_nn_heads_finalize(const void *heads_bf16_dev, int32_t ld, int32_t action_dim, int32_t n, float *logits_out_dev,
                              float *values_out_dev, const int32_t *n_valid_dev, void *stream);

/* float32 softmax exactly as the engine applies it to logits (test hook; [n][A] -> [n][A]) */
int32_t azk_softmax_rows(const float *logits_dev, int32_t n, int32_t action_dim, float *out_dev, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* AZK_H */
