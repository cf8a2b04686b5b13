/*
 * az_oracle.c - CPU restatement of the reference's self-play hot path.
 *
 * TEST INFRASTRUCTURE.  This file is the parity ORACLE: only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (alpha-zero_amd/, libazk.so)
 * never links, loads or calls anything in oracle/.
 *
 * Parity status: PINNED.  Every function below is checked bit-for-bit against golden vectors
 * produced by running the reference itself (tests/golden/generate_golden.py; CPython 3.10.12,
 * numpy 2.2.6): rule KATs, whole search trees (sha256 over every node), whole self-play games.
 *
 * Each function cites the reference lines it restates (paths relative to the reference root).
 * Two behaviours of the reference are interpreter/library facts rather than code it spells out,
 * and are restated here because results depend on them:
 *   (1) games/gomoku.py:93-106 returns list(set_of_(r,c)_tuples): the child ORDER is CPython's
 *       set iteration order.  py_set_* below restates CPython 3.10 Objects/setobject.c
 *       (open addressing, LINEAR_PROBES 9, PERTURB_SHIFT 5, resize at fill*5 >= mask*3 to the
 *       first power of two > 4*used) and Objects/tupleobject.c tuplehash (xxHash-style, 64-bit).
 *   (2) utils.py:29-44 mixes numpy float32 scalars (child.prior) with Python floats.  Under
 *       numpy >= 2 (NEP 50) Python floats are "weak", so the UCB of a node whose prior is a
 *       float32 is computed in FLOAT32; root priors after Dirichlet mixing are float64
 *       (utils.py:24-25) and so is their UCB.  ucb_f32()/ucb_f64() restate both.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define AZO_TTT 0
#define AZO_C4 1
#define AZO_GOMOKU 2
#define AZO_MAX_CELLS 512

typedef struct {
    int kind, rows, cols, planes, win_len, action_dim, state_dim;
} azo_game_t;

/* games/tictactoe.py:10-12, games/connect4.py:8-10, games/gomoku.py:10-13 (rows/cols overridable) */
int azo_game_init(azo_game_t *g, int kind, int rows, int cols) {
    memset(g, 0, sizeof *g);
    g->kind = kind;
    if (kind == AZO_TTT) { rows = 3; cols = 3; g->planes = 3; g->win_len = 3; g->action_dim = 9; }
    else if (kind == AZO_C4) { rows = 6; cols = 7; g->planes = 3; g->win_len = 4; g->action_dim = 7; }
    else if (kind == AZO_GOMOKU) { g->planes = 2; g->win_len = 5; g->action_dim = rows * cols; }
    else return -1;
    if (rows * cols > AZO_MAX_CELLS || rows < 1 || cols < 1) return -1;
    g->rows = rows; g->cols = cols; g->state_dim = rows * cols;
    return 0;
}

static inline int occupied(const azo_game_t *g, const float *b, int cell) {
    int rc = g->rows * g->cols;
    return !(b[cell] == 0.0f && b[rc + cell] == 0.0f);
}

/* get_action_idx: tictactoe.py:34, connect4.py:29 (column), gomoku.py:48 */
int azo_action_idx(const azo_game_t *g, int cell) {
    return g->kind == AZO_C4 ? cell % g->cols : cell;
}

/* ---------------------------------------------------------------------------------------
 * CPython 3.10 set of (r, c) tuples: iteration order.  (Objects/setobject.c, tupleobject.c)
 * ------------------------------------------------------------------------------------- */
#define XXPRIME_1 11400714785074694791ULL
#define XXPRIME_2 14029467366897019727ULL
#define XXPRIME_5 2870177450012600261ULL

uint64_t azo_py_tuple2_hash(int r, int c) {
    uint64_t acc = XXPRIME_5;
    uint64_t lanes[2] = {(uint64_t)(int64_t)r, (uint64_t)(int64_t)c}; /* hash(small int >= 0) == the int */
    for (int i = 0; i < 2; i++) {
        acc += lanes[i] * XXPRIME_2;
        acc = (acc << 31) | (acc >> 33);
        acc *= XXPRIME_1;
    }
    acc += 2ULL ^ (XXPRIME_5 ^ 3527539ULL);
    if (acc == (uint64_t)-1) return 1546275796ULL;
    return acc;
}

typedef struct {
    int16_t key[2048];   /* cell + 1; 0 = empty slot (<= 512 keys -> table <= 2048) */
    uint64_t hash[2048];
    int mask, fill;      /* no deletions ever happen: fill == used */
} py_set_t;

static void py_set_init(py_set_t *s) { memset(s->key, 0, sizeof s->key); s->mask = 7; s->fill = 0; }

static void py_set_insert_clean(int16_t *key, uint64_t *hash, size_t mask, int16_t k, uint64_t h) {
    size_t perturb = h, i = (size_t)h & mask;
    for (;;) {
        if (key[i] == 0) goto found;
        if (i + 9 <= mask) {
            for (int j = 0; j < 9; j++) { i++; if (key[i] == 0) goto found; }
            /* CPython advances a pointer `entry`, leaving i unchanged; only the slot test matters */
            i -= 9;
        }
        perturb >>= 5;
        i = (i * 5 + 1 + perturb) & mask;
    }
found:
    key[i] = k; hash[i] = h;
}

static void py_set_add(py_set_t *s, int16_t k, uint64_t h) {
    size_t mask = (size_t)s->mask, perturb = h, i = (size_t)h & mask, slot;
    for (;;) {
        int probes = (i + 9 <= mask) ? 9 : 0;
        slot = i;
        do {
            if (s->key[slot] == 0) goto unused;
            if (s->hash[slot] == h && s->key[slot] == k) return; /* already present */
            slot++;
        } while (probes--);
        perturb >>= 5;
        i = (i * 5 + 1 + perturb) & mask;
    }
unused:
    s->key[slot] = k; s->hash[slot] = h; s->fill++;
    if ((size_t)s->fill * 5 < mask * 3) return;
    {   /* set_table_resize(so, used*4) (used <= 50000) */
        size_t minused = (size_t)s->fill * 4, newsize = 8;
        while (newsize <= minused) newsize <<= 1;
        static _Thread_local int16_t ok[2048];
        static _Thread_local uint64_t oh[2048];
        size_t oldmask = mask;
        memcpy(ok, s->key, (oldmask + 1) * sizeof(int16_t));
        memcpy(oh, s->hash, (oldmask + 1) * sizeof(uint64_t));
        memset(s->key, 0, newsize * sizeof(int16_t));
        s->mask = (int)(newsize - 1);
        for (size_t e = 0; e <= oldmask; e++)
            if (ok[e]) py_set_insert_clean(s->key, s->hash, newsize - 1, ok[e], oh[e]);
    }
}

/* Exposed for the unit test that compares against real CPython sets. */
int azo_py_set_order(const int *rs, const int *cs, int n, int *out_idx) {
    py_set_t s; py_set_init(&s);
    /* keys are 1-based insertion indices here (so arbitrary (r,c) pairs can be tested) */
    for (int i = 0; i < n; i++) {
        int dup = -1;
        for (int j = 0; j < i; j++) if (rs[j] == rs[i] && cs[j] == cs[i]) { dup = j; break; }
        if (dup >= 0) continue;
        py_set_add(&s, (int16_t)(i + 1), azo_py_tuple2_hash(rs[i], cs[i]));
    }
    int m = 0;
    for (int e = 0; e <= s.mask; e++) if (s.key[e]) out_idx[m++] = s.key[e] - 1;
    return m;
}

/* ---------------------------------------------------------------------------------------
 * Board rules.  Boards are float32 [planes][rows][cols], cells tested == 0/1.
 * A move is a cell index r*cols + c.
 * ------------------------------------------------------------------------------------- */

/* get_valid_moves: tictactoe.py:82-83 (row-major empties); connect4.py:31-53 (per column with an
 * empty top cell: (lowest empty row, col)); gomoku.py:93-106 (empty 8-neighbours of any stone, as
 * list(set); centre if none).  Returns the count; cells_out in the reference's list order. */
int azo_get_valid_moves(const azo_game_t *g, const float *b, int *cells_out) {
    int R = g->rows, C = g->cols, n = 0;
    if (g->kind == AZO_TTT) {
        for (int cell = 0; cell < R * C; cell++) if (!occupied(g, b, cell)) cells_out[n++] = cell;
        return n;
    }
    if (g->kind == AZO_C4) {
        for (int col = 0; col < g->action_dim; col++) {
            if (occupied(g, b, col)) continue;             /* top cell (row 0) */
            for (int row = R - 1; row >= 0; row--)
                if (!occupied(g, b, row * C + col)) { cells_out[n++] = row * C + col; break; }
        }
        return n;
    }
    static const int DR[4] = {0, 1, 1, 1}, DC[4] = {1, 0, 1, -1};
    py_set_t s; py_set_init(&s);
    for (int r = 0; r < R; r++)
        for (int c = 0; c < C; c++) {
            if (!(b[r * C + c] == 1.0f || b[R * C + r * C + c] == 1.0f)) continue;
            for (int d = 0; d < 4; d++) {
                int r1 = r + DR[d], c1 = c + DC[d];
                if (r1 >= 0 && r1 < R && c1 >= 0 && c1 < C && !occupied(g, b, r1 * C + c1))
                    py_set_add(&s, (int16_t)(r1 * C + c1 + 1), azo_py_tuple2_hash(r1, c1));
                int r2 = r - DR[d], c2 = c - DC[d];
                if (r2 >= 0 && r2 < R && c2 >= 0 && c2 < C && !occupied(g, b, r2 * C + c2))
                    py_set_add(&s, (int16_t)(r2 * C + c2 + 1), azo_py_tuple2_hash(r2, c2));
            }
        }
    if (s.fill == 0) { cells_out[0] = (R / 2) * C + (C / 2); return 1; }
    for (int e = 0; e <= s.mask; e++) if (s.key[e]) cells_out[n++] = s.key[e] - 1;
    return n;
}

/* make_move: tictactoe.py:37-45, connect4.py:56-63, gomoku.py:51-58.  Returns the next player;
 * TicTacToe/Gomoku return `player` unchanged (and touch nothing) on an occupied cell; Connect4
 * never checks occupancy. */
int azo_make_move(const azo_game_t *g, float *b, int player, int cell) {
    int rc = g->rows * g->cols;
    if (g->kind != AZO_C4 && occupied(g, b, cell)) return player;
    b[player * rc + cell] = 1.0f;
    if (g->planes == 3) for (int i = 0; i < rc; i++) b[2 * rc + i] = (float)(1 - player);
    return 1 - player;
}

/* undo_move: tictactoe.py:48-51, connect4.py:67-70, gomoku.py:61-63 */
void azo_undo_move(const azo_game_t *g, float *b, int current_player, int cell) {
    int rc = g->rows * g->cols;
    b[(1 - current_player) * rc + cell] = 0.0f;
    if (g->planes == 3) for (int i = 0; i < rc; i++) b[2 * rc + i] = (float)(1 - current_player);
}

/* check_winner: tictactoe.py:54-79, connect4.py:73-98, gomoku.py:66-91.  The reference's stack walk
 * visits every contiguous `player` stone on both sides of (r,c) along a direction exactly once and
 * starts its counter at 1 WITHOUT testing the origin cell, i.e. it tests 1 + run(+) + run(-) >= K. */
int azo_check_winner(const azo_game_t *g, const float *b, int player, int cell) {
    static const int DR[4] = {0, 1, 1, 1}, DC[4] = {1, 0, 1, -1};
    int R = g->rows, C = g->cols, r0 = cell / C, c0 = cell % C;
    const float *pl = b + player * R * C;
    for (int d = 0; d < 4; d++) {
        int cnt = 1;
        for (int r = r0 + DR[d], c = c0 + DC[d]; r >= 0 && r < R && c >= 0 && c < C && pl[r * C + c] == 1.0f; r += DR[d], c += DC[d]) cnt++;
        for (int r = r0 - DR[d], c = c0 - DC[d]; r >= 0 && r < R && c >= 0 && c < C && pl[r * C + c] == 1.0f; r -= DR[d], c -= DC[d]) cnt++;
        if (cnt >= g->win_len) return player;
    }
    return -1;
}

/* get_canonical_board: gomoku.py:34-40; 3-plane form = ai/mcts.py:126-137 (commented history):
 * swap planes 0/1 for player 1, keep plane 2. */
void azo_canonical_board(const azo_game_t *g, const float *b, int player, float *out) {
    int rc = g->rows * g->cols;
    if (player == 0) { memcpy(out, b, sizeof(float) * g->planes * rc); return; }
    memcpy(out, b + rc, sizeof(float) * rc);
    memcpy(out + rc, b, sizeof(float) * rc);
    if (g->planes == 3) memcpy(out + 2 * rc, b + 2 * rc, sizeof(float) * rc);
}

/* ---------------------------------------------------------------------------------------
 * Deterministic float32 softmax shared bit-for-bit with the HIP engine.
 * The reference computes np.exp(l)/np.sum(np.exp(l)) in float32 without max subtraction
 * (ai/mcts.py:48-49).  numpy's SIMD expf is not reproducible on a GPU, so engine and oracle share
 * azo_exp_det: exp evaluated in float64 from correctly rounded primitives only (fma/mul/add, exact
 * 2^n scaling) and rounded once to float32 (|err| <= 0.5 ulp + 2^-40); the sum restates numpy's
 * float32 pairwise summation (numpy/core/src/umath/loops_utils.h.src pairwise_sum).  The oracle is
 * pinned against the reference with numpy's own softmax through the eval callback; this function
 * is pinned against numpy's softmax to <= 4 ulp (tests/test_oracle_softmax.py).
 * ------------------------------------------------------------------------------------- */
double azo_exp_det64(double x) {
    /* x = n*ln2 + r, |r| <= ln2/2 ; exp(r) by degree-13 Taylor in Horner form with explicit fma */
    const double LOG2E = 1.4426950408889634, LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    if (x > 700.0) return INFINITY;
    if (x < -740.0) return 0.0;
    double n = nearbyint(x * LOG2E);
    double r = fma(-n, LN2_HI, x);
    r = fma(-n, LN2_LO, r);
    double p = 1.0 / 6227020800.0;
    p = fma(p, r, 1.0 / 479001600.0);
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    /* scale by 2^n in two exact steps (covers the subnormal range of float32 after rounding) */
    int ni = (int)n, h = ni / 2;
    union { uint64_t u; double d; } s1, s2;
    s1.u = (uint64_t)(1023 + h) << 52;
    s2.u = (uint64_t)(1023 + (ni - h)) << 52;
    return p * s1.d * s2.d;
}

float azo_exp_det(float x) { return (float)azo_exp_det64((double)x); }

static float pairwise_sum_f32(const float *a, int n) {
    if (n < 8) {
        float res = 0.f;
        for (int i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= 128) {
        float r[8];
        int i;
        for (i = 0; i < 8; i++) r[i] = a[i];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        int n2 = n / 2;
        n2 -= n2 % 8;
        return pairwise_sum_f32(a, n2) + pairwise_sum_f32(a + n2, n - n2);
    }
}

float azo_pairwise_sum_f32(const float *a, int n) { return pairwise_sum_f32(a, n); }

void azo_softmax_det(const float *logits, int n, float *out) {
    float e[AZO_MAX_CELLS] = {0};
    for (int i = 0; i < n; i++) e[i] = azo_exp_det(logits[i]);
    float s = pairwise_sum_f32(e, n);
    for (int i = 0; i < n; i++) out[i] = e[i] / s;
}

/* ---------------------------------------------------------------------------------------
 * Tree (ai/node.py:21-40 mapped to arrays) and search (ai/mcts.py:11-60).
 * ------------------------------------------------------------------------------------- */
typedef int (*azo_eval_fn)(void *ctx, const float *canonical, float *priors_f32, float *value_f32);
typedef int (*azo_randint_fn)(void *ctx, int n);

typedef struct {
    long long mcts_count, matched, evals;                 /* ai/mcts.py:8-9,17,44 */
    long long edges_scanned, trace_nodes, edges_created, terminal_sims, expansions; /* SURVEY 8(d) */
} azo_counters_t;

typedef struct {
    int cap, n_nodes;
    long long *N;      /* Node.visit */
    double *W;         /* Node.value (running sum, from the parent's mover's perspective) */
    double *P;         /* Node.prior (float32 priors held exactly) */
    int16_t *cell;     /* Node.prevAction as r*cols+c ; -1 for the root */
    int *first_child;  /* index of children[0]; children are contiguous in list order */
    int *n_children;
    int root_player, root_move_count;
    int root_prior_f64; /* root children got Dirichlet-mixed float64 priors (utils.py:24-25) */
} azo_tree_t;

azo_tree_t *azo_tree_new(int cap) {
    azo_tree_t *t = calloc(1, sizeof *t);
    t->cap = cap < 16 ? 16 : cap;
    t->N = malloc(sizeof(long long) * t->cap); t->W = malloc(sizeof(double) * t->cap);
    t->P = malloc(sizeof(double) * t->cap); t->cell = malloc(sizeof(int16_t) * t->cap);
    t->first_child = malloc(sizeof(int) * t->cap); t->n_children = malloc(sizeof(int) * t->cap);
    return t;
}

void azo_tree_free(azo_tree_t *t) {
    if (!t) return;
    free(t->N); free(t->W); free(t->P); free(t->cell); free(t->first_child); free(t->n_children); free(t);
}

static void tree_grow(azo_tree_t *t, int need) {
    if (need <= t->cap) return;
    int cap = t->cap;
    while (cap < need) cap *= 2;
    t->N = realloc(t->N, sizeof(long long) * cap); t->W = realloc(t->W, sizeof(double) * cap);
    t->P = realloc(t->P, sizeof(double) * cap); t->cell = realloc(t->cell, sizeof(int16_t) * cap);
    t->first_child = realloc(t->first_child, sizeof(int) * cap); t->n_children = realloc(t->n_children, sizeof(int) * cap);
    t->cap = cap;
}

/* Node(None, None, current_player, move_count): gomoku.py:134 */
void azo_tree_reset(azo_tree_t *t, int player, int move_count) {
    t->n_nodes = 1; t->N[0] = 0; t->W[0] = 0.0; t->P[0] = 0.0; t->cell[0] = -1;
    t->first_child[0] = -1; t->n_children[0] = 0;
    t->root_player = player; t->root_move_count = move_count; t->root_prior_f64 = 0;
}

/* utils.py:29-44, network mode, float32 prior (numpy>=2 weak-scalar promotion => float32 math) */
static inline float ucb_f32(long long Nc, double Wc, double Pc, long long Np) {
    float s = (float)sqrt((double)Np);
    float u = ((float)Pc * s) / (float)(Nc + 1);
    if (Nc == 0) return u;
    return (float)(Wc / (double)Nc) + u;
}

/* same, float64 prior (root after utils.add_dirichlet_noise) */
static inline double ucb_f64(long long Nc, double Wc, double Pc, long long Np) {
    double u = Pc * sqrt((double)Np) / (double)(Nc + 1);
    if (Nc == 0) return u;
    return Wc / (double)Nc + u;
}

/* utils.py:29-44, mode == 'normal' (vanilla UCB1) */
static inline double ucb_vanilla(long long Nc, double Wc, long long Np) {
    double u = sqrt(2.0 * log((double)Np) / (double)(Nc + 1));
    if (Nc == 0) return u;
    return Wc / (double)Nc + u;
}

/* Node.select: node.py:42-47 - max() keeps the FIRST maximum in children order */
static int select_child(const azo_tree_t *t, int node, int network) {
    int fc = t->first_child[node], n = t->n_children[node], best = 0;
    long long Np = t->N[node];
    if (!network) {
        double bu = -INFINITY;
        for (int i = 0; i < n; i++) { double u = ucb_vanilla(t->N[fc + i], t->W[fc + i], Np); if (i == 0 || u > bu) { bu = u; best = i; } }
    } else if (node == 0 && t->root_prior_f64) {
        double bu = 0;
        for (int i = 0; i < n; i++) { double u = ucb_f64(t->N[fc + i], t->W[fc + i], t->P[fc + i], Np); if (i == 0 || u > bu) { bu = u; best = i; } }
    } else {
        float bu = 0;
        for (int i = 0; i < n; i++) { float u = ucb_f32(t->N[fc + i], t->W[fc + i], t->P[fc + i], Np); if (i == 0 || u > bu) { bu = u; best = i; } }
    }
    return fc + best;
}

/* eval cache: MCTS.cache (ai/mcts.py:7,38-44,51): key = canonical_board.tobytes() */
typedef struct {
    int key_bytes, action_dim, cap, used;
    uint8_t *keys; float *priors; float *values; uint8_t *full;
} azo_cache_t;

azo_cache_t *azo_cache_new(int key_bytes, int action_dim, int cap_pow2) {
    azo_cache_t *c = calloc(1, sizeof *c);
    c->key_bytes = key_bytes; c->action_dim = action_dim; c->cap = cap_pow2;
    c->keys = malloc((size_t)cap_pow2 * key_bytes); c->priors = malloc(sizeof(float) * (size_t)cap_pow2 * action_dim);
    c->values = malloc(sizeof(float) * cap_pow2); c->full = calloc(cap_pow2, 1);
    return c;
}
void azo_cache_free(azo_cache_t *c) { if (!c) return; free(c->keys); free(c->priors); free(c->values); free(c->full); free(c); }
void azo_cache_clear(azo_cache_t *c) { memset(c->full, 0, c->cap); c->used = 0; }
int azo_cache_size(const azo_cache_t *c) { return c->used; }

static uint64_t fnv1a(const uint8_t *p, int n) {
    uint64_t h = 1469598103934665603ULL;
    for (int i = 0; i < n; i++) { h ^= p[i]; h *= 1099511628211ULL; }
    return h;
}

static void cache_grow(azo_cache_t *c);

static long cache_find(azo_cache_t *c, const uint8_t *key, int *found) {
    size_t mask = (size_t)c->cap - 1, i = (size_t)fnv1a(key, c->key_bytes) & mask;
    while (c->full[i]) {
        if (!memcmp(c->keys + i * c->key_bytes, key, c->key_bytes)) { *found = 1; return (long)i; }
        i = (i + 1) & mask;
    }
    *found = 0;
    return (long)i;
}

static void cache_put(azo_cache_t *c, const uint8_t *key, const float *priors, float value) {
    if ((size_t)c->used * 2 >= (size_t)c->cap) cache_grow(c);
    int found; long i = cache_find(c, key, &found);
    if (found) return;
    c->full[i] = 1; c->used++;
    memcpy(c->keys + (size_t)i * c->key_bytes, key, c->key_bytes);
    memcpy(c->priors + (size_t)i * c->action_dim, priors, sizeof(float) * c->action_dim);
    c->values[i] = value;
}

static void cache_grow(azo_cache_t *c) {
    azo_cache_t old = *c;
    c->cap = old.cap * 2; c->used = 0;
    c->keys = malloc((size_t)c->cap * c->key_bytes); c->priors = malloc(sizeof(float) * (size_t)c->cap * c->action_dim);
    c->values = malloc(sizeof(float) * c->cap); c->full = calloc(c->cap, 1);
    for (int i = 0; i < old.cap; i++)
        if (old.full[i]) cache_put(c, old.keys + (size_t)i * old.key_bytes, old.priors + (size_t)i * old.action_dim, old.values[i]);
    free(old.keys); free(old.priors); free(old.values); free(old.full);
}

/* Node.backup: node.py:62-74 */
static void backup(const azo_game_t *g, azo_tree_t *t, const int *trace, int depth, double value, float *board, azo_counters_t *cnt) {
    for (int i = depth; i >= 0; i--) {
        int node = trace[i];
        t->N[node] += 1;
        t->W[node] += value;
        value = -value;
        if (i > 0) {
            /* node.currentPlayer = root player flipped i times */
            int cur = (t->root_player + i) & 1;
            azo_undo_move(g, board, cur, t->cell[node]);
        }
    }
    if (cnt) cnt->trace_nodes += depth + 1;
}

/* MCTS.simulate: mcts.py:62-79 (vanilla rollouts) */
static double simulate(const azo_game_t *g, const float *board, int node_player, int node_move_count,
                       azo_randint_fn rnd, void *rctx) {
    float sim[3 * AZO_MAX_CELLS];
    int moves[AZO_MAX_CELLS];
    memcpy(sim, board, sizeof(float) * g->planes * g->rows * g->cols);
    int cur = node_player, mc = node_move_count, winner = -1;
    while (winner == -1 && mc < g->state_dim) {
        int n = azo_get_valid_moves(g, sim, moves);
        int a = moves[rnd(rctx, n)];
        cur = azo_make_move(g, sim, cur, a);
        mc++;
        winner = azo_check_winner(g, sim, 1 - cur, a);
    }
    if (winner != -1) return winner == (1 - node_player) ? 1.0 : -1.0;
    return 0.0;
}

/* MCTS.mcts: mcts.py:11-60.  `noise` = the Dirichlet draw for this search (NULL: dirichlet=False);
 * `eval` NULL selects vanilla mode (model=None).  Returns 0, or <0 on callback failure. */
int azo_mcts(const azo_game_t *g, azo_tree_t *t, float *board, int n_iter,
             azo_eval_fn eval, void *ectx, const double *noise, azo_cache_t *cache,
             azo_randint_fn rnd, void *rctx, azo_counters_t *cnt) {
    int trace[AZO_MAX_CELLS + 2], moves[AZO_MAX_CELLS];
    float canon[3 * AZO_MAX_CELLS], priors[AZO_MAX_CELLS];
    int rc = g->rows * g->cols, network = eval != NULL;
    for (int it = 0; it < n_iter; it++) {
        if (cnt) cnt->mcts_count++;
        int node = 0, depth = 0;
        trace[0] = 0;
        while (t->n_children[node] > 0) {                                  /* mcts.py:20 */
            if (cnt) cnt->edges_scanned += t->n_children[node];
            int child = select_child(t, node, network);
            int mover = (t->root_player + depth) & 1;                     /* 1 - child.currentPlayer */
            node = child; depth++; trace[depth] = node;
            azo_make_move(g, board, mover, t->cell[node]);                /* mcts.py:23 */
        }
        int node_player = (t->root_player + depth) & 1, node_mc = t->root_move_count + depth;
        if (depth > 0) {                                                   /* mcts.py:25-32 */
            int w = azo_check_winner(g, board, 1 - node_player, t->cell[node]);
            if (w != -1) { if (cnt) cnt->terminal_sims++; backup(g, t, trace, depth, 1.0, board, cnt); continue; }
            if (node_mc == g->state_dim) { if (cnt) cnt->terminal_sims++; backup(g, t, trace, depth, 0.0, board, cnt); continue; }
        }
        int nv = azo_get_valid_moves(g, board, moves);                    /* mcts.py:34 */
        double result;
        int is_root = depth == 0;
        tree_grow(t, t->n_nodes + nv);
        int fc = t->n_nodes;
        if (network) {
            float value;
            azo_canonical_board(g, board, node_player, canon);            /* mcts.py:37 */
            int hit = 0;
            if (cache) {
                long slot = cache_find(cache, (const uint8_t *)canon, &hit);
                if (hit) { memcpy(priors, cache->priors + (size_t)slot * g->action_dim, sizeof(float) * g->action_dim); value = cache->values[slot]; if (cnt) cnt->matched++; }
            }
            if (!hit) {
                if (eval(ectx, canon, priors, &value) != 0) return -2;     /* mcts.py:46-49 (softmax in the callback) */
                if (cnt) cnt->evals++;
                if (cache) cache_put(cache, (const uint8_t *)canon, priors, value);
            }
            int mix = is_root && noise != NULL;                           /* mcts.py:42-43,52-53 */
            if (is_root) t->root_prior_f64 = mix;
            for (int i = 0; i < nv; i++) {                                 /* Node.expand: node.py:50-59 */
                int a = azo_action_idx(g, moves[i]);
                double p;
                if (mix) p = (double)(0.75f * priors[a]) + 0.25 * noise[a]; /* utils.py:24-25: f32*py-float -> f32; + f64 */
                else p = (double)priors[a];
                t->N[fc + i] = 0; t->W[fc + i] = 0.0; t->P[fc + i] = p; t->cell[fc + i] = (int16_t)moves[i];
                t->first_child[fc + i] = -1; t->n_children[fc + i] = 0;
            }
            result = -(double)value;                                       /* mcts.py:56 */
        } else {
            for (int i = 0; i < nv; i++) {
                t->N[fc + i] = 0; t->W[fc + i] = 0.0; t->P[fc + i] = 0.0; t->cell[fc + i] = (int16_t)moves[i];
                t->first_child[fc + i] = -1; t->n_children[fc + i] = 0;
            }
            result = simulate(g, board, node_player, node_mc, rnd, rctx);  /* mcts.py:58-59 */
        }
        t->first_child[node] = fc; t->n_children[node] = nv; t->n_nodes += nv;
        if (cnt) { cnt->edges_created += nv; cnt->expansions++; }
        backup(g, t, trace, depth, result, board, cnt);                   /* mcts.py:60 */
        (void)rc;
    }
    return 0;
}

/* ---------------------------------------------------------------------------------------
 * Virtual-loss expansion: the engine's OPT-IN mode (azk_config.leaves_per_step = K > 1; csrc/azk_engine.hip
 * k_tree<.., MULTI> with K slots) restated sequentially.  NOT reference behaviour - ai/mcts.py:16-60 is strictly
 * sequential, north_star asks for "virtual-loss expansion" on top of it - so there is no reference output to pin
 * this function to: it pins the KERNEL to a plain sequential statement of the schedule, built from the pinned
 * pieces above (select_child, the rules, the expansion of azo_mcts).
 *
 * Schedule.  A search runs "launches"; in a launch the game's K slots take turns k = 0 .. K-1, each:
 *   1. if the slot holds a pending leaf (selected in an earlier launch): evaluate it, expand it exactly as
 *      azo_mcts does (mcts.py:46-55) and back its value up along the recorded path.  The visit was counted at
 *      selection and the path carries a lost game, so the backup is W = (W + value_i) + 1, N unchanged;
 *   2. if the game has started fewer than n_sims simulations: start one - the PUCT walk of mcts.py:20-23 on the
 *      tree as it stands (virtual losses of the other slots included).  A walk that ends on a node whose
 *      expansion is pending (first_child == -2) gives up and does not count.  A terminal leaf (mcts.py:25-32) is
 *      backed up at once, normally.  Otherwise the leaf becomes the slot's pending leaf: every node of the path
 *      gets N += 1, W -= 1 (a visit and a lost game) and the leaf is marked first_child = -2.
 * The search ends after the launch in which n_sims simulations have been started and no slot is pending.
 * Returns the number of launches, or < 0 on callback failure.
 * ------------------------------------------------------------------------------------- */
typedef struct {
    int node, depth, nv, player;
    int trace[AZO_MAX_CELLS + 2], moves[AZO_MAX_CELLS];
    float canon[3 * AZO_MAX_CELLS];
} azo_vl_slot_t;

int azo_mcts_vl(const azo_game_t *g, azo_tree_t *t, float *board, int n_sims, int K,
                azo_eval_fn eval, void *ectx, const double *noise, azo_cache_t *cache, azo_counters_t *cnt) {
    if (K < 1 || K > 64 || !eval) return -1;
    azo_vl_slot_t *slots = calloc((size_t)K, sizeof *slots);
    float priors[AZO_MAX_CELLS];
    int started = 0, launches = 0, rc = 0;
    for (int k = 0; k < K; k++) slots[k].node = -1;
    for (;;) {
        launches++;
        for (int k = 0; k < K && rc == 0; k++) {
            azo_vl_slot_t *s = &slots[k];
            if (s->node >= 0) {                                            /* 1. expand + backup of the pending leaf */
                float value;
                int hit = 0;
                if (cache) {
                    long e = cache_find(cache, (const uint8_t *)s->canon, &hit);
                    if (hit) { memcpy(priors, cache->priors + (size_t)e * g->action_dim, sizeof(float) * g->action_dim); value = cache->values[e]; if (cnt) cnt->matched++; }
                }
                if (!hit) {
                    if (eval(ectx, s->canon, priors, &value) != 0) { rc = -2; break; }
                    if (cnt) cnt->evals++;
                    if (cache) cache_put(cache, (const uint8_t *)s->canon, priors, value);
                }
                int nv = s->nv, node = s->node;
                tree_grow(t, t->n_nodes + nv);
                int fc = t->n_nodes, mix = s->depth == 0 && noise != NULL;
                if (s->depth == 0) t->root_prior_f64 = mix;
                for (int i = 0; i < nv; i++) {                             /* Node.expand: node.py:50-59 */
                    int a = azo_action_idx(g, s->moves[i]);
                    double p = mix ? (double)(0.75f * priors[a]) + 0.25 * noise[a] : (double)priors[a];
                    t->N[fc + i] = 0; t->W[fc + i] = 0.0; t->P[fc + i] = p; t->cell[fc + i] = (int16_t)s->moves[i];
                    t->first_child[fc + i] = -1; t->n_children[fc + i] = 0;
                }
                t->first_child[node] = fc; t->n_children[node] = nv; t->n_nodes += nv;
                if (cnt) { cnt->edges_created += nv; cnt->expansions++; cnt->trace_nodes += s->depth + 1; }
                const double v = -(double)value;                           /* mcts.py:56 */
                for (int i = 0; i <= s->depth; i++) {
                    const double sv = ((s->depth - i) & 1) ? -v : v;
                    t->W[s->trace[i]] = (t->W[s->trace[i]] + sv) + 1.0;    /* the visit was counted at selection */
                }
                s->node = -1;
            }
            if (started >= n_sims) continue;                               /* 2. a new simulation */
            started++;
            int node = 0, depth = 0, trace[AZO_MAX_CELLS + 2];
            trace[0] = 0;
            while (t->n_children[node] > 0) {                              /* mcts.py:20-23 */
                if (cnt) cnt->edges_scanned += t->n_children[node];
                int child = select_child(t, node, 1);
                int mover = (t->root_player + depth) & 1;
                node = child; depth++; trace[depth] = node;
                azo_make_move(g, board, mover, t->cell[node]);
            }
            const int node_player = (t->root_player + depth) & 1, node_mc = t->root_move_count + depth;
            int keep = 0;
            if (t->first_child[node] == -2) {
                started--;                                                 /* expansion pending in another slot: no simulation */
            } else {
                if (cnt) cnt->mcts_count++;
                int term = -1;
                if (depth > 0) {                                           /* mcts.py:25-32 */
                    if (azo_check_winner(g, board, 1 - node_player, t->cell[node]) != -1) term = 1;
                    else if (node_mc == g->state_dim) term = 0;
                }
                if (term >= 0) {
                    if (cnt) { cnt->terminal_sims++; cnt->trace_nodes += depth + 1; }
                    double value = (double)term;
                    for (int i = depth; i >= 0; i--) { t->N[trace[i]] += 1; t->W[trace[i]] += value; value = -value; }
                } else keep = 1;
            }
            if (keep) {
                s->nv = azo_get_valid_moves(g, board, s->moves);           /* mcts.py:34 */
                azo_canonical_board(g, board, node_player, s->canon);      /* mcts.py:37 */
                for (int i = 0; i <= depth; i++) { t->N[trace[i]] += 1; t->W[trace[i]] -= 1.0; s->trace[i] = trace[i]; }
                t->first_child[node] = -2;
                s->node = node; s->depth = depth; s->player = node_player;
            }
            for (int i = depth; i > 0; i--) azo_undo_move(g, board, (t->root_player + i) & 1, t->cell[trace[i]]);   /* the caller's board comes back as it was */
        }
        if (rc != 0) break;
        int pending = 0;
        for (int k = 0; k < K; k++) pending |= slots[k].node >= 0;
        if (started >= n_sims && !pending) break;
        if (launches > 4 * n_sims + 16) { rc = -3; break; }                /* cannot happen: every launch starts or finishes a simulation */
    }
    free(slots);
    return rc != 0 ? rc : launches;
}

/* accessors */
int azo_tree_n_nodes(const azo_tree_t *t) { return t->n_nodes; }
long long azo_tree_root_visit(const azo_tree_t *t) { return t->N[0]; }
double azo_tree_root_value(const azo_tree_t *t) { return t->W[0]; }
int azo_tree_root_children(const azo_tree_t *t, int *cells, long long *visits, double *values, double *priors) {
    int fc = t->first_child[0], n = t->n_children[0];
    for (int i = 0; i < n; i++) { cells[i] = t->cell[fc + i]; visits[i] = t->N[fc + i]; values[i] = t->W[fc + i]; priors[i] = t->P[fc + i]; }
    return n;
}

/* DFS pre-order export (children in list order) for whole-tree digests */
int azo_tree_export(const azo_tree_t *t, int max, int *depth, int *cell, long long *visit, double *value, double *prior) {
    int *stack_n = malloc(sizeof(int) * (t->n_nodes + 1)), *stack_d = malloc(sizeof(int) * (t->n_nodes + 1));
    int sp = 0, m = 0;
    stack_n[sp] = 0; stack_d[sp++] = 0;
    while (sp) {
        int node = stack_n[--sp], d = stack_d[sp];
        if (m < max) { depth[m] = d; cell[m] = t->cell[node]; visit[m] = t->N[node]; value[m] = t->W[node]; prior[m] = t->P[node]; }
        m++;
        for (int i = t->n_children[node] - 1; i >= 0; i--) { stack_n[sp] = t->first_child[node] + i; stack_d[sp++] = d + 1; }
    }
    free(stack_n); free(stack_d);
    return m;
}

/* utils.get_probablity_distribution_of_children: utils.py:46-55 (float64; integer sums are exact) */
void azo_root_pi(const azo_game_t *g, const azo_tree_t *t, double *pi) {
    int fc = t->first_child[0], n = t->n_children[0];
    double sum = 0.0;
    for (int a = 0; a < g->action_dim; a++) pi[a] = 0.0;
    for (int i = 0; i < n; i++) pi[azo_action_idx(g, t->cell[fc + i])] = (double)t->N[fc + i];
    for (int a = 0; a < g->action_dim; a++) sum += pi[a];
    for (int a = 0; a < g->action_dim; a++) pi[a] = pi[a] / sum;
}

/* Node.max_visit_child: node.py:76-81 (first maximum in children order). Returns the child's cell. */
int azo_root_max_visit_cell(const azo_tree_t *t) {
    int fc = t->first_child[0], n = t->n_children[0], best = 0;
    for (int i = 1; i < n; i++) if (t->N[fc + i] > t->N[fc + best]) best = i;
    return t->cell[fc + best];
}

/* Node.sample_child: node.py:83-93 -> np.random.choice(children_by_action, p=pi) with one uniform u:
 * legacy RandomState.choice: cdf = p.cumsum(); cdf /= cdf[-1]; idx = cdf.searchsorted(u, 'right').
 * Returns the ACTION index drawn (the caller maps it to the child holding that action). */
int azo_sample_action(const double *pi, int n, double u) {
    double cdf[AZO_MAX_CELLS], acc = 0.0;
    for (int i = 0; i < n; i++) { acc += pi[i]; cdf[i] = acc; }
    double last = cdf[n - 1];
    for (int i = 0; i < n; i++) cdf[i] /= last;
    int lo = 0, hi = n;                      /* first i with cdf[i] > u */
    while (lo < hi) { int mid = (lo + hi) / 2; if (u < cdf[mid]) hi = mid; else lo = mid + 1; }
    return lo;
}

int azo_root_cell_for_action(const azo_game_t *g, const azo_tree_t *t, int action) {
    int fc = t->first_child[0], n = t->n_children[0];
    for (int i = 0; i < n; i++) if (azo_action_idx(g, t->cell[fc + i]) == action) return t->cell[fc + i];
    return -1;
}

void azo_counters_zero(azo_counters_t *c) { memset(c, 0, sizeof *c); }
