// azk_engine.hip - batched self-play engine for MI355X (gfx950): tree + board-rule kernels and the C ABI
// declared in include/azk.h.  One 64-lane wavefront (one 64-thread workgroup) owns one game; the tree is
// a structure-of-arrays arena in HBM whose child blocks are contiguous, so a PUCT scan is a coalesced read
// of the N / W / P columns; per-wave scratch (board, path, move list, emulated CPython set) lives in LDS.
// Built with -ffp-contract=off: every float result is the same sequence of IEEE operations the oracle runs.
// Reference lines cited as file:line relative to the reference root.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#include "azk.h"
#include "azk_device.h"

namespace {

enum { CNT_SIMS = 0, CNT_SCANNED, CNT_TRACE, CNT_CREATED, CNT_LEAVES, CNT_TERMINAL, CNT_MOVES, CNT_CACHE_HITS, CNT_N };

struct __attribute__((aligned(16))) NodeH { int N; float P; uint32_t meta; int fc; };
typedef float f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));   // four consecutive floats of a row that is only 4-byte aligned (A = 225: 900-byte rows)

struct Dev {               // device view of the engine, passed to kernels by value
    GameDesc g;
    int G, cap, path_cap, rc_pad, leaf_dtype, table_size, lds_bytes;
    int lds_off[12];       // lds_layout()'s offsets, computed once on the host (k_tree reads them instead of redoing the arithmetic in every wave)
    int K;                 // leaves in flight per game (virtual-loss mode, opt-in; 1 = the reference's sequential search).  Every
                           // pending-leaf array below is [G * K], slot v = g * K + k
    // game state
    uint8_t *cells;        // [G][rc_pad]  cell codes (1 = player 0, 2 = player 1)
    int *to_move, *move_count, *done, *winner;   // [G]
    // tree arena, [G][cap] each (ai/node.py:21-40 as columns)
    NodeH *H;              // per node, ONE 16-byte record: Node.visit N, Node.prior P (float32 softmax entry), meta = (Node.prevAction as
                           // r*cols+c) << 16 | len(Node.children) (cell 0xFFFF = root), fc = index of children[0] in this game's arena
                           // (-1 = not expanded): a PUCT candidate costs one 16-byte load + W instead of five column loads
    double *W;             // Node.value (running sum)
    double *rootP;         // [G][rc] float64 root priors after Dirichlet mixing (utils.py:24-25), by child position
    int *root_f64;         // [G] root children use rootP (float64 UCB) instead of P (float32 UCB)
    int *arena_top;        // [G] bump allocator
    // pending leaf of the current simulation
    int *leaf_node, *leaf_depth, *leaf_nmoves, *leaf_slot;   // [G]
    int *path;             // [G][path_cap]
    uint8_t *leaf_cells;   // [G][rc_pad] board at the leaf
    int16_t *leaf_moves;   // [G][rc] valid moves at the leaf, reference list order
    uint8_t *leaf_flag;    // [G * K] 1 = this slot contributes a leaf to the evaluator batch this step
    int *to_move_v;        // [G * K] to_move of the slot's game (K > 1: what the leaf hand-off kernels index by slot)
    const double *noise;   // [G][A] or nullptr (asynchronous moves: [G][2][A], see noise_sel)
    const long long *noise_sel;   // asynchronous moves: [G] the slot's move counter - its low bit selects the row of the game's CURRENT search; else nullptr
    // eval cache (MCTS.cache, ai/mcts.py:7,38-51): per-game direct-mapped table keyed by the exact canonical position
    int cache_entries, key_words;          // entries per game (power of two, 0 = off); 64-bit words per key
    unsigned long long *cache_key;         // [G][E][key_words] own-stone bit plane, opponent bit plane (+ side bit)
    float *cache_logits;                   // [G][E][A] the evaluator's logits row
    float *cache_value;                    // [G][E]
    int *leaf_cache;                       // [G] >= 0: pending leaf was a cache hit (entry index); < 0: miss, insert at -(x)-1
    // shared mode (one table for every game of the engine, like the reference's process-global MCTS.cache): entries are written
    // at EXPANSION by whichever game wins the entry's claim word for the current launch stamp, and read at selection only when
    // their claim stamp is older than the current launch (the kernel boundary is the only cross-CU ordering relied on); a hit is
    // copied into the game's own buffers at once, because another game may overwrite the entry before this game expands
    int cache_shared;
    unsigned long long cache_mask;         // shared: entries - 1 (entries = the largest power of two <= G * cache_entries)
    unsigned *cache_claim;                 // shared: [entries] launch stamp of the entry's last write, 0 = never written
    unsigned *cache_stamp;                 // shared: [1] stamp of the current launch (bumped by the leaf hand-off kernels)
    unsigned long long *leaf_key;          // shared: [G][key_words] key of the pending (missed) leaf
    float *hit_logits, *hit_value;         // shared: [G][A], [G] private copy of a hit
    int16_t *traj_action;  // [G][state_dim] cell played at each ply of the current game (square boards only, else null)
    double *traj_pi;       // [G][state_dim][A] visit distribution recorded at each ply
    long long *emit_base;  // [G] first tuple index (64-bit: the stream never wraps) of a game being emitted, -1 = not emitting
    int *sims_done;        // [G] simulations of the current search already run (budget stepping, azk_begin_search_budget)
    int *budget;           // [4] simulations per search, most simulations per game and launch, launch age (wall-clock ticks) up to which a
                           //     game may start another simulation (0: no limit), reserved
    long long *counters;   // [CNT_N][G]
    int *err;              // sticky error word
    int ablate;            // debug only (AZK_TREE_ABLATE): timing experiments that break parity on purpose
    long long *dbg;        // debug only: [G][8] cycle stamps per phase when ablate & 16
};

struct LdsView {
    uint8_t *board;
    int *path;
    int16_t *moves;
    float *e;
    int *cnt;
    double *cdf;
    MoveScratch ms;
};

__host__ __device__ inline int up16(int x) { return (x + 15) & ~15; }

__host__ __device__ inline int lds_layout(const GameDesc &g, int path_cap, int table_size, int *off) {
    // offsets (bytes) of: board, path, moves, e, cnt, cdf, bits, pref, ord, tabA, tabB, claim
    int o = 0;
    off[0] = o; o += up16(g.rc);
    off[1] = o; o += up16(path_cap * 4);
    off[2] = o; o += up16(g.rc * 2);
    int ea = g.action_dim > g.rc ? g.action_dim : g.rc;
    off[3] = o; o += up16(ea * 4);
    off[4] = o; o += up16(ea * 4);
    off[5] = o; o += up16(ea * 8);
    int nwords = (g.rc * 8 + 31) >> 5;
    off[6] = o; o += up16((nwords > (table_size >> 5) + 2 ? nwords : (table_size >> 5) + 2) * 4);   // key bitmap, later the set table's occupancy bitmap
    off[7] = o; o += up16(nwords * 2);
    off[8] = o; o += up16(g.rc * 2);
    off[9] = o; o += up16(table_size * 2);
    off[10] = o; o += up16(table_size * 2);
    off[11] = o; o += up16(table_size * 4);
    return o;
}

extern __shared__ __attribute__((aligned(16))) unsigned char azk_smem[];

__device__ __forceinline__ LdsView carve_at(const int *off, int table_size) {
    LdsView L;
    L.board = azk_smem + off[0];
    L.path = (int *)(azk_smem + off[1]);
    L.moves = (int16_t *)(azk_smem + off[2]);
    L.e = (float *)(azk_smem + off[3]);
    L.cnt = (int *)(azk_smem + off[4]);
    L.cdf = (double *)(azk_smem + off[5]);
    L.ms.bits = (uint32_t *)(azk_smem + off[6]);
    L.ms.pref = (uint16_t *)(azk_smem + off[7]);
    L.ms.ord = (int16_t *)(azk_smem + off[8]);
    L.ms.tabA = (uint16_t *)(azk_smem + off[9]);
    L.ms.tabB = (uint16_t *)(azk_smem + off[10]);
    L.ms.claim = (uint32_t *)(azk_smem + off[11]);
    L.ms.table_size = table_size;
    return L;
}

__device__ __forceinline__ LdsView carve(const GameDesc &g, int path_cap, int table_size) {
    int off[12];
    lds_layout(g, path_cap, table_size, off);
    LdsView L;
    L.board = azk_smem + off[0];
    L.path = (int *)(azk_smem + off[1]);
    L.moves = (int16_t *)(azk_smem + off[2]);
    L.e = (float *)(azk_smem + off[3]);
    L.cnt = (int *)(azk_smem + off[4]);
    L.cdf = (double *)(azk_smem + off[5]);
    L.ms.bits = (uint32_t *)(azk_smem + off[6]);
    L.ms.pref = (uint16_t *)(azk_smem + off[7]);
    L.ms.ord = (int16_t *)(azk_smem + off[8]);
    L.ms.tabA = (uint16_t *)(azk_smem + off[9]);
    L.ms.tabB = (uint16_t *)(azk_smem + off[10]);
    L.ms.claim = (uint32_t *)(azk_smem + off[11]);
    L.ms.table_size = table_size;
    return L;
}

__device__ __forceinline__ uint32_t meta_pack(int cell, int nch) { return ((uint32_t)(cell & 0xffff) << 16) | (uint32_t)nch; }
__device__ __forceinline__ int meta_cell(uint32_t m) { return (int)(m >> 16); }
__device__ __forceinline__ int meta_nch(uint32_t m) { return (int)(m & 0xffffu); }

// device counters: fire-and-forget atomics (no load -> add -> store round trip on the simulation's critical path)
__device__ __forceinline__ void count_add(const Dev &d, int which, int g, long long v) {
    atomicAdd((unsigned long long *)&d.counters[(size_t)which * d.G + g], (unsigned long long)v);
}

// Node.backup (node.py:62-74): the node at trace index i gets value * (-1)^(depth - i); lanes take one node each.
__device__ __forceinline__ void backup_path(const Dev &d, size_t base, const int *path, int depth, double value, bool undo_virtual_loss = false) {
    for (int i = azk_lane(); i <= depth; i += AZK_WAVE) {
        int nd = path[i];
        double sv = ((depth - i) & 1) ? -value : value;
        if (undo_virtual_loss) { d.W[base + nd] = (d.W[base + nd] + sv) + 1.0; continue; }      // the visit was counted at selection (same association as the short-path form)
        d.H[base + nd].N += 1;
        d.W[base + nd] += sv;
    }
}

// ================================================================================================
// k_tree<EXPAND, SELECT>: one simulation step for every game.
//   EXPAND: mcts.py:46-60 for the leaf selected by the previous step (softmax, noise, expand, backup)
//   SELECT: mcts.py:18-37 (PUCT walk, terminal test + backup, valid moves, leaf hand-off)
// ================================================================================================
// DBG: the AZK_TREE_ABLATE experiments and cycle stamps exist only in the <.., true> instantiation; the product kernel
// (DBG = false) carries none of their branches.
// MULTI: a game keeps simulating inside the launch for as long as its simulations need no evaluator - a terminal leaf is backed
// up at once (mcts.py:25-32) and a leaf served by the eval cache is expanded from the cached row (mcts.py:38-44) - and stops at
// the first leaf that misses the cache (one pending evaluation per game, as before), at its simulation budget, or after
// budget[1] simulations.  The order of a game's simulations is untouched (they are sequential inside one wave), so every tree
// is bit-identical to one-simulation-per-launch stepping; what changes is that ~55 % of the simulations no longer wait for a
// kernel boundary and every launch hands the evaluator a (nearly) full batch.
// KSL: cells (and actions) per lane the kernel is compiled for - 4 covers boards of up to 256 cells (every shipped game), 7 the rest
// (make_game allows 400).  A compile-time bound: the per-lane load sequences, their registers and the move generator's cell groups are
// unrolled to it, and a 15 x 15 board does not pay for three empty groups in each of them.
template <bool EXPAND, bool SELECT, bool DBG, bool MULTI = false, int KSL = 7>
__global__ __launch_bounds__(AZK_WAVE) void k_tree(Dev dd, const float *__restrict__ logits, const float *__restrict__ values) {
    const Dev &d = dd;
    const int ablate = DBG ? dd.ablate : 0;
    const int g = blockIdx.x;
    const int lane0 = azk_lane();
    const GameDesc &gd = d.g;
    const int A = gd.action_dim, rc = gd.rc;
    const size_t base = (size_t)g * (size_t)d.cap;
    LdsView L = carve_at(d.lds_off, d.table_size);
    const bool wrec = DBG && (ablate & 8192) != 0;      // debug only: ONE record per wave and launch (overwritten), for the distribution of wave times
    const bool stamp = (ablate & 16) != 0 || wrec;
    long long t0 = stamp ? clock64() : 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
    int rec_type = -1, rec_depth = 0, rec_nv = 0, rec_env = 0;   // wrec: -1 idle game, 0 terminal leaf, 1 eval-cache hit, 2 leaf for the evaluator
    int done_sims = MULTI ? d.sims_done[g] : 0;
    const int sim_target = MULTI ? d.budget[0] : 0, max_iter = MULTI ? d.budget[1] : 1;
    // a launch lasts as long as its slowest wave: a game whose simulation needed no evaluator starts another one only while the launch is
    // YOUNG (budget[2] ticks of the constant-rate clock; scalar, so the decision is wave-uniform) - a cheap simulation (terminal leaf) then
    // makes room for a second one, an expensive one does not push the wave past the launch's slowest.  Scheduling only: a game's
    // simulations stay in order, the trees do not change.
    const int young = MULTI ? d.budget[2] : 0;
    const long long t_launch = MULTI && young > 0 ? (long long)wall_clock64() : 0;
    auto still_young = [&]() { return young <= 0 || (long long)wall_clock64() - t_launch < (long long)young; };
    for (int it = 0; it < max_iter; it++) {
    // MULTI: the lane index is made opaque per iteration - left alone, the compiler hoists every lane-dependent address of the loop
    // body (the ~40 per-lane loads of a simulation) out of the loop and keeps them live across it: 294 VGPRs, one wave per SIMD,
    // the 2 048 waves of a launch in TWO rounds (measured 63 us for one simulation per launch against 36 us for the plain kernel)
    int lane = lane0;
    if (MULTI) asm volatile("" : "+v"(lane));
    if (MULTI && it > 0) __syncthreads();       // the previous simulation's LDS scratch is free and its tree / leaf writes are done
    // virtual-loss mode (K > 1, opt-in, changes search results): iteration k serves slot k - it expands the slot's pending leaf,
    // then selects a new one with a virtual loss left on its path so that the other slots' selections move elsewhere
    const bool vl = MULTI && d.K > 1;
    const int vi = g * d.K + (vl ? it : 0);

    // ---- every load whose address depends only on the game index is issued here, together: ONE memory round trip for
    //      the pending leaf's record, its path and move list, the game's state and board, and the root header ----
    const bool shared = d.cache_entries && d.cache_shared;
    int e_node, e_slot, e_depth, e_nv, e_top, e_centry, s_done, s_player, s_mc, s_rootf64, r_fc, r_N;
    uint32_t r_meta;
    unsigned cstamp;
    constexpr bool ONE_LOAD = EXPAND && SELECT && !MULTI;
    int uw = 0;
    if (ONE_LOAD) {
        // The fourteen per-game words below are uniform, and left to the compiler they become scalar loads issued in four
        // dependent groups (SGPR pressure) - four round trips before the first branch.  Here lane k fetches word k: ONE vector
        // load, first in the queue, and the words come back through v_readlane.
        const int *up = d.leaf_node + vi;                                 // lane 0 (and every lane without a word of its own)
        up = lane == 1 ? d.leaf_slot + vi : up;
        up = lane == 2 ? d.leaf_depth + vi : up;
        up = lane == 3 ? d.leaf_nmoves + vi : up;
        up = lane == 4 ? d.arena_top + g : up;
        up = (lane == 5 && d.cache_entries) ? d.leaf_cache + vi : up;
        up = (lane == 6 && shared) ? (const int *)d.cache_stamp : up;
        up = lane == 7 ? d.done + g : up;
        up = lane == 8 ? d.to_move + g : up;
        up = lane == 9 ? d.move_count + g : up;
        up = lane == 10 ? d.root_f64 + g : up;
        up = lane == 11 ? &d.H[base].fc : up;
        up = lane == 12 ? &d.H[base].N : up;
        up = lane == 13 ? (const int *)&d.H[base].meta : up;
        uw = *up;
    } else {
        e_node = EXPAND ? d.leaf_node[vi] : -1; e_slot = EXPAND ? d.leaf_slot[vi] : 0; e_depth = EXPAND ? d.leaf_depth[vi] : 0;
        e_nv = EXPAND ? d.leaf_nmoves[vi] : 0; e_top = EXPAND ? d.arena_top[g] : 0;
        e_centry = (EXPAND && d.cache_entries) ? d.leaf_cache[vi] : -1;
        cstamp = shared ? d.cache_stamp[0] : 0u;
        s_done = SELECT ? d.done[g] : 1; s_player = SELECT ? d.to_move[g] : 0; s_mc = SELECT ? d.move_count[g] : 0;
        s_rootf64 = SELECT ? d.root_f64[g] : 0;
        r_fc = SELECT ? d.H[base].fc : -1; r_N = SELECT ? d.H[base].N : 0;
        r_meta = SELECT ? d.H[base].meta : 0u;
    }
    unsigned long long e_key = 0ull;
    if (EXPAND && shared) e_key = d.leaf_key[(size_t)vi * d.key_words + min(lane, d.key_words - 1)];
    // (every per-lane load below is UNCONDITIONAL with a clamped index: a load under a lane predicate - `lane < n ? p[lane] : 0` -
    //  is compiled as a branch around the load plus a wait for its result right behind it, and the eighteen loads of this entry
    //  sequence then cost one memory round trip EACH instead of one together)
    const int e_path = EXPAND ? d.path[(size_t)vi * d.path_cap + min(lane, d.path_cap - 1)] : 0;     // trace nodes 0..63 (deeper ones: below)
    int e_mv[KSL] = {};
    if (EXPAND) {
#pragma unroll
        for (int k4 = 0; k4 < KSL; k4++) e_mv[k4] = d.leaf_moves[(size_t)vi * rc + min(lane + AZK_WAVE * k4, rc - 1)];
    }
    // the game's cell codes, four to a register: a row of `cells` is rc_pad bytes (a multiple of 16, zeros past rc), so the board comes in as
    // NCW dword loads per lane instead of KSL byte loads, and goes to LDS - and later out to leaf_cells - the same way
    constexpr int NCW = (KSL * AZK_WAVE / 4 + AZK_WAVE - 1) / AZK_WAVE;
    const int ncw = d.rc_pad >> 2;
    uint32_t s_cw[NCW] = {};
    if (SELECT) {
        const uint32_t *cw = (const uint32_t *)(d.cells + (size_t)g * d.rc_pad);
#pragma unroll
        for (int q = 0; q < NCW; q++) s_cw[q] = cw[min(lane + AZK_WAVE * q, ncw - 1)];
    }
    if (ONE_LOAD) {                                          // (behind the loads that do not depend on them)
        e_node = __builtin_amdgcn_readlane(uw, 0); e_slot = __builtin_amdgcn_readlane(uw, 1); e_depth = __builtin_amdgcn_readlane(uw, 2);
        e_nv = __builtin_amdgcn_readlane(uw, 3); e_top = __builtin_amdgcn_readlane(uw, 4);
        e_centry = d.cache_entries ? __builtin_amdgcn_readlane(uw, 5) : -1;
        cstamp = shared ? (unsigned)__builtin_amdgcn_readlane(uw, 6) : 0u;
        s_done = __builtin_amdgcn_readlane(uw, 7); s_player = __builtin_amdgcn_readlane(uw, 8); s_mc = __builtin_amdgcn_readlane(uw, 9);
        s_rootf64 = __builtin_amdgcn_readlane(uw, 10); r_fc = __builtin_amdgcn_readlane(uw, 11); r_N = __builtin_amdgcn_readlane(uw, 12);
        r_meta = (uint32_t)__builtin_amdgcn_readlane(uw, 13);
    }

    if (EXPAND) {
        const int node = uniform_i32(e_node);
        if (MULTI && node >= 0 && !(d.cache_entries && uniform_i32(e_centry) >= 0) && ((!vl && it > 0) || logits == nullptr)) {
            if (lane == 0) atomicExch(d.err, AZK_ERR_STATE);          // a pending evaluation without its logits: caller error
            break;
        }
        if (node >= 0) {
            const bool xst = (ablate & 1024) != 0;              // debug only: cycle stamps of the expansion's sub-phases
            long long x0 = 0, x1 = 0, x2 = 0, x3 = 0, x4 = 0;
            if (xst) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); x0 = clock64(); }
            const int slot = uniform_i32(e_slot);
            const int depth = uniform_i32(e_depth);
            const int nv = uniform_i32(e_nv);
            if (wrec) rec_env = nv;
            const int centry = d.cache_entries ? uniform_i32(e_centry) : -1;
            const bool hit = d.cache_entries && centry >= 0;
            const size_t crow = shared ? (size_t)(hit ? centry : -(centry + 1)) : ((size_t)g * d.cache_entries + (hit ? centry : -(centry + 1)));
            const float *lg = hit ? (shared ? d.hit_logits + (size_t)vi * A : d.cache_logits + crow * A) : logits + (size_t)slot * A;
            unsigned claim_now = 0u;
            if (shared && !hit) claim_now = __hip_atomic_load(d.cache_claim + crow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // Node.backup operands (trace nodes 0..depth, one per lane) are fetched now, next to the logits: second round trip
            const bool shortpath = depth < AZK_WAVE;
            // (unconditional, the lanes beyond the path read the root: a load under a lane predicate is followed by a wait for it,
            //  which put a whole round trip between these two loads and the logits below)
            const int bnode = (shortpath && lane <= depth) ? e_path : 0;
            const int bN = d.H[base + bnode].N;
            const double bW = d.W[base + bnode];
            // second (and last) round trip of the expansion, all straight-line: logits, value, the node's header, root noise
            // a lane takes FOUR consecutive logits per load (actions 4 lane .. 4 lane + 3 of each block of 256; the lane at the row's end
            // takes the row's last four, overlapping its neighbour - the same values twice): one 16-byte load instead of four 4-byte ones,
            // here and for the eval-cache row's stores and loads below
            constexpr int NV4 = (KSL * AZK_WAVE + 255) / 256;
            int la[NV4];                                              // first action of the lane's group
            bool lact[NV4];
            f32x4_a4 lgv[NV4];
#pragma unroll
            for (int q = 0; q < NV4; q++) {
                lact[q] = 256 * q + 4 * lane < A;
                la[q] = min(256 * q + 4 * lane, A - 4);
                lgv[q] = *(const f32x4_a4 *)(lg + la[q]);
            }
            const float vraw = hit ? (shared ? d.hit_value[vi] : d.cache_value[crow]) : values[slot];
            const uint32_t node_meta = d.H[base + node].meta;
            const bool mix = depth == 0 && d.noise != nullptr;        // mcts.py:42-43,52-53
            double nzv[KSL] = {};
            if (mix) {
                // asynchronous moves keep two rows per game - the current search's and the next one's, generated a whole search ahead
                // (k_noise_ahead) - and the slot's move counter says which is which
                const size_t nrow = (MULTI && d.noise_sel != nullptr) ? (size_t)g * 2 + (size_t)(uniform_i32((int)d.noise_sel[g]) & 1) : (size_t)g;
#pragma unroll
                for (int k4 = 0; k4 < KSL; k4++) {
                    const int i = lane + AZK_WAVE * k4;
                    nzv[k4] = d.noise[nrow * A + azk_action_idx(gd, i < nv ? e_mv[k4] : 0)];   // (a lane's e_mv beyond nv is stale memory)
                }
            }
            bool cache_write = d.cache_entries && !hit;               // MCTS.cache[board_key] = (...)  (mcts.py:51)
            if (shared && !hit) {
                // one writer per entry and launch: the claim word moves to this launch's stamp by compare-and-swap; an entry
                // already claimed in this launch (by any game) is left alone.  The round trip hides under the softmax below.
                const unsigned cur = (unsigned)uniform_i32((int)claim_now);
                unsigned got = cur;
                if (cur != cstamp && lane == 0) got = atomicCAS(d.cache_claim + crow, cur, cstamp);
                cache_write = cur != cstamp && (unsigned)uniform_i32((int)got) == cur;
                if (cache_write && lane < d.key_words) d.cache_key[crow * d.key_words + lane] = e_key;
            }
            if (xst) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); x1 = clock64(); }
            // float32 softmax, no max subtraction (mcts.py:48-49)
#pragma unroll
            for (int q = 0; q < NV4; q++) {
                if (256 * q >= A) break;
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const float ev = (ablate & 1) ? 1.0f : azk_exp_det(lgv[q][c]);
                    if (lact[q]) L.e[la[q] + c] = ev;
                }
            }
            azk_wave_sync();
            if (cache_write) {
                // (behind the exponentials: by now every logit is in its register, and the stores go out back to back - placed
                //  right behind the loads, each store waited for the one before it, one write round trip per 64 actions)
#pragma unroll
                for (int q = 0; q < NV4; q++) if (lact[q]) *(f32x4_a4 *)(d.cache_logits + crow * A + la[q]) = lgv[q];
                if (lane == 0) d.cache_value[crow] = vraw;
            }
            if (xst) x2 = clock64();
            const float s = azk_pairwise_sum(L.e, A);
            if (xst) x3 = clock64();
            const int fc = uniform_i32(e_top);
            const bool fits = fc + nv <= d.cap;
            if (fits) {
#pragma unroll
                for (int k4 = 0; k4 < KSL; k4++) {                    // Node.expand (node.py:50-59)
                    const int i = lane + AZK_WAVE * k4;
                    if (i >= nv) break;
                    const int cell = e_mv[k4];
                    const int a = azk_action_idx(gd, cell);
                    const float p = L.e[a] / s;
                    const size_t idx = base + fc + i;
                    d.H[idx] = NodeH{0, p, meta_pack(cell, 0), -1}; d.W[idx] = 0.0;
                    if (mix) d.rootP[(size_t)g * rc + i] = (double)(0.75f * p) + 0.25 * nzv[k4];   // utils.py:24-25
                }
                if (lane == 0) {
                    d.H[base + node].fc = fc;
                    d.H[base + node].meta = (node_meta & 0xffff0000u) | (uint32_t)nv;
                    d.arena_top[g] = fc + nv;
                    if (depth == 0) d.root_f64[g] = mix ? 1 : 0;
                    count_add(d, CNT_CREATED, g, nv);
                }
                if (node == 0) { r_fc = fc; r_meta = (r_meta & 0xffff0000u) | (uint32_t)nv; s_rootf64 = mix ? 1 : 0; }   // the root header loaded above is stale now
            } else if (lane == 0) {
                atomicExch(d.err, AZK_ERR_ARENA_FULL);
            }
            const double v = -(double)vraw;                          // mcts.py:56
            if (shortpath) {                                          // Node.backup (node.py:62-74) on the operands fetched above
                // virtual-loss mode: the visit was already counted at selection and the value carries the loss (-1) left there
                const int nN = vl ? bN : bN + 1;
                const double nW = bW + (((depth - lane) & 1) ? -v : v) + (vl ? 1.0 : 0.0);
                if (lane <= depth) {
                    d.H[base + e_path].N = nN;
                    d.W[base + e_path] = nW;
                }
            } else {
                backup_path(d, base, d.path + (size_t)vi * d.path_cap, depth, v, vl);
            }
            if (!vl) r_N += 1;                                        // the root is trace node 0 of every simulation
            if (xst && lane == 0) {
                x4 = clock64();
                long long *qq = d.dbg + (size_t)g * 8;
                qq[0] += x0 - t0; qq[1] += x1 - x0; qq[2] += x2 - x1; qq[3] += x3 - x2; qq[4] += x4 - x3; qq[6] += 1;
            }
            if (lane == 0) {
                d.leaf_node[vi] = -1;
                count_add(d, CNT_TRACE, g, depth + 1);
            }
        }
        azk_wave_sync();   // this wave's tree writes are visible to its own SELECT reads below
    }

    if (SELECT) {
        const bool active = uniform_i32(s_done) == 0;
        if (!active || (MULTI && done_sims >= sim_target)) {          // finished game / simulation budget of this search used up
            if (lane == 0 && (!MULTI || it == 0 || vl)) d.leaf_flag[vi] = 0;
            if (vl) continue;                                         // the other slots may still hold leaves to expand
            break;
        }
        done_sims++;
        if (stamp) t1 = clock64();
#pragma unroll
        for (int q = 0; q < NCW; q++) { const int i = lane + AZK_WAVE * q; if (i < ncw) ((uint32_t *)L.board)[i] = s_cw[q]; }
        const int root_player = uniform_i32(s_player);
        const int root_mc = uniform_i32(s_mc);
        if (lane == 0) L.path[0] = 0;
        azk_wave_sync();
        int node = 0, depth = 0, scanned = 0;
        // header of the current node, carried in registers: one dependent round trip per level (the child scan itself
        // brings every candidate's header along, and the winner's is taken from the winning lane)
        int fc = uniform_i32(r_fc);
        int Np = uniform_i32(r_N);
        uint32_t nmeta = (uint32_t)uniform_i32((int)r_meta);
        int node_cell = -1;
        const bool root_f64 = uniform_i32(s_rootf64) != 0;
        long long seg_a = 0, seg_b = 0, seg_c = 0, seg_d = 0, seg_t = 0;      // debug only (ablate & 64)
        for (;;) {                                                    // mcts.py:20-23
            const int nch = meta_nch(nmeta);
            if (nch <= 0 || (ablate & 2)) break;
            if (ablate & 64) seg_t = clock64();
            const bool f64 = node == 0 && root_f64;
            double bu64 = 0.0;
            float bu32 = 0.f;
            int best = 0x7fffffff, bN = 0, bfc = -1;
            uint32_t bmeta = 0;
            if (nch <= AZK_WAVE && !(ablate & 2048)) {
                // the common case (a Gomoku position has ~50 candidate moves): one candidate per lane, one load per column, the
                // argmax as a DPP maximum + ballot - "first maximum wins" (node.py:47) is the lowest lane holding the maximum
                const bool valid = lane < nch;
                const size_t ci = base + fc + (valid ? lane : 0);
                const NodeH hc = d.H[ci];                             // one 16-byte load: N, P, meta, first_child
                const double Wc = d.W[ci];
                const int Nc = hc.N, fcc = hc.fc;
                const uint32_t mc = hc.meta;
                const float P32 = hc.P;
                unsigned long long winners;
                if (f64) {                                            // root after Dirichlet mixing: float64 priors => float64 UCB
                    const double P64 = d.rootP[(size_t)g * rc + (valid ? lane : 0)];
                    const double s = sqrt((double)Np);
                    const double u0 = P64 * s / (double)(Nc + 1);
                    const double q = Wc / (double)Nc;                 // N = 0: inf/nan, discarded by the select
                    const double u = valid ? (Nc != 0 ? q + u0 : u0) : -__builtin_huge_val();
                    const double um = wave_max_f64(u);               // (all lanes take part: never under the short-circuit below)
                    winners = __ballot(valid & (u == um));
                } else {                                              // float32 priors => float32 UCB (numpy >= 2)
                    const float s = (float)sqrt((double)Np);
                    const float u0 = (P32 * s) / (float)(Nc + 1);
                    const float q = (float)(Wc / (double)Nc);
                    const float u = valid ? (Nc != 0 ? q + u0 : u0) : -__builtin_huge_valf();
                    const float um = wave_max_f32(u);
                    winners = __ballot(valid & (u == um));
                }
                best = __ffsll((long long)winners) - 1;
                bN = Nc; bmeta = mc; bfc = fcc;
            } else if (nch <= 2 * AZK_WAVE && !(ablate & 2048)) {
                // 65 .. 128 candidates (late plies: every node of the tree): two per lane, the same straight-line shape - ten loads,
                // one round trip.  (The general loop below sinks its prior loads into per-slot branches: one more round trip per
                // 64 candidates, on every level of a late-game walk.)  First maximum wins: indices below 64 before the others.
                const bool va = true, vb = lane + AZK_WAVE < nch;
                const size_t ca = base + fc + lane, cb = base + fc + (vb ? lane + AZK_WAVE : 0);
                const NodeH ha = d.H[ca], hb = d.H[cb];
                const double Wa = d.W[ca], Wb = d.W[cb];
                const int Na = ha.N, Nb = hb.N, fa = ha.fc, fb = hb.fc;
                const uint32_t ma = ha.meta, mb = hb.meta;
                const float Pa = ha.P, Pb = hb.P;
                unsigned long long wa, wb;
                if (f64) {
                    const double Qa = d.rootP[(size_t)g * rc + lane], Qb = d.rootP[(size_t)g * rc + (vb ? lane + AZK_WAVE : 0)];
                    const double s = sqrt((double)Np);
                    const double u0a = Qa * s / (double)(Na + 1), u0b = Qb * s / (double)(Nb + 1);
                    const double qa = Wa / (double)Na, qb = Wb / (double)Nb;
                    const double ua = Na != 0 ? qa + u0a : u0a;
                    const double ub = vb ? (Nb != 0 ? qb + u0b : u0b) : -__builtin_huge_val();
                    const double um = wave_max_f64(fmax(ua, ub));
                    wa = __ballot(va & (ua == um)); wb = __ballot(vb & (ub == um));
                } else {
                    const float s = (float)sqrt((double)Np);
                    const float u0a = (Pa * s) / (float)(Na + 1), u0b = (Pb * s) / (float)(Nb + 1);
                    const float qa = (float)(Wa / (double)Na), qb = (float)(Wb / (double)Nb);
                    const float ua = Na != 0 ? qa + u0a : u0a;
                    const float ub = vb ? (Nb != 0 ? qb + u0b : u0b) : -__builtin_huge_valf();
                    const float um = wave_max_f32(fmaxf(ua, ub));
                    wa = __ballot(va & (ua == um)); wb = __ballot(vb & (ub == um));
                }
                const bool first = wa != 0ull;
                best = first ? __ffsll((long long)wa) - 1 : AZK_WAVE + __ffsll((long long)wb) - 1;
                bN = first ? Na : Nb; bmeta = first ? ma : mb; bfc = first ? fa : fb;
            } else {
            // all of this level's loads are issued before any arithmetic: 4 candidates per lane per 256-child chunk
            for (int c0 = 0; c0 < nch; c0 += 4 * AZK_WAVE) {
                int Nc[4], fcc[4];
                double Wc[4], P64[4] = {0.0, 0.0, 0.0, 0.0};
                float P32[4];
                uint32_t mc[4];
                // straight-line loads, no per-slot control flow: a branch inside this loop makes the compiler wait for each
                // slot's prior before issuing the next slot (four serial round trips per level instead of one)
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int i = c0 + lane + AZK_WAVE * k;
                    const size_t ci = base + fc + (i < nch ? i : 0);
                    const NodeH hk = d.H[ci];
                    Nc[k] = hk.N; Wc[k] = d.W[ci]; mc[k] = hk.meta; fcc[k] = hk.fc; P32[k] = hk.P;
                }
                if (f64) {                                            // root after Dirichlet mixing: float64 priors by child position
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const int i = c0 + lane + AZK_WAVE * k;
                        P64[k] = d.rootP[(size_t)g * rc + (i < nch ? i : 0)];
                    }
                }
                if (ablate & 64) {   // debug only: time the level's memory round trip separately from its arithmetic
                    const long long ta = clock64();
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    seg_a += clock64() - ta;
                    seg_d += ta - seg_t;                              // (reusing seg_d: issue of the level's loads)
                }
                // branch-free on purpose: every `if` around a division or a compare chain becomes a saveexec/branch pair on
                // this target, and a level of the walk is a few hundred cycles of arithmetic buried under thousands of those
                if (f64) {                                            // float64 priors => float64 UCB
                    const double s = sqrt((double)Np);
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        if (c0 + AZK_WAVE * k >= nch) break;             // wave-uniform: no candidate in this slot at all
                        const int i = c0 + lane + AZK_WAVE * k;
                        const double u0 = P64[k] * s / (double)(Nc[k] + 1);
                        const double q = Wc[k] / (double)Nc[k];          // N = 0: inf/nan, discarded by the select below
                        const double u = Nc[k] != 0 ? q + u0 : u0;
                        const bool take = (i < nch) & ((best == 0x7fffffff) | (u > bu64));
                        bu64 = take ? u : bu64; best = take ? i : best; bN = take ? Nc[k] : bN;
                        bmeta = take ? mc[k] : bmeta; bfc = take ? fcc[k] : bfc;
                    }
                } else {                                              // float32 priors => float32 UCB (numpy>=2)
                    const float s = (float)sqrt((double)Np);
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        if (c0 + AZK_WAVE * k >= nch) break;             // wave-uniform: no candidate in this slot at all
                        const int i = c0 + lane + AZK_WAVE * k;
                        const float u0 = (P32[k] * s) / (float)(Nc[k] + 1);
                        const float q = (float)(Wc[k] / (double)Nc[k]);  // N = 0: inf/nan, discarded by the select below
                        const float u = Nc[k] != 0 ? q + u0 : u0;
                        const bool take = (i < nch) & ((best == 0x7fffffff) | (u > bu32));
                        bu32 = take ? u : bu32; best = take ? i : best; bN = take ? Nc[k] : bN;
                        bmeta = take ? mc[k] : bmeta; bfc = take ? fcc[k] : bfc;
                    }
                }
            }
            if (ablate & 64) { const long long tn = clock64(); seg_b += tn - seg_t; seg_t = tn; }
            if (f64) wave_argmax_first_lane63<double>(bu64, best);
            else wave_argmax_first_lane63<float>(bu32, best);
            best = __builtin_amdgcn_readlane(best, 63);               // DPP reduction: the wave's result lives in lane 63
            }
            const int wl = best & 63;                                 // the lane whose own best candidate won
            scanned += nch;
            const int child = fc + best;
            Np = __builtin_amdgcn_readlane(bN, wl);
            nmeta = (uint32_t)__builtin_amdgcn_readlane((int)bmeta, wl);
            fc = __builtin_amdgcn_readlane(bfc, wl);
            if (ablate & 64) { const long long tn = clock64(); seg_c += tn - seg_t; seg_t = tn; }
            const int cellc = meta_cell(nmeta);
            const int mover = (root_player + depth) & 1;
            depth++;
            node = child;
            node_cell = cellc;
            if (lane == 0) {
                L.path[depth] = node;
                // make_move (gomoku.py:51-58 / tictactoe.py:37-45 test emptiness; connect4.py:56-63 does not)
                if (gd.kind == AZK_KIND_C4) L.board[cellc] |= (uint8_t)(1 << mover);
                else if (L.board[cellc] == 0) L.board[cellc] = (uint8_t)(1 << mover);
            }
            if (depth + 1 >= d.path_cap) break;
        }
        if ((ablate & 64) && lane == 0) {
            long long *qq = d.dbg + (size_t)g * 8;
            qq[0] += seg_a; qq[1] += seg_b; qq[2] += seg_c; qq[3] += seg_d; qq[5] += depth; qq[6] += 1;
        }
        azk_wave_sync();
        if (stamp) t2 = clock64();
        if (vl && fc == -2) {                                         // the walk ended on a node another slot is already evaluating: no simulation
            done_sims--;
            if (lane == 0) d.leaf_flag[vi] = 0;
            continue;
        }
        const int node_player = (root_player + depth) & 1;
        const int node_mc = root_mc + depth;
        int term = -1;
        if (depth > 0) {                                              // mcts.py:25-32 (root is never tested)
            const int w = azk_check_winner(L.board, gd, 1 - node_player, node_cell);
            if (w != -1) term = 1;
            else if (node_mc == gd.state_dim) term = 0;
        }
        if (lane == 0) {
            count_add(d, CNT_SIMS, g, 1);
            count_add(d, CNT_SCANNED, g, scanned);
        }
        if (wrec) rec_depth = depth;
        if (term >= 0) {
            if (wrec) { rec_type = 0; t3 = t4 = clock64(); }
            backup_path(d, base, L.path, depth, (double)term);
            if (lane == 0) {
                d.leaf_flag[vi] = 0;
                count_add(d, CNT_TERMINAL, g, 1);
                count_add(d, CNT_TRACE, g, depth + 1);
            }
            if (MULTI && (vl || still_young())) continue;             // no evaluation needed: the next simulation starts at once
            break;
        }
        if (stamp) t3 = clock64();
        // ---- eval-cache probe (mcts.py:37-44: key = canonical board bytes), ISSUED HERE and looked at after the legal moves: the
        //      table sits in HBM, and its round trips (claim word + key + row, then the claim word again for a hit) pass under the
        //      move generation instead of behind it.  key = ballots of "own stone" / "opponent stone" over the cells (own = the
        //      side to move at the leaf)
        bool cached = false;
        int entry = 0;
        unsigned long long mykey = 0ull, kw = 0ull;
        unsigned c1v = 0u, c2v = 0u;
        constexpr int NV4P = (KSL * AZK_WAVE + 255) / 256;           // the cached row, four consecutive logits per lane and load (as at the expansion)
        f32x4_a4 row[NV4P] = {};
        float vv = 0.f;
        bool maybe_hit = false;                                       // shared table: key and claim word say "hit" - the second claim read decides
        const int KW = d.key_words;
        if (d.cache_entries) {
            const int half = KW >> 1;
            unsigned long long h = 0x9E3779B97F4A7C15ull;
            for (int q = 0; q < half; q++) {
                const int c = q * AZK_WAVE + lane;
                const uint8_t code = c < rc ? L.board[c] : (uint8_t)0;
                unsigned long long own = __ballot((code >> node_player) & 1);
                const unsigned long long opp = __ballot((code >> (node_player ^ 1)) & 1);
                if (q == half - 1) own |= (unsigned long long)node_player << 63;   // side to move (3-plane games; cell 63 of the last word is never a cell)
                if (lane == q) mykey = own;
                if (lane == half + q) mykey = opp;
                h = (h ^ own) * 0xFF51AFD7ED558CCDull; h ^= h >> 29;
                h = (h ^ opp) * 0xC4CEB9FE1A85EC53ull; h ^= h >> 32;
            }
            if (shared) {
                entry = (int)(h & d.cache_mask);
                // the claim word, the key and the entry's row (fetched on speculation: most probes miss, a row is 900 bytes) together
                c1v = __hip_atomic_load(d.cache_claim + entry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                kw = d.cache_key[(size_t)entry * KW + min(lane, KW - 1)];
#pragma unroll
                for (int q = 0; q < NV4P; q++) row[q] = *(const f32x4_a4 *)(d.cache_logits + (size_t)entry * A + min(256 * q + 4 * lane, A - 4));
                vv = d.cache_value[entry];
            } else {
                entry = (int)(h & (unsigned long long)(d.cache_entries - 1));
                kw = d.cache_key[((size_t)g * d.cache_entries + entry) * KW + min(lane, KW - 1)];
            }
        }
        // called by the move generator once its first phase is behind it (a few thousand cycles after the loads above went out):
        // the copy of the row is in registers before the claim word is read again
        auto probe_mid = [&]() {
            if (!(d.cache_entries && shared)) return;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned c1 = (unsigned)uniform_i32((int)c1v);
            const bool same = lane < KW ? kw == mykey : true;
            maybe_hit = c1 != 0u && c1 < cstamp && __ballot(!same) == 0ull;   // written in an earlier launch (complete and visible), same position
            if (maybe_hit) c2v = __hip_atomic_load(d.cache_claim + entry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        // (the second claim read is consumed inside each branch: a load still in flight where the branches meet makes the compiler
        //  wait for it wherever its register is reused - here that was the head of the Gomoku move generation, in front of
        //  everything the probe is meant to pass under)
        int nv;
        unsigned c2 = 0u;
        if (ablate & 4) { nv = 1; if (lane == 0) L.moves[0] = (int16_t)(gd.rc / 2); __syncthreads(); probe_mid(); c2 = (unsigned)uniform_i32((int)c2v); }
        else if (gd.kind == AZK_KIND_GOMOKU) {                         // mcts.py:34
            nv = azk_valid_moves_gomoku<KSL>(L.board, gd, L.moves, L.ms, (ablate & 8) != 0, (ablate & 32) ? d.dbg + (size_t)g * 8 : nullptr, probe_mid);
            c2 = (unsigned)uniform_i32((int)c2v);
        } else { probe_mid(); c2 = (unsigned)uniform_i32((int)c2v); nv = azk_valid_moves_small(L.board, gd, L.moves); }
        if (stamp) t4 = clock64();
        if (d.cache_entries) {
            if (shared) {
                if (maybe_hit && c2 == (unsigned)uniform_i32((int)c1v)) {   // nobody started rewriting the entry meanwhile: the copy is whole
                    cached = true;
#pragma unroll
                    for (int q = 0; q < NV4P; q++) if (256 * q + 4 * lane < A) *(f32x4_a4 *)(d.hit_logits + (size_t)vi * A + min(256 * q + 4 * lane, A - 4)) = row[q];
                    if (lane == 0) d.hit_value[vi] = vv;
                }
                if (!cached && lane < KW) d.leaf_key[(size_t)vi * KW + lane] = mykey;      // written into the table at expansion
            } else {
                unsigned long long *kp = d.cache_key + ((size_t)g * d.cache_entries + entry) * KW;
                const bool same = lane < KW ? kw == mykey : true;
                cached = __ballot(!same) == 0ull;
                if (!cached && lane < KW) kp[lane] = mykey;            // claim the slot now; logits/value land at expansion
            }
            if (lane == 0) d.leaf_cache[vi] = cached ? entry : -(entry + 1);
        }
        for (int i = lane; i < nv; i += AZK_WAVE) d.leaf_moves[(size_t)vi * rc + i] = L.moves[i];
        {
            uint32_t *lw = (uint32_t *)(d.leaf_cells + (size_t)vi * d.rc_pad);
#pragma unroll
            for (int q = 0; q < NCW; q++) { const int i = lane + AZK_WAVE * q; if (i < ncw) lw[i] = ((const uint32_t *)L.board)[i]; }
        }
        for (int i = lane; i <= depth; i += AZK_WAVE) d.path[(size_t)vi * d.path_cap + i] = L.path[i];
        if (vl) {                                                     // virtual loss: the path counts a visit now and a lost game until its value arrives
            for (int i = lane; i <= depth; i += AZK_WAVE) { const int nd = L.path[i]; d.H[base + nd].N += 1; d.W[base + nd] -= 1.0; }
            if (lane == 0) d.H[base + node].fc = -2;           // "expansion pending": a second slot arriving here gives up
        }
        if (lane == 0) {
            d.leaf_node[vi] = node; d.leaf_depth[vi] = depth; d.leaf_nmoves[vi] = nv;
            // 1 + cost class: the evaluator's embedding kernel works through the boards of a launch from the stone-heavy ones down
            // (a board's cost is the number of tokens a stone can reach: 0.93 correlated with its stone count)
            d.leaf_flag[vi] = cached ? 0 : (uint8_t)(1 + min(7, node_mc / 6));
            if (cached) count_add(d, CNT_CACHE_HITS, g, 1);
            count_add(d, CNT_LEAVES, g, cached ? 0 : 1);
            if (wrec) { rec_type = cached ? 1 : 2; rec_nv = nv; }
            if (stamp && !wrec) {
                long long *q = d.dbg + (size_t)g * 8;
                const long long tend = clock64();
                q[0] += t1 - t0; q[1] += t2 - t1; q[2] += t3 - t2; q[3] += t4 - t3; q[4] += tend - t4; q[5] += depth; q[6] += 1;
                if (!(ablate & 64) && tend - t0 > q[7]) q[7] = tend - t0;   // slowest simulation of this game
            }
        }
        if (MULTI && (vl || (cached && still_young()))) continue;     // served by the cache: expand it and go on, in this launch (vl: next slot)
    }
    if (vl) continue;
    break;
    }
    if (MULTI && lane0 == 0) d.sims_done[g] = done_sims;
    if (wrec) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the wave's own stores are out: the record covers its whole life
        if (lane0 == 0) {
            long long *q = d.dbg + (size_t)g * 8;
            const long long tend = clock64();
            if (rec_type < 0) { t1 = t2 = t3 = t4 = tend; }
            q[0] = t1 - t0; q[1] = t2 - t1; q[2] = t3 - t2; q[3] = t4 - t3; q[4] = tend - t4; q[5] = rec_depth;
            q[6] = (long long)(rec_type + 1) | ((long long)rec_nv << 8) | ((long long)rec_env << 20); q[7] = tend - t0;
        }
    }
}

// launch: the instantiation compiled for this engine's cells-per-lane bound
#define AZK_LAUNCH_TREE(E_, S_, D_, M_, ARGS_) do { \
        if (d.g.rc <= 4 * AZK_WAVE && d.g.action_dim <= 4 * AZK_WAVE) k_tree<E_, S_, D_, M_, 4><<<d.G, AZK_WAVE, d.lds_bytes, st>>> ARGS_; \
        else k_tree<E_, S_, D_, M_, 7><<<d.G, AZK_WAVE, d.lds_bytes, st>>> ARGS_; } while (0)

// Leaf compaction: slot = number of leaf games with a lower index (deterministic order); writes the
// canonical board (gomoku.py:34-40; 3-plane: mcts.py:126-137) of each leaf into the evaluator batch.
__global__ __launch_bounds__(AZK_WAVE) void k_gather(Dev d, void *__restrict__ leaf_boards, int *__restrict__ n_leaf_out) {
    const int g = blockIdx.x, lane = azk_lane();          // g = slot index over the G * K pending-leaf slots
    const int NV = d.G * d.K;
    // prefix over byte flags, 8 flags per lane per load (leaf_flag is padded to a multiple of 512 bytes)
    int before = 0, total = 0;
    const unsigned long long *fw = (const unsigned long long *)d.leaf_flag;
    const int nw = (NV + 7) >> 3;
    const bool last = g == NV - 1;
    const int limit_words = last ? nw : ((g + 8) >> 3);
    for (int w0 = 0; w0 < limit_words; w0 += AZK_WAVE) {
        int w = w0 + lane;
        unsigned long long x = w < nw ? fw[w] : 0ull;
        x = (((x & 0x7f7f7f7f7f7f7f7full) + 0x7f7f7f7f7f7f7f7full) | x) & 0x8080808080808080ull;   // one bit per non-zero flag byte (a flag carries its leaf's cost class)
        total += __popcll(x);
        // flags strictly before game g
        int lo = w * 8;
        if (lo + 8 <= g) before += __popcll(x);
        else if (lo < g) before += __popcll(x & ((1ull << ((g - lo) * 8)) - 1ull));
    }
    before = wave_sum_i32(before);
    if (last) {
        total = wave_sum_i32(total);
        if (lane == 0) {
            *n_leaf_out = total;
            if (d.cache_entries && d.cache_shared) d.cache_stamp[0] += 1u;   // the next tree launch may read what the last one cached
        }
    }
    if (!d.leaf_flag[g]) return;
    const int slot = before;
    if (lane == 0) d.leaf_slot[g] = slot;
    const int rc = d.g.rc, F = d.g.planes;
    const int player = (d.to_move[g / d.K] + d.leaf_depth[g]) & 1;     // node.currentPlayer
    const uint8_t *b = d.leaf_cells + (size_t)g * d.rc_pad;
    const size_t o = (size_t)slot * F * rc;
    for (int i = lane; i < F * rc; i += AZK_WAVE) {
        const int plane = i / rc, c = i - plane * rc;
        float v;
        if (plane == 2) v = (float)player;                            // side-to-move plane (tictactoe.py:41)
        else v = (float)((b[c] >> (plane ^ player)) & 1);             // own stones first for player 1
        if (d.leaf_dtype == AZK_LEAF_BF16) ((__hip_bfloat16 *)leaf_boards)[o + i] = __float2bfloat16(v);
        else ((float *)leaf_boards)[o + i] = v;
    }
}


// ================================================================================================
// Vanilla mode (model=None): mcts.py:57-59 (expand with no priors, rollout), MCTS.simulate mcts.py:62-79, UCB1 of
// utils.py:29-44 mode 'normal'.  No evaluator => a whole simulation (and n_sims of them) runs inside one launch.
// Random numbers: np.random.randint(len(valid_moves)) of the legacy global RandomState = MT19937 (randomkit) + numpy's
// masked rejection (random_bounded_uint64_fill, use_masked) - reproduced here on a per-game MT19937 state so that a
// search seeded with np.random.get_state() consumes the very same stream as the reference.
// ================================================================================================
__device__ void mt_twist(uint32_t *mt) {                          // all lanes; mt[624] in LDS (mt19937_gen)
    const int lane = azk_lane();
    auto phase = [&](int k0, int k1) {                            // every read of the phase happens before its writes
        uint32_t nv[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int k = k0 + lane + AZK_WAVE * q;
            if (k < k1) {
                const uint32_t y = (mt[k] & 0x80000000u) | (mt[k + 1] & 0x7fffffffu);
                const int src = k + 397 < 624 ? k + 397 : k - 227;
                nv[q] = mt[src] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int k = k0 + lane + AZK_WAVE * q;
            if (k < k1) mt[k] = nv[q];
        }
        __syncthreads();
    };
    phase(0, 227);        // sources mt[k+397]: old words
    phase(227, 454);      // sources mt[k-227] in [0, 227): already new
    phase(454, 623);      // sources in [227, 396): already new
    if (lane == 0) {
        const uint32_t y = (mt[623] & 0x80000000u) | (mt[0] & 0x7fffffffu);
        mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    __syncthreads();
}

__device__ __forceinline__ uint32_t mt_next(uint32_t *mt, int &pos) {   // wave-uniform
    if (pos >= 624) { mt_twist(mt); pos = 0; }
    uint32_t y = mt[pos++];
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
    return y;
}

// np.random.randint(n): range 0 draws nothing; otherwise 32-bit draws & (smallest 2^k - 1 >= n - 1) until <= n - 1
__device__ __forceinline__ int np_randint(uint32_t *mt, int &pos, int n) {
    if (n <= 1) return 0;
    const uint32_t rng = (uint32_t)(n - 1);
    uint32_t mask = rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    uint32_t v;
    do { v = mt_next(mt, pos) & mask; } while (v > rng);
    return (int)v;
}

__global__ __launch_bounds__(AZK_WAVE) void k_vanilla(Dev d, int n_sims, uint32_t *__restrict__ rng_state,
                                                       const double *__restrict__ lntab, int lntab_n) {
    const int g = blockIdx.x, lane = azk_lane();
    const GameDesc &gd = d.g;
    const int rc = gd.rc;
    const size_t base = (size_t)g * (size_t)d.cap;
    LdsView L = carve(gd, d.path_cap, d.table_size);
    uint32_t *mt = (uint32_t *)(azk_smem + d.lds_bytes);
    if (uniform_i32(d.done[g]) != 0) return;
    uint32_t *gs = rng_state + (size_t)g * 625;
    for (int i = lane; i < 624; i += AZK_WAVE) mt[i] = gs[i];
    int pos = uniform_i32((int)gs[624]);
    const int root_player = uniform_i32(d.to_move[g]), root_mc = uniform_i32(d.move_count[g]);
    __syncthreads();
    for (int sim = 0; sim < n_sims; sim++) {
        for (int i = lane; i < rc; i += AZK_WAVE) L.board[i] = d.cells[(size_t)g * d.rc_pad + i];
        if (lane == 0) L.path[0] = 0;
        __syncthreads();
        int node = 0, depth = 0, node_cell = -1, scanned = 0;
        int fc = uniform_i32(d.H[base].fc);
        int Np = uniform_i32(d.H[base].N);
        uint32_t nmeta = (uint32_t)uniform_i32((int)d.H[base].meta);
        for (;;) {                                                    // mcts.py:20-23 with node.select('normal')
            const int nch = meta_nch(nmeta);
            if (nch <= 0) break;
            if (Np < 1 || Np >= lntab_n) { if (lane == 0) atomicExch(d.err, AZK_ERR_STATE); return; }
            const double l2 = 2.0 * lntab[Np];                        // 2 * math.log(node.visit)
            double bu = 0.0;
            int best = 0x7fffffff, bN = 0, bfc = -1;
            uint32_t bmeta = 0;
            for (int i = lane; i < nch; i += AZK_WAVE) {
                const size_t ci = base + fc + i;
                const NodeH hc = d.H[ci];
                const int Nc = hc.N;
                const double Wc = d.W[ci];
                double u = sqrt(l2 / (double)(Nc + 1));               // utils.py:36,43
                if (Nc != 0) u = Wc / (double)Nc + u;
                if (best == 0x7fffffff || u > bu) { bu = u; best = i; bN = Nc; bmeta = hc.meta; bfc = hc.fc; }
            }
            wave_argmax_first<double>(bu, best);
            best = uniform_i32(best);
            const int wl = best & 63;
            scanned += nch;
            const int child = fc + best;
            Np = uniform_i32(__shfl(bN, wl)); nmeta = (uint32_t)uniform_i32(__shfl((int)bmeta, wl)); fc = uniform_i32(__shfl(bfc, wl));
            const int cellc = meta_cell(nmeta);
            const int mover = (root_player + depth) & 1;
            depth++;
            node = child;
            node_cell = cellc;
            if (lane == 0) {
                L.path[depth] = node;
                if (gd.kind == AZK_KIND_C4) L.board[cellc] |= (uint8_t)(1 << mover);
                else if (L.board[cellc] == 0) L.board[cellc] = (uint8_t)(1 << mover);
            }
            if (depth + 1 >= d.path_cap) break;
        }
        __syncthreads();
        const int node_player = (root_player + depth) & 1;
        const int node_mc = root_mc + depth;
        int term = -1;
        if (depth > 0) {                                              // mcts.py:25-32
            const int w = azk_check_winner(L.board, gd, 1 - node_player, node_cell);
            if (w != -1) term = 1;
            else if (node_mc == gd.state_dim) term = 0;
        }
        if (lane == 0) {
            d.counters[(size_t)CNT_SIMS * d.G + g] += 1;
            d.counters[(size_t)CNT_SCANNED * d.G + g] += scanned;
            d.counters[(size_t)CNT_TRACE * d.G + g] += depth + 1;
        }
        double result;
        if (term >= 0) {
            result = (double)term;
            if (lane == 0) d.counters[(size_t)CNT_TERMINAL * d.G + g] += 1;
        } else {
            const int nv = azk_valid_moves(L.board, gd, L.moves, L.ms);   // mcts.py:34
            const int afc = uniform_i32(d.arena_top[g]);
            if (afc + nv > d.cap) { if (lane == 0) atomicExch(d.err, AZK_ERR_ARENA_FULL); return; }
            for (int i = lane; i < nv; i += AZK_WAVE) {               // node.expand(valid_moves, None, Game): node.py:50-59
                const size_t idx = base + afc + i;
                d.H[idx] = NodeH{0, 0.f, meta_pack(L.moves[i], 0), -1}; d.W[idx] = 0.0;
            }
            if (lane == 0) {
                d.H[base + node].fc = afc;
                d.H[base + node].meta = (d.H[base + node].meta & 0xffff0000u) | (uint32_t)nv;
                d.arena_top[g] = afc + nv;
                d.counters[(size_t)CNT_CREATED * d.G + g] += nv;
            }
            // MCTS.simulate (mcts.py:62-79): the walk board is this simulation's private copy already
            int cur = node_player, mc = node_mc, winner = -1, n = nv;
            while (winner == -1 && mc < gd.state_dim) {
                if (mc != node_mc) n = azk_valid_moves(L.board, gd, L.moves, L.ms);   // first ply: the list computed above
                const int r = np_randint(mt, pos, n);
                const int cellc = uniform_i32((int)L.moves[r]);
                __syncthreads();
                if (lane == 0) {
                    if (gd.kind == AZK_KIND_C4) L.board[cellc] |= (uint8_t)(1 << cur);
                    else if (L.board[cellc] == 0) L.board[cellc] = (uint8_t)(1 << cur);
                }
                __syncthreads();
                winner = azk_check_winner(L.board, gd, cur, cellc);   // check_winner(sim_board, 1 - current_player, action)
                cur ^= 1;
                mc++;
            }
            result = winner != -1 ? (winner == (1 - node_player) ? 1.0 : -1.0) : 0.0;
        }
        backup_path(d, base, L.path, depth, result);
        __syncthreads();
    }
    for (int i = lane; i < 624; i += AZK_WAVE) gs[i] = mt[i];
    if (lane == 0) gs[624] = (uint32_t)pos;
}

// A leaf that missed the eval cache claims its entry's key at selection and fills logits/value at expansion; if the search
// is abandoned in between (new search, reset, recycle) the half-written entry must not survive.
__device__ __forceinline__ void drop_pending_cache_claim(const Dev &d, int g) {
    for (int k = 0; k < d.K; k++) {
        const int v = g * d.K + k;
        if (d.cache_entries && !d.cache_shared && d.leaf_node[v] >= 0 && d.leaf_cache[v] < 0) {      // (shared mode claims nothing at selection)
            unsigned long long *kp = d.cache_key + ((size_t)g * d.cache_entries + (size_t)(-(d.leaf_cache[v] + 1))) * d.key_words;
            for (int w = 0; w < d.key_words; w++) kp[w] = ~0ull;
        }
    }
}

// every pending-leaf slot of game g back to "nothing pending"
__device__ __forceinline__ void clear_leaf_slots(const Dev &d, int g) {
    for (int k = 0; k < d.K; k++) { d.leaf_node[g * d.K + k] = -1; d.leaf_flag[g * d.K + k] = 0; d.to_move_v[g * d.K + k] = d.to_move[g]; }
}

// Node(None, None, current_player, move_count) for every game (gomoku.py:134)
__global__ void k_begin_search(Dev d) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= d.G) return;
    if (g == 0 && d.cache_entries && d.cache_shared) d.cache_stamp[0] += 1u;
    drop_pending_cache_claim(d, g);
    const size_t base = (size_t)g * d.cap;
    d.H[base] = NodeH{0, 0.f, meta_pack(0xffff, 0), -1}; d.W[base] = 0.0;
    d.arena_top[g] = 1; d.root_f64[g] = 0;
    clear_leaf_slots(d, g);
    d.sims_done[g] = 0;
}

__global__ void k_reset_games(Dev d, int first, int count) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count * d.rc_pad) return;
    const int g = first + t / d.rc_pad, i = t % d.rc_pad;
    d.cells[(size_t)g * d.rc_pad + i] = 0;
    if (i == 0) drop_pending_cache_claim(d, g);
    if (i == 0) { d.to_move[g] = 0; d.move_count[g] = 0; d.done[g] = 0; d.winner[g] = -2; clear_leaf_slots(d, g); }
}

// Continuous self-play: every finished game's slot restarts from Game() (empty board, player 0).
// stats[0] += games recycled, stats[1] += plies those games lasted, stats[2..4] += wins of player 0 / player 1 / draws.
__global__ void k_recycle(Dev d, long long *stats) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= d.G || !d.done[g]) return;
    atomicAdd((unsigned long long *)&stats[0], 1ull);
    atomicAdd((unsigned long long *)&stats[1], (unsigned long long)d.move_count[g]);
    const int w = d.winner[g];
    atomicAdd((unsigned long long *)&stats[w == 0 ? 2 : (w == 1 ? 3 : 4)], 1ull);
    drop_pending_cache_claim(d, g);
    for (int i = 0; i < d.rc_pad; i++) d.cells[(size_t)g * d.rc_pad + i] = 0;
    d.to_move[g] = 0; d.move_count[g] = 0; d.done[g] = 0; d.winner[g] = -2;
    clear_leaf_slots(d, g);
}

// utils.get_probablity_distribution_of_children (utils.py:46-55), root.value / root.visit (gomoku.py:140)
__global__ __launch_bounds__(AZK_WAVE) void k_root_stats(Dev d, double *pi, double *q, int *root_visit) {
    const int g = blockIdx.x, lane = azk_lane();
    const size_t base = (size_t)g * d.cap;
    const int A = d.g.action_dim;
    LdsView L = carve(d.g, d.path_cap, d.table_size);
    const int fc = d.H[base].fc, nch = meta_nch(d.H[base].meta);
    for (int a = lane; a < A; a += AZK_WAVE) L.cnt[a] = 0;
    __syncthreads();
    int sum = 0;
    for (int i = lane; i < nch; i += AZK_WAVE) {
        const int n = d.H[base + fc + i].N;
        L.cnt[azk_action_idx(d.g, meta_cell(d.H[base + fc + i].meta))] = n;
        sum += n;
    }
    sum = wave_sum_i32(sum);
    __syncthreads();
    if (pi) for (int a = lane; a < A; a += AZK_WAVE) pi[(size_t)g * A + a] = (double)L.cnt[a] / (double)sum;
    if (lane == 0) {
        if (q) q[g] = d.W[base] / (double)d.H[base].N;
        if (root_visit) root_visit[g] = d.H[base].N;
    }
}

// gomoku.py:143-162 for one game (one wave): choose (sample ~ visits | first max-visit child), record pi / the action in the
// trajectory, make_move, check_winner, draw.  Returns the chosen cell (-1: state error, already reported); *win_out / *done_out as
// k_advance's outputs; pi of the move is left in L.cnt / sum_out (visit counts per action and their sum).
__device__ __forceinline__ int advance_one(const Dev &d, LdsView &L, int g, bool have_u, double u, int sample_until, int *win_out, int *done_out,
                                           int *sum_out) {
    const int lane = azk_lane();
    const GameDesc &gd = d.g;
    const size_t base = (size_t)g * d.cap;
    const int A = gd.action_dim, rc = gd.rc;
    const int fc = uniform_i32(d.H[base].fc), nch = uniform_i32(meta_nch(d.H[base].meta));
    const int mc = uniform_i32(d.move_count[g]), mover = uniform_i32(d.to_move[g]);
    for (int i = lane; i < rc; i += AZK_WAVE) L.board[i] = d.cells[(size_t)g * d.rc_pad + i];
    for (int a = lane; a < A; a += AZK_WAVE) L.cnt[a] = 0;
    __syncthreads();
    int sum = 0;
    for (int i = lane; i < nch; i += AZK_WAVE) {
        const int n = d.H[base + fc + i].N;
        L.cnt[azk_action_idx(gd, meta_cell(d.H[base + fc + i].meta))] = n;
        sum += n;
    }
    sum = wave_sum_i32(sum);
    __syncthreads();
    *sum_out = sum;
    int cellc = -1;
    if (nch <= 0 || sum <= 0) {
        if (lane == 0) atomicExch(d.err, AZK_ERR_STATE);
        return -1;
    }
    if (have_u && mc < sample_until) {
        // Node.sample_child (node.py:83-93) -> legacy np.random.choice(p=pi): cdf = cumsum(pi); cdf /= cdf[-1];
        // index = searchsorted(cdf, u, side='right').  cumsum is sequential in float64.
        if (lane == 0) {
            double acc = 0.0;
            for (int a = 0; a < A; a++) { acc += (double)L.cnt[a] / (double)sum; L.cdf[a] = acc; }
            const double lastv = L.cdf[A - 1];
            int lo = 0, hi = A;
            while (lo < hi) { int mid = (lo + hi) >> 1; if (u < L.cdf[mid] / lastv) hi = mid; else lo = mid + 1; }
            L.path[0] = lo < A ? lo : A - 1;                         // action drawn (path scratch: cnt[] is still needed)
        }
        __syncthreads();
        const int act = L.path[0];
        int found = 0x7fffffff;
        for (int i = lane; i < nch; i += AZK_WAVE)
            if (azk_action_idx(gd, meta_cell(d.H[base + fc + i].meta)) == act && i < found) found = i;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { int o = __shfl_xor(found, off); found = o < found ? o : found; }
        cellc = found != 0x7fffffff ? meta_cell(d.H[base + fc + found].meta) : -1;
    } else {
        // Node.max_visit_child (node.py:76-81): first child with the most visits
        int best = 0x7fffffff, bn = 0;
        for (int i = lane; i < nch; i += AZK_WAVE) {
            const int n = d.H[base + fc + i].N;
            if (best == 0x7fffffff || n > bn) { bn = n; best = i; }
        }
        wave_argmax_first<int>(bn, best);
        cellc = meta_cell(d.H[base + fc + uniform_i32(best)].meta);
    }
    cellc = uniform_i32(cellc);
    if (cellc < 0) {
        if (lane == 0) atomicExch(d.err, AZK_ERR_STATE);
        return -1;
    }
    if (d.traj_pi != nullptr && mc < gd.state_dim) {               // gomoku.py:138-146: pi and the action of this ply
        double *tp = d.traj_pi + ((size_t)g * gd.state_dim + mc) * A;
        for (int a = lane; a < A; a += AZK_WAVE) tp[a] = (double)L.cnt[a] / (double)sum;
        if (lane == 0) d.traj_action[(size_t)g * gd.state_dim + mc] = (int16_t)cellc;
    }
    if (lane == 0) {
        if (gd.kind == AZK_KIND_C4) L.board[cellc] |= (uint8_t)(1 << mover);
        else if (L.board[cellc] == 0) L.board[cellc] = (uint8_t)(1 << mover);
    }
    __syncthreads();
    const int w = azk_check_winner(L.board, gd, mover, cellc);      // gomoku.py:150
    int win = -2, dn = 0;
    if (w != -1) { win = w; dn = 1; }
    else if (mc + 1 == gd.state_dim) { win = -1; dn = 1; }
    if (lane == 0) {
        d.cells[(size_t)g * d.rc_pad + cellc] = L.board[cellc];
        d.to_move[g] = 1 - mover;
        d.move_count[g] = mc + 1;
        d.winner[g] = win; d.done[g] = dn;
        d.counters[(size_t)CNT_MOVES * d.G + g] += 1;
    }
    *win_out = win; *done_out = dn;
    return cellc;
}

__global__ __launch_bounds__(AZK_WAVE) void k_advance(Dev d, const double *uniforms, int sample_until,
                                                       int *chosen, int *winner_out, int *done_out) {
    const int g = blockIdx.x, lane = azk_lane();
    LdsView L = carve(d.g, d.path_cap, d.table_size);
    if (uniform_i32(d.done[g]) != 0) {
        if (lane == 0) {
            if (chosen) chosen[g] = -1;
            if (winner_out) winner_out[g] = d.winner[g];
            if (done_out) done_out[g] = 1;
        }
        return;
    }
    int win = -2, dn = 0, sum = 0;
    const int cellc = advance_one(d, L, g, uniforms != nullptr, uniforms != nullptr ? uniforms[g] : 0.0, sample_until, &win, &dn, &sum);
    if (cellc < 0) return;
    if (lane == 0) {
        if (chosen) chosen[g] = cellc;
        if (winner_out) winner_out[g] = win;
        if (done_out) done_out[g] = dn;
    }
}

// ------------------------------------------------------------------------------------------------
// (state, pi, z) emission: train.save_data_to_buffer (train.py:30-49) with rotate_data / flip_data (train.py:8-27).
// Position i of a finished game (side to move = i & 1): z = +-1 by winner (0 for a draw), state = canonical board;
// positions 0 and 1 once, the others 8 times in the order rot0, lr(rot0), tb(rot0), rot90, lr(rot90), tb(rot90),
// rot180, rot270 (np.rot90 is counter-clockwise).  Tuple t of the stream lands in slot t % capacity (deque(maxlen)).
// ------------------------------------------------------------------------------------------------
__global__ void k_emit_alloc(Dev d, unsigned long long *cursor, long long *game_base_out) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= d.G) return;
    long long base = -1;                                          // the stream index stays 64-bit end to end (2^31 tuples = hours of self-play)
    if (d.done[g] == 1) {
        const int n = d.move_count[g];
        const int tuples = n <= 2 ? n : 2 + 8 * (n - 2);
        base = (long long)atomicAdd(cursor, (unsigned long long)tuples);
    }
    d.emit_base[g] = base;
    if (game_base_out) game_base_out[g] = base;
}

__device__ __forceinline__ int d4_source(int t, int i, int j, int N) {
    // source cell (row-major) of output cell (i, j) under transform t of the reference's emission order
    int si, sj;
    switch (t) {
        case 0: si = i; sj = j; break;                          // rot0
        case 1: si = i; sj = N - 1 - j; break;                  // lr(rot0)
        case 2: si = N - 1 - i; sj = j; break;                  // tb(rot0)
        case 3: si = j; sj = N - 1 - i; break;                  // rot90 (ccw): out[i][j] = in[j][N-1-i]
        case 4: si = N - 1 - j; sj = N - 1 - i; break;          // lr(rot90)
        case 5: si = j; sj = i; break;                          // tb(rot90)
        case 6: si = N - 1 - i; sj = N - 1 - j; break;          // rot180
        default: si = N - 1 - j; sj = i; break;                 // rot270: out[i][j] = in[N-1-j][i]
    }
    return si * N + sj;
}

template <bool LIST>     // LIST: the grid walks a list of finished games (asynchronous drain) instead of covering all G
__global__ __launch_bounds__(AZK_WAVE) void k_emit_tuples(Dev d, float *states, double *pis, float *zs, long long capacity,
                                                            const unsigned long long *cursor, const int *list, const int *n_list) {
    const int S = d.g.state_dim, A = d.g.action_dim, rc = d.g.rc, F = d.g.planes, N = d.g.rows;
    const int lane = azk_lane();
    const int nb = LIST ? *n_list * S : (int)gridDim.x;
    for (int blk = blockIdx.x; blk < nb; blk += gridDim.x) {
    if (LIST && blk != (int)blockIdx.x) __syncthreads();
    const int gi = blk / S, i = blk - gi * S;
    const int g = LIST ? list[gi] : gi;
    const long long base = d.emit_base[g];
    if (base < 0 || i >= d.move_count[g]) continue;
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    uint8_t *cells = sm;                                          // board before ply i
    for (int c = lane; c < rc; c += AZK_WAVE) cells[c] = 0;
    __syncthreads();
    const int16_t *acts = d.traj_action + (size_t)g * S;
    for (int j = lane; j < i; j += AZK_WAVE) cells[acts[j]] = (uint8_t)(1 << (j & 1));
    __syncthreads();
    const int side = i & 1, winner = d.winner[g];
    const float z = winner == -1 ? 0.0f : (side == winner ? 1.0f : -1.0f);
    const double *pi = d.traj_pi + ((size_t)g * S + i) * A;
    const int ntr = i < 2 ? 1 : 8;
    const long long first = base + (i < 2 ? i : 2 + 8 * (i - 2));
    const long long stream_end = (long long)*cursor;              // after k_emit_alloc: one past the newest tuple of this call
    for (int t = 0; t < ntr; t++) {
        if (first + t < stream_end - capacity) continue;          // already pushed out of the ring by newer tuples (deque(maxlen))
        const long long slot = (first + t) % capacity;
        float *so = states + (size_t)slot * F * rc;
        double *po = pis + (size_t)slot * A;
        for (int e = lane; e < rc; e += AZK_WAVE) {
            const int src = d4_source(t, e / N, e % N, N);
            const uint8_t code = cells[src];
            so[e] = (float)((code >> side) & 1);                    // canonical: own stones first (gomoku.py:34-40)
            so[rc + e] = (float)((code >> (side ^ 1)) & 1);
            if (F == 3) so[2 * rc + e] = (float)side;
            po[e] = pi[src];                                        // square boards: action index == cell index
        }
        if (lane == 0) zs[slot] = z;
    }
    }
}

__global__ void k_emit_mark(Dev d) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < d.G && d.done[g] == 1 && d.emit_base[g] >= 0) d.done[g] = 2;      // emitted; recycle / later calls skip it
}

__global__ void k_sum_counters(const long long *counters, int G, long long *out) {
    // one block per counter
    __shared__ long long sm[256];
    long long s = 0;
    for (int i = threadIdx.x; i < G; i += blockDim.x) s += counters[(size_t)blockIdx.x * G + i];
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) out[blockIdx.x] = sm[0];
}

// ------------------------------------------------------------------------------------------------
// Counter-based RNG for the product path: Philox4x32-10 keyed by (seed), counter = (game, move, lane idx, draw).
// Dirichlet(alpha) via Gamma(alpha) = Gamma(alpha + 1) * U^(1/alpha) (Marsaglia-Tsang for the shape > 1 part).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0];
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

__device__ __forceinline__ double u53(uint32_t hi, uint32_t lo) {     // uniform in (0, 1)
    const unsigned long long x = (((unsigned long long)hi << 32) | lo) >> 11;
    return ((double)x + 0.5) * (1.0 / 9007199254740992.0);
}

// the uniform of (seed, global game, move): np.random.choice's draw of that move
__device__ __forceinline__ double noise_uniform(unsigned long long seed, unsigned long long gg, int move) {
    uint32_t c[4] = {(uint32_t)gg, (uint32_t)(gg >> 32), (uint32_t)move, 0xFFFFFFFFu};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    return (double)((((unsigned long long)c[0] << 32) | c[1]) >> 11) * (1.0 / 9007199254740992.0);   // [0,1)
}

// the Dirichlet(alpha) row of (seed, global game, move), by one wave; red: 64 doubles of LDS scratch
__device__ __forceinline__ void noise_row(int A, unsigned long long seed, unsigned long long gg, int move, double alpha, double *row, double *red) {
    const int lane = azk_lane();
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    double part = 0.0;
    for (int a = lane; a < A; a += AZK_WAVE) {
        const double d = alpha + 1.0 - 1.0 / 3.0, cc = 1.0 / sqrt(9.0 * d);
        double gam = 0.0;
        for (uint32_t it = 0; it < 64; it++) {
            uint32_t c[4] = {(uint32_t)gg, (uint32_t)(gg >> 32) ^ ((uint32_t)a << 8), (uint32_t)move, it};
            philox4x32_10(c, k0, k1);
            uint32_t c2[4] = {(uint32_t)gg, (uint32_t)(gg >> 32) ^ ((uint32_t)a << 8), (uint32_t)move, it | 0x40000000u};
            philox4x32_10(c2, k0, k1);
            const double u1 = u53(c[0], c[1]), u2 = u53(c[2], c[3]), u3 = u53(c2[0], c2[1]), u4 = u53(c2[2], c2[3]);
            const double x = sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
            const double t = 1.0 + cc * x;
            if (t <= 0.0) continue;
            const double v = t * t * t;
            if (log(u3) < 0.5 * x * x + d - d * v + d * log(v)) { gam = d * v * pow(u4, 1.0 / alpha); break; }
        }
        row[a] = gam;
        part += gam;
    }
    red[lane] = part;
    __syncthreads();
    for (int o = 32; o > 0; o >>= 1) { if (lane < o) red[lane] += red[lane + o]; __syncthreads(); }
    const double tot = red[0];
    for (int a = lane; a < A; a += AZK_WAVE) row[a] = tot > 0.0 ? row[a] / tot : 1.0 / (double)A;
    __syncthreads();
}

__global__ __launch_bounds__(AZK_WAVE) void k_gen_noise(int A, unsigned long long seed, long long first_game, int move,
                                                         double alpha, double *noise, double *uniforms, long long row_stride) {
    const int g = blockIdx.x, lane = azk_lane();
    const unsigned long long gg = (unsigned long long)(first_game + g);
    __shared__ double red[AZK_WAVE];
    if (uniforms && lane == 0) uniforms[g] = noise_uniform(seed, gg, move);
    if (!noise) return;
    noise_row(A, seed, gg, move, alpha, noise + (size_t)g * (size_t)row_stride, red);
}

// ------------------------------------------------------------------------------------------------
// Asynchronous self-play (games/gomoku.py:132-162: a game moves as soon as ITS search is done).  After every tree launch
// k_move_async looks at every game: one whose search is complete (its simulation budget used up, nothing pending) gets its move
// - root statistics, pi into the trajectory, sampled / most-visited child, make_move, check_winner, a record into the device
// ring - and, unless the game ended, its next search at once: fresh root, Dirichlet row of (seed, global game, the slot's move
// counter).  Finished games wait for azk_async_drain ((state, pi, z) emission, statistics, restart), which the host runs every
// few launches.  Random numbers are keyed by (seed, global game index, per-slot move counter): exactly the keys of the lock-step
// driver, so a slot plays the same sequence of games move for move, whatever the timing.
// ------------------------------------------------------------------------------------------------
struct AsyncDev {
    int n_sims, sample_until, dirichlet;
    unsigned long long seed;
    long long first_game;
    double alpha;
    long long *slot_moves;     // [G] moves this slot has played since azk_async_begin (all its games): the RNG's move key
    double *noise;             // [G][2][A] engine-owned: row (k & 1) of game g is the Dirichlet row of its search with move key k, for the
                               //   current key (slot_moves[g]) and the next one - generated a whole search ahead of its use
    int *noise_key;            // [G] the highest move key whose row exists
    int *todo_list, *todo_count;   // games that moved since the last drain: their row for key slot_moves[g] + 1 is due (k_noise_ahead)
    long long *stats;          // caller's int64 [16]: games, plies, wins 0 / 1, draws, moves, record cursor, searches begun
    long long rec_cap;
    int *rec_meta; double *rec_q; double *rec_pi;
    int *fin_list, *fin_count; // games found finished by the drain
};

// Node(None, None, player, move_count) for one game (one wave).  The search's Dirichlet row is NOT made here: a row's key (seed, global
// game, slot move counter) is known a whole search before its use, so the rows are generated one search ahead, off the step's chain
// (k_noise_ahead in the drain) - in this function the Marsaglia-Tsang chain (float64 log / cos / pow, two Philox blocks per try) cost a
// moving game's wave ~30 us inside a launch every other wave had left after 1 us.
__device__ __forceinline__ void begin_search_one(const Dev &d, const AsyncDev &p, int g, double *red) {
    const int lane = azk_lane();
    if (lane == 0) {
        drop_pending_cache_claim(d, g);
        const size_t base = (size_t)g * d.cap;
        d.H[base] = NodeH{0, 0.f, meta_pack(0xffff, 0), -1}; d.W[base] = 0.0;
        d.arena_top[g] = 1; d.root_f64[g] = 0;
        clear_leaf_slots(d, g);
        d.sims_done[g] = 0;
        atomicAdd((unsigned long long *)&p.stats[7], 1ull);
    }
    (void)red;
}

// drain: the Dirichlet rows that fell due since the last drain - for every game that moved, the row of the search AFTER the one it has
// just begun (key slot_moves[g] + 1, into the buffer the finished search read from)
__global__ __launch_bounds__(AZK_WAVE) void k_noise_ahead(Dev d, AsyncDev p) {
    __shared__ double red[AZK_WAVE];
    const int n = *p.todo_count, A = d.g.action_dim;
    for (int f = blockIdx.x; f < n; f += gridDim.x) {
        const int g = p.todo_list[f];
        const int key = (int)p.slot_moves[g] + 1;
        if (uniform_i32(p.noise_key[g]) >= key) continue;
        noise_row(A, p.seed, (unsigned long long)(p.first_game + g), key, p.alpha, p.noise + ((size_t)g * 2 + (size_t)(key & 1)) * A, red);
        if (azk_lane() == 0) p.noise_key[g] = key;
        __syncthreads();
    }
}

__global__ void k_fill_i32(int *p, int n, int v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

__global__ __launch_bounds__(AZK_WAVE) void k_move_async(Dev d, AsyncDev p) {
    const int g = blockIdx.x, lane = azk_lane();
    // one vector load for the three words that decide whether this game moves now (almost never: the wave then ends at once)
    const int *up = d.done + g;
    up = lane == 1 ? d.sims_done + g : up;
    up = lane == 2 ? d.leaf_node + g : up;
    up = lane == 3 ? d.budget : up;                                // (the simulation budget lives in device memory: azk_async_set_budget)
    up = (lane == 4 && p.dirichlet) ? p.noise_key + g : up;
    up = lane == 5 ? (const int *)(p.slot_moves + g) : up;         // (low word: a slot plays far fewer than 2^31 moves)
    const int uw = *up;
    if (__builtin_amdgcn_readlane(uw, 0) != 0 || __builtin_amdgcn_readlane(uw, 1) < __builtin_amdgcn_readlane(uw, 3) || __builtin_amdgcn_readlane(uw, 2) >= 0) return;
    // the next search's Dirichlet row is made a search ahead (k_noise_ahead, every drain); a game whose whole search fitted between two
    // drains (tiny budgets only) waits for it - a scheduling delay, the game's moves do not change
    if (p.dirichlet && __builtin_amdgcn_readlane(uw, 4) < __builtin_amdgcn_readlane(uw, 5) + 1) return;
    LdsView L = carve(d.g, d.path_cap, d.table_size);
    const int A = d.g.action_dim;
    const size_t base = (size_t)g * d.cap;
    const long long mv = p.slot_moves[g];
    const double q = d.W[base] / (double)d.H[base].N;                 // root.value / root.visit (gomoku.py:140), before the tree is reset
    int win = -2, dn = 0, sum = 0;
    const double u = noise_uniform(p.seed, (unsigned long long)(p.first_game + g), (int)mv);
    const int cellc = advance_one(d, L, g, true, u, p.sample_until, &win, &dn, &sum);
    if (cellc < 0) return;
    if (p.rec_cap > 0) {                                           // the move's record: what the reference's self_play keeps per ply
        long long slot = 0;
        if (lane == 0) slot = (long long)(atomicAdd((unsigned long long *)&p.stats[6], 1ull) % (unsigned long long)p.rec_cap);
        slot = ((long long)__builtin_amdgcn_readfirstlane((int)(slot >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)slot);
        for (int a = lane; a < A; a += AZK_WAVE) p.rec_pi[(size_t)slot * A + a] = (double)L.cnt[a] / (double)sum;
        if (lane == 0) {
            p.rec_q[slot] = q;
            int *m = p.rec_meta + (size_t)slot * 4;
            m[0] = g; m[1] = (int)mv; m[2] = cellc; m[3] = win;
        }
    }
    __syncthreads();
    if (lane == 0) {
        p.slot_moves[g] = mv + 1;                                  // the NEW search's key: its row (mv + 1) & 1 has been waiting since the last move
        atomicAdd((unsigned long long *)&p.stats[5], 1ull);
        if (p.dirichlet) p.todo_list[atomicAdd(p.todo_count, 1)] = g;          // row mv + 2 is due
    }
    __syncthreads();
    if (!dn) begin_search_one(d, p, g, L.cdf);
}

// drain, step 1: list the finished games (done == 1) - the emission and restart kernels work through the list only
__global__ void k_async_list(Dev d, AsyncDev p) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < d.G && d.done[g] != 0) p.fin_list[atomicAdd(p.fin_count, 1)] = g;
}

// drain, step 3: statistics + Game() + the next search for every listed game
__global__ __launch_bounds__(AZK_WAVE) void k_async_restart(Dev d, AsyncDev p, int recycle) {
    const int lane = azk_lane();
    LdsView L = carve(d.g, d.path_cap, d.table_size);
    const int n = *p.fin_count;
    for (int f = blockIdx.x; f < n; f += gridDim.x) {
        const int g = p.fin_list[f];
        if (uniform_i32(d.done[g]) == 3) continue;                 // already counted by an earlier drain (recycle off: the slot stays finished)
        if (lane == 0) {
            atomicAdd((unsigned long long *)&p.stats[0], 1ull);
            atomicAdd((unsigned long long *)&p.stats[1], (unsigned long long)d.move_count[g]);
            const int w = d.winner[g];
            atomicAdd((unsigned long long *)&p.stats[w == 0 ? 2 : (w == 1 ? 3 : 4)], 1ull);
        }
        if (!recycle) { if (lane == 0) d.done[g] = 3; continue; }
        for (int i = lane; i < d.rc_pad; i += AZK_WAVE) d.cells[(size_t)g * d.rc_pad + i] = 0;
        if (lane == 0) { d.to_move[g] = 0; d.move_count[g] = 0; d.done[g] = 0; d.winner[g] = -2; }
        __syncthreads();
        begin_search_one(d, p, g, L.cdf);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// stateless rule kernels over float32 boards [n][F][R][C] (the reference's own board layout)
// ------------------------------------------------------------------------------------------------
enum { RULE_MOVES = 0, RULE_MASK, RULE_APPLY, RULE_UNDO, RULE_WINNER, RULE_CANON };

struct RuleArgs {
    GameDesc g;
    int mode, n, table_size;
    const float *boards_in; float *boards;
    const int *players; const int *cells;
    int16_t *moves; int *counts; uint8_t *mask; int *out_i; float *out_f;
};

__device__ __forceinline__ uint8_t code_of(float p0, float p1) {
    uint8_t c = (p0 == 1.0f ? 1 : 0) | (p1 == 1.0f ? 2 : 0);
    if ((p0 != 0.0f && p0 != 1.0f) || (p1 != 0.0f && p1 != 1.0f)) c |= 4;
    return c;
}

__global__ __launch_bounds__(AZK_WAVE) void k_rules(RuleArgs a) {
    const int b = blockIdx.x, lane = azk_lane();
    const GameDesc &g = a.g;
    const int rc = g.rc, F = g.planes;
    LdsView L = carve(g, 4, a.table_size);
    const float *src = (a.boards_in ? a.boards_in : a.boards) + (size_t)b * F * rc;
    for (int i = lane; i < rc; i += AZK_WAVE) L.board[i] = code_of(src[i], src[rc + i]);
    __syncthreads();
    if (a.mode == RULE_MOVES || a.mode == RULE_MASK) {
        const int n = azk_valid_moves(L.board, g, L.moves, L.ms);
        if (a.mode == RULE_MOVES) {
            for (int i = lane; i < n; i += AZK_WAVE) a.moves[(size_t)b * rc + i] = L.moves[i];
            if (lane == 0) a.counts[b] = n;
        } else {
            for (int i = lane; i < g.action_dim; i += AZK_WAVE) a.mask[(size_t)b * g.action_dim + i] = 0;
            __syncthreads();
            for (int i = lane; i < n; i += AZK_WAVE) a.mask[(size_t)b * g.action_dim + azk_action_idx(g, L.moves[i])] = 1;
        }
    } else if (a.mode == RULE_APPLY) {
        const int player = a.players[b], cell = a.cells[b];
        float *dst = a.boards + (size_t)b * F * rc;
        int next = player;
        if (g.kind == AZK_KIND_C4 || L.board[cell] == 0) {
            next = 1 - player;
            if (lane == 0) dst[(size_t)player * rc + cell] = 1.0f;
            if (F == 3) for (int i = lane; i < rc; i += AZK_WAVE) dst[2 * rc + i] = (float)(1 - player);
        }
        if (lane == 0) a.out_i[b] = next;
    } else if (a.mode == RULE_UNDO) {
        const int cur = a.players[b], cell = a.cells[b];
        float *dst = a.boards + (size_t)b * F * rc;
        if (lane == 0) dst[(size_t)(1 - cur) * rc + cell] = 0.0f;
        if (F == 3) for (int i = lane; i < rc; i += AZK_WAVE) dst[2 * rc + i] = (float)(1 - cur);
    } else if (a.mode == RULE_WINNER) {
        const int w = azk_check_winner(L.board, g, a.players[b], a.cells[b]);
        if (lane == 0) a.out_i[b] = w;
    } else if (a.mode == RULE_CANON) {
        const int player = a.players[b];
        float *dst = a.out_f + (size_t)b * F * rc;
        for (int i = lane; i < F * rc; i += AZK_WAVE) {
            const int plane = i / rc, c = i - plane * rc;
            const int sp = plane < 2 ? (plane ^ player) : plane;
            dst[i] = src[(size_t)sp * rc + c];
        }
    }
}

__global__ __launch_bounds__(AZK_WAVE) void k_softmax_rows(const float *logits, int A, float *out) {
    const int b = blockIdx.x, lane = azk_lane();
    float *e = (float *)azk_smem;
    for (int i = lane; i < A; i += AZK_WAVE) e[i] = azk_exp_det(logits[(size_t)b * A + i]);
    __syncthreads();
    const float s = azk_pairwise_sum(e, A);
    for (int i = lane; i < A; i += AZK_WAVE) out[(size_t)b * A + i] = e[i] / s;
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
thread_local std::string g_create_error;

bool make_game(int kind, int rows, int cols, GameDesc *g, std::string *err) {
    memset(g, 0, sizeof *g);
    g->kind = kind;
    if (kind == AZK_TICTACTOE) { rows = 3; cols = 3; g->planes = 3; g->win_len = 3; g->action_dim = 9; }
    else if (kind == AZK_CONNECT4) { rows = 6; cols = 7; g->planes = 3; g->win_len = 4; g->action_dim = 7; }
    else if (kind == AZK_GOMOKU) {
        // (<= 30 columns: azk_valid_moves_gomoku shifts the board's bit string by up to cols + 1 inside 64-bit words)
        if (rows < 1 || cols < 1 || rows * cols > 400 || cols > 30) { *err = "gomoku board must have 1..400 cells and at most 30 columns"; return false; }
        g->planes = 2; g->win_len = 5; g->action_dim = rows * cols;
    } else { *err = "unknown game id"; return false; }
    g->rows = rows; g->cols = cols; g->rc = rows * cols; g->state_dim = rows * cols;
    g->inv_cols = (65536u + (unsigned)cols - 1u) / (unsigned)cols;
    return true;
}

int table_size_for(const GameDesc &g) { return g.rc < 307 ? 512 : 2048; }   // CPython set growth: 8 -> 32 -> 128 -> 512 -> 2048

}  // namespace

struct azk_engine {
    Dev d;
    azk_config cfg;
    std::string err;
    std::vector<void *> allocs;
    long long *counter_sums = nullptr;   // device [CNT_N]
    int *n_leaf_scratch = nullptr;
    void *leaf_scratch = nullptr;        // used when the caller passes no leaf buffer
    uint32_t *vanilla_rng = nullptr;     // [G][625] MT19937 key + position (vanilla mode), allocated on first use
    double *lntab = nullptr;             // [lntab_n] math.log(N), from the host libm (the reference's math.log)
    int lntab_n = 0;
    bool multi = false;                  // budget stepping (azk_begin_search_budget): the MULTI instantiation of k_tree
    int budget_host[4] = {0, 1, 0, 0};
    int ticks_per_us = 100;              // constant-rate clock of wall_clock64()
    AsyncDev ad;                         // asynchronous self-play (azk_async_begin); ad.slot_moves == nullptr: not set up
    bool async_on = false;
    int async_recycle = 1;
};

#define HIPCHK(e, call)                                                                 \
    do {                                                                                \
        hipError_t _s = (call);                                                         \
        if (_s != hipSuccess) {                                                         \
            (e)->err = std::string(#call) + ": " + hipGetErrorString(_s);              \
            return AZK_ERR_HIP;                                                         \
        }                                                                               \
    } while (0)

template <typename T>
static hipError_t dalloc(azk_engine *e, T **p, size_t count) {
    void *q = nullptr;
    hipError_t s = hipMalloc(&q, count * sizeof(T) + 64);
    if (s != hipSuccess) return s;
    e->allocs.push_back(q);
    *p = (T *)q;
    return hipSuccess;
}

extern "C" {

int32_t azk_abi_version(void) { return AZK_ABI_VERSION; }

const char *azk_last_error(const azk_engine *e) { return e ? e->err.c_str() : g_create_error.c_str(); }

int32_t azk_create(const azk_config *cfg, azk_engine **out) {
    if (!cfg || !out) { g_create_error = "null argument"; return AZK_ERR_ARG; }
    *out = nullptr;
    azk_engine *e = new azk_engine();
    e->cfg = *cfg;
    Dev &d = e->d;
    memset(&d, 0, sizeof d);
    memset(&e->ad, 0, sizeof e->ad);
    auto fail = [&](int code, const std::string &msg) { g_create_error = msg; azk_destroy(e); return code; };
    std::string gerr;
    if (!make_game(cfg->game, cfg->rows, cfg->cols, &d.g, &gerr)) return fail(AZK_ERR_ARG, gerr);
    if (cfg->n_games < 1 || cfg->max_sims < 1) return fail(AZK_ERR_ARG, "n_games and max_sims must be >= 1");
    if (cfg->leaf_dtype != AZK_LEAF_F32 && cfg->leaf_dtype != AZK_LEAF_BF16) return fail(AZK_ERR_ARG, "bad leaf_dtype");
    if (cfg->leaves_per_step < 0 || cfg->leaves_per_step > 64) return fail(AZK_ERR_ARG, "leaves_per_step must be 0..64");
    if (cfg->cache_entries < 0 || (cfg->cache_entries & (cfg->cache_entries - 1)) != 0) return fail(AZK_ERR_ARG, "cache_entries must be 0 or a power of two");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(AZK_ERR_HIP, "no HIP device visible: libazk needs an MI355X (there is no CPU fallback)");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(AZK_ERR_ARG, "bad device ordinal");
    if (hipSetDevice(cfg->device) != hipSuccess) return fail(AZK_ERR_HIP, "hipSetDevice failed");
    { int khz = 0; if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, cfg->device) == hipSuccess && khz >= 1000) e->ticks_per_us = khz / 1000; }
    const GameDesc &g = d.g;
    const int maxch = g.kind == AZK_CONNECT4 ? g.cols : g.rc;
    long long cap = cfg->arena_nodes > 0 ? cfg->arena_nodes : 1 + (long long)cfg->max_sims * maxch;
    if (cap > 0x7ffffff0LL) return fail(AZK_ERR_ARG, "arena too large");
    d.G = cfg->n_games; d.cap = (int)cap; d.path_cap = g.state_dim + 2; d.rc_pad = up16(g.rc);
    d.leaf_dtype = cfg->leaf_dtype; d.table_size = table_size_for(g);
    { const char *ab = getenv("AZK_TREE_ABLATE"); d.ablate = ab ? atoi(ab) : 0; }
    int off[15];
    d.lds_bytes = lds_layout(g, d.path_cap, d.table_size, off);
    for (int i = 0; i < 12; i++) d.lds_off[i] = off[i];
    const size_t G = d.G, nodes = G * (size_t)d.cap;
    hipError_t s = hipSuccess;
#define DA(ptr, count) if (s == hipSuccess) s = dalloc(e, &ptr, (count))
    DA(d.cells, G * d.rc_pad); DA(d.to_move, G); DA(d.move_count, G); DA(d.done, G); DA(d.winner, G);
    DA(d.H, nodes); DA(d.W, nodes);
    DA(d.rootP, G * g.rc); DA(d.root_f64, G); DA(d.arena_top, G);
    d.K = cfg->leaves_per_step > 1 ? cfg->leaves_per_step : 1;
    const size_t GV = G * (size_t)d.K;                            // pending-leaf slots
    DA(d.leaf_node, GV); DA(d.leaf_depth, GV); DA(d.leaf_nmoves, GV); DA(d.leaf_slot, GV); DA(d.to_move_v, GV);
    DA(d.path, GV * d.path_cap); DA(d.leaf_cells, GV * d.rc_pad); DA(d.leaf_moves, GV * g.rc);
    DA(d.leaf_flag, ((GV + 511) / 512) * 512 + 512);
    DA(d.counters, (size_t)CNT_N * G); DA(d.err, 1); DA(d.dbg, G * 8); DA(d.emit_base, G); DA(d.sims_done, G); DA(d.budget, 4);
    d.cache_entries = cfg->cache_entries;
    d.key_words = 2 * ((g.rc + 64) / 64);                         // one spare bit (63 of the last own-plane word) for the side to move
    d.cache_shared = (d.cache_entries && cfg->cache_shared) ? 1 : 0;
    size_t cache_total = G * (size_t)d.cache_entries;
    if (d.cache_shared) {
        size_t p2 = 1;
        while (p2 * 2 <= cache_total && p2 * 2 <= (size_t)1 << 30) p2 *= 2;       // entry indices travel as int32
        cache_total = p2;
        d.cache_mask = (unsigned long long)cache_total - 1ull;
    }
    if (d.cache_entries) {
        DA(d.cache_key, cache_total * d.key_words); DA(d.cache_logits, cache_total * g.action_dim);
        DA(d.cache_value, cache_total); DA(d.leaf_cache, GV);
        if (d.cache_shared) {
            DA(d.cache_claim, cache_total); DA(d.cache_stamp, 1); DA(d.leaf_key, GV * d.key_words);
            DA(d.hit_logits, GV * g.action_dim); DA(d.hit_value, GV);
        }
    }
    if (g.rows == g.cols && g.action_dim == g.rc) { DA(d.traj_action, G * g.state_dim); DA(d.traj_pi, G * g.state_dim * g.action_dim); } DA(e->counter_sums, CNT_N); DA(e->n_leaf_scratch, 1);
    if (s == hipSuccess) { uint8_t *ls = nullptr; s = dalloc(e, &ls, GV * g.planes * g.rc * 4); e->leaf_scratch = ls; }
#undef DA
    if (s != hipSuccess) return fail(AZK_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(s));
    (void)hipMemset(d.leaf_flag, 0, ((GV + 511) / 512) * 512 + 512);
    (void)hipMemset(d.counters, 0, sizeof(long long) * CNT_N * G);
    (void)hipMemset(d.err, 0, sizeof(int));
    if (d.cache_entries) (void)hipMemset(d.cache_key, 0xff, sizeof(unsigned long long) * cache_total * d.key_words);   // all-ones = no position
    if (d.cache_shared) {
        (void)hipMemset(d.cache_claim, 0, sizeof(unsigned) * cache_total);
        const unsigned one = 1u;
        (void)hipMemcpy(d.cache_stamp, &one, sizeof one, hipMemcpyHostToDevice);
    }
    (void)hipMemset(d.dbg, 0, sizeof(long long) * G * 8);
    (void)hipMemset(d.leaf_node, 0xff, sizeof(int) * GV);
    k_reset_games<<<(unsigned)((G * d.rc_pad + 255) / 256), 256>>>(d, 0, d.G);
    k_begin_search<<<(unsigned)((G + 255) / 256), 256>>>(d);
    s = hipDeviceSynchronize();
    if (s != hipSuccess) return fail(AZK_ERR_HIP, std::string("init kernels: ") + hipGetErrorString(s));
    *out = e;
    return AZK_OK;
}

void azk_destroy(azk_engine *e) {
    if (!e) return;
    for (void *p : e->allocs) (void)hipFree(p);
    delete e;
}

int32_t azk_geometry(const azk_engine *e, int32_t *planes, int32_t *rows, int32_t *cols, int32_t *action_dim, int32_t *state_dim) {
    if (!e) return AZK_ERR_ARG;
    if (planes) *planes = e->d.g.planes;
    if (rows) *rows = e->d.g.rows;
    if (cols) *cols = e->d.g.cols;
    if (action_dim) *action_dim = e->d.g.action_dim;
    if (state_dim) *state_dim = e->d.g.state_dim;
    return AZK_OK;
}

int32_t azk_reset_games(azk_engine *e, int32_t first, int32_t count, void *stream) {
    if (!e || first < 0 || count < 0 || first + count > e->d.G) { if (e) e->err = "azk_reset_games: bad range"; return AZK_ERR_ARG; }
    if (count == 0) return AZK_OK;
    k_reset_games<<<(unsigned)(((size_t)count * e->d.rc_pad + 255) / 256), 256, 0, (hipStream_t)stream>>>(e->d, first, count);
    HIPCHK(e, hipGetLastError());
    return AZK_OK;
}

int32_t azk_set_positions(azk_engine *e, int32_t first, int32_t count, const int8_t *cells_host,
                          const int32_t *to_move_host, const int32_t *move_count_host, void *stream) {
    if (!e || first < 0 || count < 1 || first + count > e->d.G || !cells_host || !to_move_host || !move_count_host) {
        if (e) e->err = "azk_set_positions: bad argument";
        return AZK_ERR_ARG;
    }
    const Dev &d = e->d;
    hipStream_t st = (hipStream_t)stream;
    std::vector<uint8_t> padded((size_t)count * d.rc_pad, 0);
    std::vector<int> zeros(count, 0), win(count, -2);
    for (int i = 0; i < count; i++)
        for (int c = 0; c < d.g.rc; c++) {
            int8_t v = cells_host[(size_t)i * d.g.rc + c];
            if (v < 0 || v > 2) { e->err = "azk_set_positions: cell codes must be 0, 1 or 2"; return AZK_ERR_ARG; }
            padded[(size_t)i * d.rc_pad + c] = (uint8_t)v;
        }
    HIPCHK(e, hipMemcpyAsync(d.cells + (size_t)first * d.rc_pad, padded.data(), padded.size(), hipMemcpyHostToDevice, st));
    HIPCHK(e, hipMemcpyAsync(d.to_move + first, to_move_host, sizeof(int) * count, hipMemcpyHostToDevice, st));
    HIPCHK(e, hipMemcpyAsync(d.move_count + first, move_count_host, sizeof(int) * count, hipMemcpyHostToDevice, st));
    HIPCHK(e, hipMemcpyAsync(d.done + first, zeros.data(), sizeof(int) * count, hipMemcpyHostToDevice, st));
    HIPCHK(e, hipMemcpyAsync(d.winner + first, win.data(), sizeof(int) * count, hipMemcpyHostToDevice, st));
    HIPCHK(e, hipStreamSynchronize(st));   // host staging buffers go out of scope
    return AZK_OK;
}

int32_t azk_begin_search(azk_engine *e, const double *noise_dev, void *stream) {
    if (!e) return AZK_ERR_ARG;
    e->d.noise = noise_dev; e->d.noise_sel = nullptr;
    e->multi = false;
    k_begin_search<<<(unsigned)((e->d.G + 255) / 256), 256, 0, (hipStream_t)stream>>>(e->d);
    HIPCHK(e, hipGetLastError());
    return AZK_OK;
}

int32_t azk_begin_search_budget(azk_engine *e, const double *noise_dev, int32_t n_sims, int32_t max_sims_per_launch, void *stream) {
    if (!e || n_sims < 1 || n_sims > e->cfg.max_sims || max_sims_per_launch < 1) { if (e) e->err = "azk_begin_search_budget: bad argument"; return AZK_ERR_ARG; }
    e->d.noise = noise_dev; e->d.noise_sel = nullptr;
    e->multi = true;
    if (e->budget_host[0] != n_sims || e->budget_host[1] != (e->d.K > 1 ? e->d.K : max_sims_per_launch)) {
        // the budget lives in device memory so that a captured step graph keeps working when it changes
        e->budget_host[0] = n_sims; e->budget_host[1] = max_sims_per_launch; e->budget_host[2] = 0;
        if (e->d.K > 1) e->budget_host[1] = e->d.K;               // virtual-loss mode: one iteration per slot
        HIPCHK(e, hipMemcpyAsync(e->d.budget, e->budget_host, sizeof e->budget_host, hipMemcpyHostToDevice, (hipStream_t)stream));
        HIPCHK(e, hipStreamSynchronize((hipStream_t)stream));
    }
    k_begin_search<<<(unsigned)((e->d.G + 255) / 256), 256, 0, (hipStream_t)stream>>>(e->d);
    HIPCHK(e, hipGetLastError());
    return AZK_OK;
}

// ---- asynchronous self-play ---------------------------------------------------------------------------------------------
int32_t azk_async_begin(azk_engine *e, const azk_async_config *c, void *stream) {
    if (!e || !c || !c->stats_dev || c->n_sims < 1 || c->n_sims > e->cfg.max_sims || c->max_sims_per_launch < 1 || !(c->alpha > 0.0)) {
        if (e) e->err = "azk_async_begin: bad argument";
        return AZK_ERR_ARG;
    }
    if (c->record_capacity < 0 || (c->record_capacity > 0 && (!c->rec_meta_dev || !c->rec_q_dev || !c->rec_pi_dev))) { e->err = "azk_async_begin: record ring pointers missing"; return AZK_ERR_ARG; }
    Dev &d = e->d;
    if (d.K > 1) { e->err = "azk_async_begin: asynchronous moves run the sequential search (leaves_per_step = 1)"; return AZK_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream;
    AsyncDev &a = e->ad;
    if (!a.slot_moves) {
        HIPCHK(e, dalloc(e, &a.slot_moves, (size_t)d.G));
        HIPCHK(e, dalloc(e, &a.noise, (size_t)d.G * 2 * d.g.action_dim));
        HIPCHK(e, dalloc(e, &a.noise_key, (size_t)d.G));
        HIPCHK(e, dalloc(e, &a.todo_list, (size_t)d.G));
        HIPCHK(e, dalloc(e, &a.todo_count, 1));
        HIPCHK(e, dalloc(e, &a.fin_list, (size_t)d.G));
        HIPCHK(e, dalloc(e, &a.fin_count, 1));
    }
    a.n_sims = c->n_sims; a.sample_until = c->sample_until_move; a.dirichlet = c->dirichlet ? 1 : 0;
    a.seed = c->seed; a.first_game = c->first_global_game; a.alpha = c->alpha;
    a.stats = (long long *)c->stats_dev; a.rec_cap = c->record_capacity; a.rec_meta = c->rec_meta_dev; a.rec_q = c->rec_q_dev; a.rec_pi = c->rec_pi_dev;
    e->async_recycle = c->recycle ? 1 : 0;
    HIPCHK(e, hipMemsetAsync(a.slot_moves, 0, sizeof(long long) * (size_t)d.G, st));
    HIPCHK(e, hipMemsetAsync(a.stats, 0, sizeof(long long) * 16, st));
    // the simulation budget lives in device memory (a captured step graph keeps working when it changes)
    e->budget_host[0] = c->n_sims; e->budget_host[1] = c->max_sims_per_launch;
    e->budget_host[2] = c->young_launch_us > 0 ? c->young_launch_us * e->ticks_per_us : 0;
    HIPCHK(e, hipMemcpyAsync(d.budget, e->budget_host, sizeof e->budget_host, hipMemcpyHostToDevice, st));
    HIPCHK(e, hipStreamSynchronize(st));
    d.noise = a.dirichlet ? a.noise : nullptr;
    d.noise_sel = a.dirichlet ? a.slot_moves : nullptr;
    e->multi = true;
    e->async_on = true;
    // first search of every game: fresh roots + the Dirichlet rows of move keys 0 (this search) and 1 (the next one)
    k_begin_search<<<(unsigned)((d.G + 255) / 256), 256, 0, st>>>(d);
    HIPCHK(e, hipMemsetAsync(a.todo_count, 0, sizeof(int), st));
    if (a.dirichlet) {
        const int A = d.g.action_dim;
        k_gen_noise<<<d.G, AZK_WAVE, 0, st>>>(A, a.seed, a.first_game, 0, a.alpha, a.noise, nullptr, 2LL * A);
        k_gen_noise<<<d.G, AZK_WAVE, 0, st>>>(A, a.seed, a.first_game, 1, a.alpha, a.noise + A, nullptr, 2LL * A);
        k_fill_i32<<<(unsigned)((d.G + 255) / 256), 256, 0, st>>>(a.noise_key, d.G, 1);
    }
    HIPCHK(e, hipGetLastError());
    return AZK_OK;
}

int32_t azk_async_step(azk_engine *e, const float *logits_dev, const float *values_dev, int32_t phases, void *stream) {
    if (!e || !e->async_on) { if (e) e->err = "azk_async_step: call azk_async_begin first"; return AZK_ERR_STATE; }
    const Dev &d = e->d;
    hipStream_t st = (hipStream_t)stream;
    if (phases & 1) AZK_LAUNCH_TREE(true, true, false, true, (d, logits_dev, values_dev));
    if (phases & 2) k_move_async<<<d.G, AZK_WAVE, d.lds_bytes, st>>>(d, e->ad);
    HIPCHK(e, hipGetLastError());
    return AZK_OK;
}

int32_t azk_async_set_budget(azk_engine *e, int32_t n_sims, int32_t max_sims_per_launch, void *stream) {
    if (!e || !e->async_on || n_sims < 1 || n_sims > e->cfg.max_sims || max_sims_per_launch < 1) { if (e) e->err = "azk_async_set_budget: bad argument"; return AZK_ERR_ARG; }
    e->budget_host[0] = n_sims; e->budget_host[1] = max_sims_per_launch;
    HIPCHK(e, hipMemcpyAsync(e->d.budget, e->budget_host, sizeof e->budget_host, hipMemcpyHostToDevice, (hipStream_t)stream));
    HIPCHK(e, hipStreamSynchronize((hipStream_t)stream));
    return AZK_OK;
}

int32_t azk_async_drain(azk_engine *e, float *states_dev, double *pis_dev, float *zs_dev, int64_t capacity, int64_t *cursor_dev, void *stream) {
    if (!e || !e->async_on) { if (e) e->err = "azk_async_drain: call azk_async_begin first"; return AZK_ERR_STATE; }
    const Dev &d = e->d;
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(e, hipMemsetAsync(e->ad.fin_count, 0, sizeof(int), st));
    k_async_list<<<(unsigned)((d.G + 255) / 256), 256, 0, st>>>(d, e->ad);
    if (states_dev) {
        if (!pis_dev || !zs_dev || !cursor_dev || capacity < 1 || !d.traj_pi) { e->err = "azk_async_drain: bad replay arguments"; return AZK_ERR_ARG; }
        k_emit_alloc<<<(unsigned)((d.G + 255) / 256), 256, 0, st>>>(d, (unsigned long long *)cursor_dev, nullptr);
        const int blocks = d.G * d.g.state_dim < 16384 ? d.G * d.g.state_dim : 16384;
        k_emit_tuples<true><<<(unsigned)blocks, AZK_WAVE, up16(d.g.rc), st>>>(d, states_dev, pis_dev, zs_dev, (long long)capacity,
                                                                            (const unsigned long long *)cursor_dev, e->ad.fin_list, e->ad.fin_count);
        k_emit_mark<<<(unsigned)((d.G + 255) / 256), 256, 0, st>>>(d);
    }
    k_async_restart<<<(unsigned)(d.G < 256 ? d.G : 256), AZK_WAVE, d.lds_bytes, st>>>(d, e->ad, e->async_recycle);
    if (e->ad.dirichlet) {                                        // the rows of the searches AFTER the ones begun since the last drain
        k_noise_ahead<<<(unsigned)(d.G < 512 ? d.G : 512), AZK_WAVE, 0, st>>>(d, e->ad);
        HIPCHK(e, hipMemsetAsync(e->ad.todo_count, 0, sizeof(int), st));
    }
    HIPCHK(e, hipGetLastError());
    return AZK_OK;
}

namespace {
__global__ void k_unfinished(Dev d, int *out) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    bool open = g < d.G && d.done[g] == 0 && d.sims_done[g] < d.budget[0];
    if (g < d.G && d.done[g] == 0)
        for (int k = 0; k < d.K; k++)        // a leaf awaiting the evaluator; with K > 1 also one served by the cache (only a tree launch expands it)
            open = open || (d.leaf_node[g * d.K + k] >= 0 && (d.leaf_flag[g * d.K + k] || d.K > 1));
    const unsigned long long m = __ballot(open);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(out, __popcll(m));
}
}  // namespace

int32_t azk_search_unfinished(azk_engine *e, int32_t *count_host, void *stream) {
    if (!e || !count_host) return AZK_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(e, hipMemsetAsync(e->n_leaf_scratch, 0, sizeof(int), st));
    k_unfinished<<<(unsigned)((e->d.G + 255) / 256), 256, 0, st>>>(e->d, e->n_leaf_scratch);
    HIPCHK(e, hipMemcpyAsync(count_host, e->n_leaf_scratch, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(e, hipStreamSynchronize(st));
    return AZK_OK;
}

static int32_t launch_tree(azk_engine *e, bool expand, bool select, const float *logits, const float *values,
                           void *leaf_boards, int32_t *n_leaf, hipStream_t st) {
    const Dev &d = e->d;
    if (expand && (!logits || !values)) { e->err = "expand needs logits_dev and values_dev"; return AZK_ERR_ARG; }
    if (select && (!leaf_boards || !n_leaf)) { e->err = "select needs leaf_boards_dev and n_leaf_dev"; return AZK_ERR_ARG; }
    if (d.ablate) {
        if (expand && select) AZK_LAUNCH_TREE(true, true, true, false, (d, logits, values));
        else if (expand) AZK_LAUNCH_TREE(true, false, true, false, (d, logits, values));
        else AZK_LAUNCH_TREE(false, true, true, false, (d, logits, values));
    } else if (e->multi && select) {
        // budget stepping always carries the expansion code: a leaf served by the cache is expanded inside the launch (logits may
        // be null when no game has a pending evaluation, e.g. in a search's first launch)
        AZK_LAUNCH_TREE(true, true, false, true, (d, logits, values));
    } else if (expand && select) AZK_LAUNCH_TREE(true, true, false, false, (d, logits, values));
    else if (expand) AZK_LAUNCH_TREE(true, false, false, false, (d, logits, values));
    else AZK_LAUNCH_TREE(false, true, false, false, (d, logits, values));
    HIPCHK(e, hipGetLastError());
    if (select) {
        k_gather<<<d.G * d.K, AZK_WAVE, 0, st>>>(d, leaf_boards, n_leaf);
        HIPCHK(e, hipGetLastError());
    }
    return AZK_OK;
}

int32_t azk_step_select(azk_engine *e, void *leaf_boards_dev, int32_t *n_leaf_dev, void *stream) {
    if (!e) return AZK_ERR_ARG;
    return launch_tree(e, false, true, nullptr, nullptr, leaf_boards_dev, n_leaf_dev, (hipStream_t)stream);
}

int32_t azk_step_expand_backup(azk_engine *e, const float *logits_dev, const float *values_dev, void *stream) {
    if (!e) return AZK_ERR_ARG;
    return launch_tree(e, true, false, logits_dev, values_dev, nullptr, nullptr, (hipStream_t)stream);
}

int32_t azk_step(azk_engine *e, const float *logits_dev, const float *values_dev, void *leaf_boards_dev,
                 int32_t *n_leaf_dev, void *stream) {
    if (!e) return AZK_ERR_ARG;
    return launch_tree(e, logits_dev != nullptr, true, logits_dev, values_dev, leaf_boards_dev, n_leaf_dev, (hipStream_t)stream);
}

int32_t azk_step_tree(azk_engine *e, const float *logits_dev, const float *values_dev, void *stream) {
    if (!e) return AZK_ERR_ARG;
    const Dev &d = e->d;
    hipStream_t st = (hipStream_t)stream;
    if (logits_dev && !values_dev) { e->err = "values_dev missing"; return AZK_ERR_ARG; }
    if (d.ablate) {
        if (logits_dev) AZK_LAUNCH_TREE(true, true, true, false, (d, logits_dev, values_dev));
        else AZK_LAUNCH_TREE(false, true, true, false, (d, nullptr, nullptr));
    } else if (e->multi) {
        AZK_LAUNCH_TREE(true, true, false, true, (d, logits_dev, values_dev));
    } else if (logits_dev) AZK_LAUNCH_TREE(true, true, false, false, (d, logits_dev, values_dev));
    else AZK_LAUNCH_TREE(false, true, false, false, (d, nullptr, nullptr));
    HIPCHK(e, hipGetLastError());
    return AZK_OK;
}

int32_t azk_step_gather(azk_engine *e, void *leaf_boards_dev, int32_t *n_leaf_dev, void *stream) {
    if (!e || !leaf_boards_dev || !n_leaf_dev) return AZK_ERR_ARG;
    k_gather<<<e->d.G, AZK_WAVE, 0, (hipStream_t)stream>>>(e->d, leaf_boards_dev, n_leaf_dev);
    HIPCHK(e, hipGetLastError());
    return AZK_OK;
}


static int32_t vanilla_prepare(azk_engine *e) {
    if (e->vanilla_rng) return AZK_OK;
    const size_t G = e->d.G;
    HIPCHK(e, dalloc(e, &e->vanilla_rng, G * 625));
    e->lntab_n = e->cfg.max_sims + 2;
    HIPCHK(e, dalloc(e, &e->lntab, (size_t)e->lntab_n));
    std::vector<double> t(e->lntab_n, 0.0);
    for (int i = 1; i < e->lntab_n; i++) t[i] = log((double)i);
    HIPCHK(e, hipMemcpy(e->lntab, t.data(), sizeof(double) * t.size(), hipMemcpyHostToDevice));
    std::vector<uint32_t> st(G * 625);
    for (size_t g = 0; g < G; g++) {                                  // default streams: init_genrand(5489 + game)
        uint32_t *m = st.data() + g * 625;
        m[0] = 5489u + (uint32_t)g;
        for (int i = 1; i < 624; i++) m[i] = 1812433253u * (m[i - 1] ^ (m[i - 1] >> 30)) + (uint32_t)i;
        m[624] = 624;
    }
    HIPCHK(e, hipMemcpy(e->vanilla_rng, st.data(), sizeof(uint32_t) * st.size(), hipMemcpyHostToDevice));
    return AZK_OK;
}

int32_t azk_vanilla_set_rng(azk_engine *e, int32_t first, int32_t count, const uint32_t *mt_states_host, void *stream) {
    if (!e || !mt_states_host || first < 0 || count < 1 || first + count > e->d.G) { if (e) e->err = "azk_vanilla_set_rng: bad argument"; return AZK_ERR_ARG; }
    for (int i = 0; i < count; i++)
        if (mt_states_host[(size_t)i * 625 + 624] > 624u) { e->err = "azk_vanilla_set_rng: position must be in [0, 624]"; return AZK_ERR_ARG; }
    int32_t rc = vanilla_prepare(e);
    if (rc != AZK_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(e, hipMemcpyAsync(e->vanilla_rng + (size_t)first * 625, mt_states_host, sizeof(uint32_t) * 625 * (size_t)count, hipMemcpyHostToDevice, st));
    HIPCHK(e, hipStreamSynchronize(st));
    return AZK_OK;
}

int32_t azk_vanilla_get_rng(azk_engine *e, int32_t first, int32_t count, uint32_t *mt_states_host, void *stream) {
    if (!e || !mt_states_host || first < 0 || count < 1 || first + count > e->d.G) { if (e) e->err = "azk_vanilla_get_rng: bad argument"; return AZK_ERR_ARG; }
    int32_t rc = vanilla_prepare(e);
    if (rc != AZK_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(e, hipMemcpyAsync(mt_states_host, e->vanilla_rng + (size_t)first * 625, sizeof(uint32_t) * 625 * (size_t)count, hipMemcpyDeviceToHost, st));
    HIPCHK(e, hipStreamSynchronize(st));
    return AZK_OK;
}

int32_t azk_vanilla_search(azk_engine *e, int32_t n_sims, void *stream) {
    if (!e || n_sims < 0) { if (e) e->err = "azk_vanilla_search: bad argument"; return AZK_ERR_ARG; }
    int32_t rc = vanilla_prepare(e);
    if (rc != AZK_OK) return rc;
    if (n_sims == 0) return AZK_OK;
    const Dev &d = e->d;
    k_vanilla<<<d.G, AZK_WAVE, d.lds_bytes + 625 * 4, (hipStream_t)stream>>>(d, n_sims, e->vanilla_rng, e->lntab, e->lntab_n);
    HIPCHK(e, hipGetLastError());
    return AZK_OK;
}

int32_t azk_leaf_source_of(azk_engine *e, int32_t *n_leaf_dev, azk_leaf_source *out) {
    if (!e || !n_leaf_dev || !out) return AZK_ERR_ARG;
    const Dev &d = e->d;
    out->leaf_flag = d.leaf_flag; out->leaf_cells = d.leaf_cells; out->to_move = d.K > 1 ? d.to_move_v : d.to_move; out->leaf_depth = d.leaf_depth;
    out->leaf_slot = d.leaf_slot; out->n_leaf = n_leaf_dev;
    out->n_games = d.G * d.K; out->rows = d.g.rows; out->cols = d.g.cols; out->rc = d.g.rc; out->rc_pad = d.rc_pad; out->planes = d.g.planes;
    out->flag_bytes = ((d.G * d.K + 511) / 512) * 512 + 512;
    out->cache_stamp = (d.cache_entries && d.cache_shared) ? d.cache_stamp : nullptr;
    return AZK_OK;
}

int32_t azk_emit_finished(azk_engine *e, float *states_dev, double *pis_dev, float *zs_dev, int64_t capacity,
                          int64_t *cursor_dev, int64_t *game_base_dev, void *stream) {
    if (!e || !states_dev || !pis_dev || !zs_dev || !cursor_dev || capacity < 1) return AZK_ERR_ARG;
    const Dev &d = e->d;
    if (!d.traj_pi) { e->err = "azk_emit_finished: (state, pi, z) emission needs a square board with one action per cell"; return AZK_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream;
    k_emit_alloc<<<(unsigned)((d.G + 255) / 256), 256, 0, st>>>(d, (unsigned long long *)cursor_dev, (long long *)game_base_dev);
    k_emit_tuples<false><<<(unsigned)(d.G * d.g.state_dim), AZK_WAVE, up16(d.g.rc), st>>>(d, states_dev, pis_dev, zs_dev, (long long)capacity,
                                                                                         (const unsigned long long *)cursor_dev, nullptr, nullptr);
    k_emit_mark<<<(unsigned)((d.G + 255) / 256), 256, 0, st>>>(d);
    HIPCHK(e, hipGetLastError());
    return AZK_OK;
}

int32_t azk_clear_cache(azk_engine *e, void *stream) {
    if (!e) return AZK_ERR_ARG;
    const Dev &d = e->d;
    if (!d.cache_entries) return AZK_OK;
    if (d.cache_shared) {       // an entry whose claim word is 0 was never written: clearing the claims empties the table
        HIPCHK(e, hipMemsetAsync(d.cache_claim, 0, sizeof(unsigned) * (size_t)(d.cache_mask + 1ull), (hipStream_t)stream));
        return AZK_OK;
    }
    HIPCHK(e, hipMemsetAsync(d.cache_key, 0xff, sizeof(unsigned long long) * (size_t)d.G * d.cache_entries * d.key_words, (hipStream_t)stream));
    return AZK_OK;
}

int32_t azk_recycle_finished(azk_engine *e, int64_t *stats_dev, void *stream) {
    if (!e || !stats_dev) return AZK_ERR_ARG;
    k_recycle<<<(unsigned)((e->d.G + 255) / 256), 256, 0, (hipStream_t)stream>>>(e->d, (long long *)stats_dev);
    HIPCHK(e, hipGetLastError());
    return AZK_OK;
}

int32_t azk_root_stats(azk_engine *e, double *pi_dev, double *q_dev, int32_t *root_visit_dev, void *stream) {
    if (!e) return AZK_ERR_ARG;
    k_root_stats<<<e->d.G, AZK_WAVE, e->d.lds_bytes, (hipStream_t)stream>>>(e->d, pi_dev, q_dev, root_visit_dev);
    HIPCHK(e, hipGetLastError());
    return AZK_OK;
}

int32_t azk_advance(azk_engine *e, const double *uniforms_dev, int32_t sample_until_move, int32_t *chosen_cell_dev,
                    int32_t *winner_dev, int32_t *done_dev, void *stream) {
    if (!e) return AZK_ERR_ARG;
    k_advance<<<e->d.G, AZK_WAVE, e->d.lds_bytes, (hipStream_t)stream>>>(e->d, uniforms_dev, sample_until_move,
                                                                         chosen_cell_dev, winner_dev, done_dev);
    HIPCHK(e, hipGetLastError());
    return AZK_OK;
}

// copy one game's used arena to the host
struct HostTree {
    std::vector<int> N, first_child;
    std::vector<double> W, rootP;
    std::vector<float> P;
    std::vector<uint32_t> meta;
    std::vector<int> cell, nch;
    int top = 0, root_f64 = 0;
};

static int32_t fetch_tree(azk_engine *e, int game, HostTree *t, hipStream_t st) {
    const Dev &d = e->d;
    if (game < 0 || game >= d.G) { e->err = "bad game index"; return AZK_ERR_ARG; }
    HIPCHK(e, hipMemcpyAsync(&t->top, d.arena_top + game, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(e, hipMemcpyAsync(&t->root_f64, d.root_f64 + game, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(e, hipStreamSynchronize(st));
    const size_t n = (size_t)t->top, base = (size_t)game * d.cap;
    t->N.resize(n); t->first_child.resize(n); t->W.resize(n); t->P.resize(n); t->cell.resize(n); t->nch.resize(n); t->meta.resize(n);
    t->rootP.resize(d.g.rc);
    std::vector<NodeH> recs(n);
    HIPCHK(e, hipMemcpyAsync(recs.data(), d.H + base, n * sizeof(NodeH), hipMemcpyDeviceToHost, st));
    HIPCHK(e, hipMemcpyAsync(t->W.data(), d.W + base, n * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(e, hipMemcpyAsync(t->rootP.data(), d.rootP + (size_t)game * d.g.rc, d.g.rc * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(e, hipStreamSynchronize(st));
    for (size_t i = 0; i < n; i++) { t->N[i] = recs[i].N; t->first_child[i] = recs[i].fc; t->P[i] = recs[i].P; t->meta[i] = recs[i].meta; }
    for (size_t i = 0; i < n; i++) {
        const int c = (int)(t->meta[i] >> 16);
        t->cell[i] = c == 0xffff ? -1 : c;
        t->nch[i] = (int)(t->meta[i] & 0xffffu);
    }
    return AZK_OK;
}

int32_t azk_root_children(azk_engine *e, int32_t game, int32_t cap, int32_t *cells_host, int32_t *visits_host,
                          double *values_host, double *priors_host, void *stream) {
    if (!e) return AZK_ERR_ARG;
    HostTree t;
    int32_t rc = fetch_tree(e, game, &t, (hipStream_t)stream);
    if (rc != AZK_OK) return rc;
    const int fc = t.first_child[0], n = t.nch[0];
    for (int i = 0; i < n && i < cap; i++) {
        if (cells_host) cells_host[i] = t.cell[fc + i];
        if (visits_host) visits_host[i] = t.N[fc + i];
        if (values_host) values_host[i] = t.W[fc + i];
        if (priors_host) priors_host[i] = t.root_f64 ? t.rootP[i] : (double)t.P[fc + i];
    }
    return n;
}

int32_t azk_export_tree(azk_engine *e, int32_t game, int32_t cap, int32_t *depth_host, int32_t *cell_host,
                        int32_t *visit_host, double *value_host, double *prior_host, void *stream) {
    if (!e) return AZK_ERR_ARG;
    HostTree t;
    int32_t rc = fetch_tree(e, game, &t, (hipStream_t)stream);
    if (rc != AZK_OK) return rc;
    std::vector<std::pair<int, int>> stack;
    stack.push_back({0, 0});
    int m = 0;
    const int rfc = t.first_child[0];
    while (!stack.empty()) {
        auto [node, dep] = stack.back();
        stack.pop_back();
        if (m < cap) {
            if (depth_host) depth_host[m] = dep;
            if (cell_host) cell_host[m] = t.cell[node];
            if (visit_host) visit_host[m] = t.N[node];
            if (value_host) value_host[m] = t.W[node];
            if (prior_host) {
                const bool root_child = dep == 1 && t.root_f64;
                prior_host[m] = root_child ? t.rootP[node - rfc] : (double)t.P[node];
            }
        }
        m++;
        for (int i = t.nch[node] - 1; i >= 0; i--) stack.push_back({t.first_child[node] + i, dep + 1});
    }
    return m;
}

int32_t azk_get_positions(azk_engine *e, int8_t *cells_host, int32_t *to_move_host, int32_t *move_count_host, void *stream) {
    if (!e) return AZK_ERR_ARG;
    const Dev &d = e->d;
    hipStream_t st = (hipStream_t)stream;
    std::vector<uint8_t> padded((size_t)d.G * d.rc_pad);
    HIPCHK(e, hipMemcpyAsync(padded.data(), d.cells, padded.size(), hipMemcpyDeviceToHost, st));
    if (to_move_host) HIPCHK(e, hipMemcpyAsync(to_move_host, d.to_move, sizeof(int) * d.G, hipMemcpyDeviceToHost, st));
    if (move_count_host) HIPCHK(e, hipMemcpyAsync(move_count_host, d.move_count, sizeof(int) * d.G, hipMemcpyDeviceToHost, st));
    HIPCHK(e, hipStreamSynchronize(st));
    if (cells_host)
        for (int g = 0; g < d.G; g++) memcpy(cells_host + (size_t)g * d.g.rc, padded.data() + (size_t)g * d.rc_pad, d.g.rc);
    return AZK_OK;
}

int32_t azk_get_counters(azk_engine *e, azk_counters *out, void *stream) {
    if (!e || !out) return AZK_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    k_sum_counters<<<CNT_N, 256, 0, st>>>(e->d.counters, e->d.G, e->counter_sums);
    HIPCHK(e, hipGetLastError());
    long long h[CNT_N];
    HIPCHK(e, hipMemcpyAsync(h, e->counter_sums, sizeof h, hipMemcpyDeviceToHost, st));
    HIPCHK(e, hipStreamSynchronize(st));
    memset(out, 0, sizeof *out);
    out->sims = h[CNT_SIMS]; out->edges_scanned = h[CNT_SCANNED]; out->trace_nodes = h[CNT_TRACE];
    out->edges_created = h[CNT_CREATED]; out->leaves_evaluated = h[CNT_LEAVES]; out->terminal_sims = h[CNT_TERMINAL];
    out->moves_played = h[CNT_MOVES]; out->cache_hits = h[CNT_CACHE_HITS];
    return AZK_OK;
}

int32_t azk_debug_stamps(azk_engine *e, int64_t *out8_host) {
    if (!e || !out8_host) return AZK_ERR_ARG;
    std::vector<long long> h((size_t)e->d.G * 8);
    if (hipMemcpy(h.data(), e->d.dbg, h.size() * sizeof(long long), hipMemcpyDeviceToHost) != hipSuccess) return AZK_ERR_HIP;
    for (int k = 0; k < 8; k++) { long long s = 0; for (int g = 0; g < e->d.G; g++) s += h[(size_t)g * 8 + k]; out8_host[k] = s; }
    { long long mx = 0; for (int g = 0; g < e->d.G; g++) if (h[(size_t)g * 8 + 7] > mx) mx = h[(size_t)g * 8 + 7]; if (getenv("AZK_STAMP_MAX")) out8_host[7] = mx; }
    return AZK_OK;
}

int32_t azk_debug_stamps_raw(azk_engine *e, int64_t *out_host, int32_t n_games) {
    if (!e || !out_host || n_games != e->d.G) return AZK_ERR_ARG;
    return hipMemcpy(out_host, e->d.dbg, (size_t)e->d.G * 8 * sizeof(long long), hipMemcpyDeviceToHost) == hipSuccess ? AZK_OK : AZK_ERR_HIP;
}

int32_t azk_reset_counters(azk_engine *e, void *stream) {
    if (!e) return AZK_ERR_ARG;
    HIPCHK(e, hipMemsetAsync(e->d.counters, 0, sizeof(long long) * CNT_N * e->d.G, (hipStream_t)stream));
    return AZK_OK;
}

int32_t azk_check_device_error(azk_engine *e, void *stream) {
    if (!e) return AZK_ERR_ARG;
    int h = 0;
    HIPCHK(e, hipMemcpyAsync(&h, e->d.err, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPCHK(e, hipStreamSynchronize((hipStream_t)stream));
    if (h == AZK_ERR_ARENA_FULL) e->err = "tree arena full: raise azk_config.max_sims / arena_nodes";
    else if (h != 0) e->err = "device-side state error (search advanced without visits?)";
    return h;
}

int32_t azk_gen_noise(azk_engine *e, uint64_t seed, int64_t first_global_game, int32_t move_index, double alpha,
                      double *noise_dev, double *uniforms_dev, void *stream) {
    if (!e || alpha <= 0.0) return AZK_ERR_ARG;
    k_gen_noise<<<e->d.G, AZK_WAVE, 0, (hipStream_t)stream>>>(e->d.g.action_dim, seed, first_global_game, move_index, alpha,
                                                            noise_dev, uniforms_dev, (long long)e->d.g.action_dim);
    HIPCHK(e, hipGetLastError());
    return AZK_OK;
}

// ---- stateless rule kernels ---------------------------------------------------------------------
static int32_t run_rules(int mode, int32_t game, int32_t rows, int32_t cols, const float *in, float *inout, int32_t n,
                         const int32_t *players, const int32_t *cells, int16_t *moves, int32_t *counts, uint8_t *mask,
                         int32_t *out_i, float *out_f, void *stream) {
    RuleArgs a;
    memset(&a, 0, sizeof a);
    std::string err;
    if (!make_game(game, rows, cols, &a.g, &err)) { g_create_error = err; return AZK_ERR_ARG; }
    if (n < 0) { g_create_error = "negative batch"; return AZK_ERR_ARG; }
    if (n == 0) return AZK_OK;
    a.mode = mode; a.n = n; a.table_size = table_size_for(a.g);
    a.boards_in = in; a.boards = inout; a.players = players; a.cells = cells;
    a.moves = moves; a.counts = counts; a.mask = mask; a.out_i = out_i; a.out_f = out_f;
    int off[15];
    const int lds = lds_layout(a.g, 4, a.table_size, off);
    k_rules<<<n, AZK_WAVE, lds, (hipStream_t)stream>>>(a);
    hipError_t s = hipGetLastError();
    if (s != hipSuccess) { g_create_error = std::string("k_rules: ") + hipGetErrorString(s); return AZK_ERR_HIP; }
    return AZK_OK;
}

int32_t azk_rules_legal_moves(int32_t game, int32_t rows, int32_t cols, const float *boards_dev, int32_t n,
                              int16_t *moves_dev, int32_t *counts_dev, void *stream) {
    if (!boards_dev || !moves_dev || !counts_dev) return AZK_ERR_ARG;
    return run_rules(RULE_MOVES, game, rows, cols, boards_dev, nullptr, n, nullptr, nullptr, moves_dev, counts_dev, nullptr, nullptr, nullptr, stream);
}
int32_t azk_rules_legal_mask(int32_t game, int32_t rows, int32_t cols, const float *boards_dev, int32_t n,
                             uint8_t *mask_dev, void *stream) {
    if (!boards_dev || !mask_dev) return AZK_ERR_ARG;
    return run_rules(RULE_MASK, game, rows, cols, boards_dev, nullptr, n, nullptr, nullptr, nullptr, nullptr, mask_dev, nullptr, nullptr, stream);
}
int32_t azk_rules_apply_move(int32_t game, int32_t rows, int32_t cols, float *boards_dev, int32_t n,
                             const int32_t *players_dev, const int32_t *cells_dev, int32_t *next_player_dev, void *stream) {
    if (!boards_dev || !players_dev || !cells_dev || !next_player_dev) return AZK_ERR_ARG;
    return run_rules(RULE_APPLY, game, rows, cols, nullptr, boards_dev, n, players_dev, cells_dev, nullptr, nullptr, nullptr, next_player_dev, nullptr, stream);
}
int32_t azk_rules_undo_move(int32_t game, int32_t rows, int32_t cols, float *boards_dev, int32_t n,
                            const int32_t *current_players_dev, const int32_t *cells_dev, void *stream) {
    if (!boards_dev || !current_players_dev || !cells_dev) return AZK_ERR_ARG;
    return run_rules(RULE_UNDO, game, rows, cols, nullptr, boards_dev, n, current_players_dev, cells_dev, nullptr, nullptr, nullptr, nullptr, nullptr, stream);
}
int32_t azk_rules_check_winner(int32_t game, int32_t rows, int32_t cols, const float *boards_dev, int32_t n,
                               const int32_t *players_dev, const int32_t *cells_dev, int32_t *winners_dev, void *stream) {
    if (!boards_dev || !players_dev || !cells_dev || !winners_dev) return AZK_ERR_ARG;
    return run_rules(RULE_WINNER, game, rows, cols, boards_dev, nullptr, n, players_dev, cells_dev, nullptr, nullptr, nullptr, winners_dev, nullptr, stream);
}
int32_t azk_rules_canonical(int32_t game, int32_t rows, int32_t cols, const float *boards_dev, int32_t n,
                            const int32_t *players_dev, float *out_dev, void *stream) {
    if (!boards_dev || !players_dev || !out_dev) return AZK_ERR_ARG;
    return run_rules(RULE_CANON, game, rows, cols, boards_dev, nullptr, n, players_dev, nullptr, nullptr, nullptr, nullptr, nullptr, out_dev, stream);
}

int32_t azk_softmax_rows(const float *logits_dev, int32_t n, int32_t action_dim, float *out_dev, void *stream) {
    if (!logits_dev || !out_dev || n < 0 || action_dim < 1 || action_dim > 512) return AZK_ERR_ARG;
    if (n == 0) return AZK_OK;
    const int lds = (((action_dim + 31) & ~31) + 32) * 4;
    k_softmax_rows<<<n, AZK_WAVE, lds, (hipStream_t)stream>>>(logits_dev, action_dim, out_dev);
    return hipGetLastError() == hipSuccess ? AZK_OK : AZK_ERR_HIP;
}

}  // extern "C"
