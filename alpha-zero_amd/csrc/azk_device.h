// azk_device.h - device-side building blocks shared by the engine and the stateless rule kernels.
// gfx950 only: 64-lane wavefronts; one wavefront (= one 64-thread workgroup) owns one game/board.
// Reference lines cited as file:line relative to the reference root.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define AZK_WAVE 64
#define AZK_KIND_TTT 0
#define AZK_KIND_C4 1
#define AZK_KIND_GOMOKU 2

// Cell codes held in LDS / HBM boards: bit0 = player-0 stone (plane0 == 1), bit1 = player-1 stone
// (plane1 == 1), bit2 = "occupied by something that is not exactly 1.0" (only reachable through the
// stateless float32 API).  Empty <=> code == 0 (the reference tests plane0 == 0 and plane1 == 0).
struct GameDesc {
    int kind, rows, cols, rc, planes, win_len, action_dim, state_dim;
    unsigned inv_cols;      // ceil(65536 / cols): cell / cols == (cell * inv_cols) >> 16, exact for cell < 65536 / cols (make_game)
};

__device__ __forceinline__ int azk_action_idx(const GameDesc &g, int cell) {
    return g.kind == AZK_KIND_C4 ? cell % g.cols : cell;   // connect4.py:29 / gomoku.py:48 / tictactoe.py:34
}

__device__ __forceinline__ int azk_lane() { return threadIdx.x & 63; }

__device__ __forceinline__ int uniform_i32(int v) { return __builtin_amdgcn_readfirstlane(v); }

// ---- single-wave workgroups: ordering without waiting -------------------------------------------
// Every kernel built on this header runs one 64-lane wave per workgroup.  The LDS executes a wave's instructions in issue
// order, so "lane A writes, lane B reads" needs no s_waitcnt between the two - only that the compiler keeps their order.
// (__syncthreads() in a single-wave workgroup is already barrier-free, but it still drains the LDS queue: one full round trip.)
__device__ __forceinline__ void azk_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// bit `lane` of a wave-uniform 64-bit mask selects between two per-lane values: the mask goes to v_cndmask as its scalar
// select operand (no per-lane shift / and / compare)
__device__ __forceinline__ unsigned azk_sel_mask(unsigned long long mask, unsigned if_set, unsigned if_clear) {
    unsigned r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(mask));
    return r;
}

__device__ __forceinline__ unsigned long long azk_readlane_u64(unsigned long long v, int lane) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, lane), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), lane);
    return ((unsigned long long)hi << 32) | lo;
}

// ---- wave reductions -------------------------------------------------------------------------
// argmax with "first maximum wins" (Python max(), node.py:47,81): larger value, then lower index.
// Steps inside a 16-lane row use DPP lane permutes (quad_perm, row_half_mirror, row_mirror: any pairing of disjoint
// halves is a valid reduction step); the two cross-row steps use ds_bpermute.  Every lane ends with the winner.
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false); }
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) { return __builtin_bit_cast(float, dpp_i32<CTRL>(__builtin_bit_cast(int, v))); }
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    long long b = __builtin_bit_cast(long long, v);
    int lo = dpp_i32<CTRL>((int)b), hi = dpp_i32<CTRL>((int)(b >> 32));
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}
template <int CTRL> __device__ __forceinline__ float dpp_val(float v) { return dpp_f32<CTRL>(v); }
template <int CTRL> __device__ __forceinline__ double dpp_val(double v) { return dpp_f64<CTRL>(v); }
template <int CTRL> __device__ __forceinline__ int dpp_val(int v) { return dpp_i32<CTRL>(v); }

template <typename T>
__device__ __forceinline__ void argmax_combine(T &v, int &idx, T ov, int oi) {
    // bitwise on purpose (no short-circuit): one compare chain and two selects instead of nested exec-mask branches
    const bool take = (oi < 0x7fffffff) & ((idx == 0x7fffffff) | (ov > v) | ((ov == v) & (oi < idx)));
    v = take ? ov : v;
    idx = take ? oi : idx;
}

template <typename T>
__device__ __forceinline__ void wave_argmax_first(T &v, int &idx) {
    argmax_combine(v, idx, dpp_val<0xB1>(v), dpp_i32<0xB1>(idx));     // quad_perm [1,0,3,2]
    argmax_combine(v, idx, dpp_val<0x4E>(v), dpp_i32<0x4E>(idx));     // quad_perm [2,3,0,1]
    argmax_combine(v, idx, dpp_val<0x141>(v), dpp_i32<0x141>(idx));   // row_half_mirror
    argmax_combine(v, idx, dpp_val<0x140>(v), dpp_i32<0x140>(idx));   // row_mirror
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {
        T ov = __shfl_xor(v, off);
        int oi = __shfl_xor(idx, off);
        argmax_combine(v, idx, ov, oi);
    }
}

// Same reduction without LDS-crossbar shuffles: the four in-row steps, then row_bcast:15 into rows 1,3 and row_bcast:31 into
// rows 2,3 (GFX9 DPP); the wave's result is valid in LANE 63 ONLY - read it with __builtin_amdgcn_readlane(x, 63).
template <int CTRL, int ROWMASK>
__device__ __forceinline__ int dpp_bcast_i32(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, ROWMASK, 0xF, false); }

template <typename T>
__device__ __forceinline__ void wave_argmax_first_lane63(T &v, int &idx) {
    argmax_combine(v, idx, dpp_val<0xB1>(v), dpp_i32<0xB1>(idx));     // quad_perm [1,0,3,2]
    argmax_combine(v, idx, dpp_val<0x4E>(v), dpp_i32<0x4E>(idx));     // quad_perm [2,3,0,1]
    argmax_combine(v, idx, dpp_val<0x141>(v), dpp_i32<0x141>(idx));   // row_half_mirror
    argmax_combine(v, idx, dpp_val<0x140>(v), dpp_i32<0x140>(idx));   // row_mirror
    if constexpr (sizeof(T) == 4) {
        {
            const int oi = dpp_bcast_i32<0x142, 0xA>(0x7fffffff, idx);                                    // rows 1, 3 <- lane 15 of rows 0, 2
            const T ov = __builtin_bit_cast(T, dpp_bcast_i32<0x142, 0xA>(0, __builtin_bit_cast(int, v)));
            argmax_combine(v, idx, ov, oi);
        }
        {
            const int oi = dpp_bcast_i32<0x143, 0xC>(0x7fffffff, idx);                                    // rows 2, 3 <- lane 31
            const T ov = __builtin_bit_cast(T, dpp_bcast_i32<0x143, 0xC>(0, __builtin_bit_cast(int, v)));
            argmax_combine(v, idx, ov, oi);
        }
    } else {
        {
            const int oi = dpp_bcast_i32<0x142, 0xA>(0x7fffffff, idx);
            const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
            const unsigned lo = (unsigned)dpp_bcast_i32<0x142, 0xA>(0, (int)(unsigned)b), hi = (unsigned)dpp_bcast_i32<0x142, 0xA>(0, (int)(unsigned)(b >> 32));
            argmax_combine(v, idx, __builtin_bit_cast(T, ((unsigned long long)hi << 32) | lo), oi);
        }
        {
            const int oi = dpp_bcast_i32<0x143, 0xC>(0x7fffffff, idx);
            const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
            const unsigned lo = (unsigned)dpp_bcast_i32<0x143, 0xC>(0, (int)(unsigned)b), hi = (unsigned)dpp_bcast_i32<0x143, 0xC>(0, (int)(unsigned)(b >> 32));
            argmax_combine(v, idx, __builtin_bit_cast(T, ((unsigned long long)hi << 32) | lo), oi);
        }
    }
}

// Wave-wide maximum without LDS-crossbar shuffles (four in-row DPP steps, row_bcast:15, row_bcast:31); the result is valid
// in LANE 63 ONLY and is returned from there through readlane.  Inputs must not be NaN.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float dpp_keep_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), CTRL, ROWMASK, 0xF, false));
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_keep_f64(double v) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)b, (int)(unsigned)b, CTRL, ROWMASK, 0xF, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)(b >> 32), (int)(unsigned)(b >> 32), CTRL, ROWMASK, 0xF, false);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ float wave_max_f32(float v) {
    v = fmaxf(v, dpp_keep_f32<0xB1, 0xF>(v));
    v = fmaxf(v, dpp_keep_f32<0x4E, 0xF>(v));
    v = fmaxf(v, dpp_keep_f32<0x141, 0xF>(v));
    v = fmaxf(v, dpp_keep_f32<0x140, 0xF>(v));
    v = fmaxf(v, dpp_keep_f32<0x142, 0xA>(v));
    v = fmaxf(v, dpp_keep_f32<0x143, 0xC>(v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ double wave_max_f64(double v) {
    v = fmax(v, dpp_keep_f64<0xB1, 0xF>(v));
    v = fmax(v, dpp_keep_f64<0x4E, 0xF>(v));
    v = fmax(v, dpp_keep_f64<0x141, 0xF>(v));
    v = fmax(v, dpp_keep_f64<0x140, 0xF>(v));
    v = fmax(v, dpp_keep_f64<0x142, 0xA>(v));
    v = fmax(v, dpp_keep_f64<0x143, 0xC>(v));
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, 63), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), 63);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// ---- k-in-a-row through a cell: tictactoe.py:54-79, connect4.py:73-98, gomoku.py:66-91 ----------
// 1 + run(+dir) + run(-dir) >= K for any of the four directions; the origin cell itself is NOT
// tested (the reference's counter starts at 1).  Lanes 0..7 each walk one (direction, side).
__device__ __forceinline__ int azk_check_winner(const uint8_t *b, const GameDesc &g, int player, int cell) {
    const int lane = azk_lane();
    // lanes 0..7: one (direction, side) each.  The up to win_len - 1 <= 4 cells of the walk are fetched together (one LDS round trip,
    // not one per step) and the run is the number of leading own stones.
    const int d = (lane >> 1) & 3, sg = (lane & 1) ? -1 : 1;
    const int dr = (d > 0 ? 1 : 0) * sg;
    const int dc = (d == 0 ? 1 : (d == 1 ? 0 : (d == 2 ? 1 : -1))) * sg;
    const int r0 = (int)(((unsigned)cell * g.inv_cols) >> 16), c0 = cell - r0 * g.cols;
    uint8_t v[4];
    bool in[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int r = r0 + dr * (k + 1), c = c0 + dc * (k + 1);
        in[k] = k + 1 < g.win_len && r >= 0 && r < g.rows && c >= 0 && c < g.cols;
        v[k] = b[in[k] ? r * g.cols + c : 0];
    }
    int cnt = 0;
    bool run = lane < 8;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        run = run && in[k] && ((v[k] >> player) & 1);
        cnt += run ? 1 : 0;
    }
    const int other = dpp_i32<0xB1>(cnt);                              // the opposite side of the same direction (quad_perm [1,0,3,2])
    const bool win = lane < 8 && (1 + cnt + other >= g.win_len);
    return __ballot(win) != 0ull ? player : -1;
}

// ---- CPython 3.10 tuple hash of (r, c) (Objects/tupleobject.c) -----------------------------------
__device__ __forceinline__ unsigned long long py_tuple2_hash(int r, int c) {
    const unsigned long long P1 = 11400714785074694791ULL, P2 = 14029467366897019727ULL, P5 = 2870177450012600261ULL;
    unsigned long long acc = P5;
    acc += (unsigned long long)r * P2; acc = (acc << 31) | (acc >> 33); acc *= P1;
    acc += (unsigned long long)c * P2; acc = (acc << 31) | (acc >> 33); acc *= P1;
    acc += 2ULL ^ (P5 ^ 3527539ULL);
    if (acc == ~0ULL) return 1546275796ULL;
    return acc;
}

// LDS scratch for move generation (pointers carved by the caller; all 16-byte aligned)
struct MoveScratch {
    uint32_t *bits;              // [ceil(rc*8/32)] first-adder key bitmap
    uint16_t *pref;              // [ceil(rc*8/32)] exclusive popcount prefix
    int16_t *ord;                // [rc] candidate cells in first-insertion order
    uint16_t *tabA, *tabB;       // [table_size] emulated set tables (cell+1, 0 = empty)
    uint32_t *claim;             // [table_size] per-slot lowest claiming lane of the current round (0xffffffff = none)
    int table_size;
};

// get_valid_moves in the reference's LIST ORDER; returns the count (wave-uniform); moves[] in LDS.
//   TicTacToe  tictactoe.py:82-83   empty cells, row-major
//   Connect4   connect4.py:44-53    per column with an empty top cell: (lowest empty row, col)
//   Gomoku     gomoku.py:93-106     empty 8-neighbours of any stone as list(set(...)): CPython set
//                                   iteration order; centre cell when there is no candidate.
// All lanes must call this (it synchronises the single-wave workgroup).
struct AzkNoHook { __device__ __forceinline__ void operator()() const {} };

// The two small games (all lanes must call; synchronises the single-wave workgroup).
__device__ __forceinline__ int azk_valid_moves_small(const uint8_t *b, const GameDesc &g, int16_t *moves) {
    const int lane = azk_lane();
    int n = 0;
    if (g.kind == AZK_KIND_TTT) {
        for (int base = 0; base < g.rc; base += AZK_WAVE) {
            int i = base + lane;
            bool emp = i < g.rc && b[i] == 0;
            unsigned long long m = __ballot(emp);
            if (emp) moves[n + __popcll(m & ((1ull << lane) - 1ull))] = (int16_t)i;
            n += __popcll(m);
        }
        __syncthreads();
        return n;
    }
    int cell = -1;                                                    // Connect4
    if (lane < g.cols && b[lane] == 0) {
        for (int row = g.rows - 1; row >= 0; row--)
            if (b[row * g.cols + lane] == 0) { cell = row * g.cols + lane; break; }
    }
    unsigned long long m = __ballot(cell >= 0);
    if (cell >= 0) moves[__popcll(m & ((1ull << lane) - 1ull))] = (int16_t)cell;
    __syncthreads();
    return __popcll(m);
}

// Gomoku.  `mid` is called exactly once, by all lanes, after the first-adder keys - a few thousand cycles into the function:
// k_tree looks at its eval-cache probe there, whose loads it issued before the call.
template <int KMAX = 7, typename Mid = AzkNoHook>                    // KMAX: cell groups of 64 the function is unrolled for (rc <= 64 KMAX)
__device__ int azk_valid_moves_gomoku(const uint8_t *b, const GameDesc &g, int16_t *moves, const MoveScratch &ms, bool skip_set = false,
                                      long long *dbgv = nullptr, Mid &&mid = Mid()) {
    const int lane = azk_lane();
    int n = 0;
    // ---- Gomoku ----
    // The workgroup is ONE wave: LDS instructions of a wave execute in issue order, so a write followed by another lane's read
    // needs no wait between them, only a compiler-level fence (azk_wave_sync).  Every wait in this function is a data wait.
    const int R = g.rows, C = g.cols, rc = g.rc;
    const int nwords = (rc * 8 + 31) >> 5;
    long long s0 = dbgv ? clock64() : 0, s1 = 0, s2 = 0, s3 = 0;
    // this lane's cells (lane + 64 k): board codes and columns once, straight-line (no load sits behind a branch);
    // e / C by multiplication: exact for e < 65536 / C
    const unsigned inv = g.inv_cols;
    uint8_t code[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; k++) {
        const int e = lane + AZK_WAVE * k;
        code[k] = b[e < rc ? e : 0];
    }
    for (int w = lane; w < nwords; w += AZK_WAVE) ms.bits[w] = 0u;
    // 1. per empty cell: key = (row-major index of the first stone that adds it) * 8 + (its slot in that stone's add
    //    order: (0,+1) (0,-1) (+1,0) (-1,0) (+1,+1) (-1,-1) (+1,-1) (-1,+1)), gomoku.py:97-102.  The smallest key belongs
    //    to the smallest stone index, i.e. row r-1 (c-1, c, c+1), then row r (c-1, c+1), then row r+1 (c-1, c, c+1).
    //    The board as a bit string in scalar registers (one ballot per 64 cells): "is there a stone at cell e + delta" for the
    //    cell of lane l is bit l of the string shifted by delta - a wave-uniform 64-bit word used directly as the select mask
    //    of v_cndmask.  Neighbours across a row end are cut by the column masks (left neighbours need c >= 1, right ones
    //    c <= C - 2); rows outside the board are zeros of the string.  No LDS round trip.  Needs C + 1 <= 63 (make_game: <= 30).
    unsigned long long S[KMAX + 1], CL[KMAX], CR[KMAX];
    bool empty[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; k++) {
        S[k] = 0ull; CL[k] = 0ull; CR[k] = 0ull; empty[k] = false;
        if (AZK_WAVE * k >= rc) continue;                             // wave-uniform
        const int e = lane + AZK_WAVE * k;
        const bool in = e < rc;
        const int r = (int)(((unsigned)(in ? e : 0) * inv) >> 16), c = (in ? e : 0) - r * C;
        S[k] = __ballot(in && (code[k] & 3));
        CL[k] = __ballot(c >= 1);
        CR[k] = __ballot(c <= C - 2);
        empty[k] = in && code[k] == 0;
    }
    S[KMAX] = 0ull;
    // value of a key relative to its cell: (stone index - cell) * 8 + slot, per neighbour; the select chain runs over these eight
    // constants (one register each, shared by all of a lane's cells) and the cell's own e * 8 is added once
    const int cm1 = C > 1 ? C - 1 : 1;                               // (C = 1: the diagonal masks are empty, the shift just has to be legal)
    const unsigned NONE = 0x7fffffffu;
    const unsigned v_dr = (unsigned)((C + 1) * 8 + 5), v_d = (unsigned)(C * 8 + 3), v_dl = (unsigned)((C - 1) * 8 + 7), v_r = 9u;
    const unsigned v_l = (unsigned)(-8), v_ur = (unsigned)(-(C - 1) * 8 + 6), v_u = (unsigned)(-C * 8 + 2), v_ul = (unsigned)(-(C + 1) * 8 + 4);
    unsigned key[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; k++) {
        key[k] = 0xffffffffu;
        if (AZK_WAVE * k >= rc) continue;                             // wave-uniform
        const unsigned long long below = k > 0 ? S[k - 1] : 0ull, here = S[k], above = S[k + 1];
        // bit l of up(a) = stone at cell (64 k + l) - a;  of down(d) = stone at cell (64 k + l) + d   (1 <= a, d <= 63)
        auto up = [&](int a) { return (here << a) | (below >> (64 - a)); };
        auto down = [&](int d) { return (here >> d) | (above << (64 - d)); };
        // select chain from the last candidate to the first (the first adder wins)
        unsigned kk = NONE;
        kk = azk_sel_mask(down(C + 1) & CR[k], v_dr, kk);             // (r+1, c+1) through (-1,-1)
        kk = azk_sel_mask(down(C), v_d, kk);                          // (r+1, c)   through (-1, 0)
        kk = azk_sel_mask(down(cm1) & CL[k], v_dl, kk);               // (r+1, c-1) through (-1,+1)
        kk = azk_sel_mask(down(1) & CR[k], v_r, kk);                  // (r, c+1)   through (0,-1)
        kk = azk_sel_mask(up(1) & CL[k], v_l, kk);                    // (r, c-1)   through (0,+1)
        kk = azk_sel_mask(up(cm1) & CR[k], v_ur, kk);                 // (r-1, c+1) through (+1,-1)
        kk = azk_sel_mask(up(C), v_u, kk);                            // (r-1, c)   through (+1, 0)
        kk = azk_sel_mask(up(C + 1) & CL[k], v_ul, kk);               // stone (r-1, c-1) adds e through (+1,+1)
        kk = (empty[k] && kk != NONE) ? (unsigned)(lane + AZK_WAVE * k) * 8u + kk : 0xffffffffu;
        if (kk != 0xffffffffu) atomicOr(&ms.bits[kk >> 5], 1u << (kk & 31));
        key[k] = kk;
    }
    azk_wave_sync();
    mid();
    if (dbgv) s1 = clock64();
    // 2. exclusive popcount prefix over the key bitmap (<= 128 words)
    int total = 0;
    for (int base = 0; base < nwords; base += AZK_WAVE) {
        const int w = base + lane;
        const int p = w < nwords ? __popc(ms.bits[w]) : 0;
        // inclusive wave scan on DPP (GFX9): row_shr 1,2,4,8 with zero fill, then row_bcast:15 into rows 1,3 and
        // row_bcast:31 into rows 2,3 - no LDS-crossbar shuffles
        int incl = p;
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, true);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, true);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, true);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, true);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xA, 0xF, false);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xC, 0xF, false);
        if (w < nwords) ms.pref[w] = (uint16_t)(total + incl - p);
        total += __builtin_amdgcn_readlane(incl, AZK_WAVE - 1);
    }
    azk_wave_sync();
    const int m = total;
    if (m == 0) {                                                 // gomoku.py:103-104
        if (lane == 0) moves[0] = (int16_t)((R / 2) * C + (C / 2));
        azk_wave_sync();
        return 1;
    }
    // 3. rank every candidate by its key -> first-insertion order (ord[] holds cell + 1)
    {
        unsigned pf[KMAX], bw[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; k++) {                            // straight-line LDS reads (invalid keys read word 0)
            const unsigned wd = key[k] != 0xffffffffu ? key[k] >> 5 : 0u;
            pf[k] = ms.pref[wd];
            bw[k] = ms.bits[wd];
        }
#pragma unroll
        for (int k = 0; k < KMAX; k++) {
            if (key[k] != 0xffffffffu) {
                const int rank = (int)pf[k] + __popc(bw[k] & ((1u << (key[k] & 31)) - 1u));
                ms.ord[rank] = (int16_t)(lane + AZK_WAVE * k + 1);
            }
        }
    }
    azk_wave_sync();
    if (skip_set) {   // timing experiment only: first-insertion order instead of CPython set order
        for (int i = lane; i < m; i += AZK_WAVE) moves[i] = (int16_t)(ms.ord[i] - 1);
        azk_wave_sync();
        return m;
    }
    if (dbgv) s2 = clock64();
    // 4. replay the inserts into the emulated CPython set (set_add_entry + set_table_resize; both probe slot i .. i+9 when
    //    i + 9 <= mask, else slot i alone, then i = i*5 + 1 + (perturb >>= 5)).
    //    4a. The 8-slot and the 32-slot generation (the first 5, then up to 19 keys - every position goes through them) run
    //        SEQUENTIALLY ON THE SCALAR UNIT: key j and its hash sit in lane j, the table in one register (lane s = slot s), the
    //        occupancy in a scalar 64-bit word; an insert is three v_readlane, a few scalar shifts / ff1 and one select -
    //        no LDS round trips, no claims, no retries.  (As parallel LDS rounds these two generations cost as many dependent
    //        LDS round trips as all later ones together.)
    constexpr int GEN2_FILL = 19;                                    // 32-slot table: resize once fill * 5 >= 31 * 3
    auto hash_cell1 = [&](int cell1) {                              // hash((r, c)) of the key "cell + 1"
        const int cell = cell1 > 0 ? cell1 - 1 : 0;
        const int hr = (int)(((unsigned)cell * inv) >> 16);
        return py_tuple2_hash(hr, cell - hr * C);
    };
    auto probe_scalar = [](unsigned long long h, unsigned msk, unsigned long long occ64) {   // all operands wave-uniform, msk <= 63
        unsigned long long perturb = h;
        unsigned i = (unsigned)h & msk;
        for (;;) {
            if (i + 9 <= msk) {
                const unsigned z = (unsigned)((~occ64) >> i) & 0x3ffu;
                if (z) return i + (unsigned)__ffs((int)z) - 1u;
            } else if (((occ64 >> i) & 1ull) == 0ull) return i;
            perturb >>= 5;
            i = (unsigned)((unsigned long long)i * 5 + 1 + perturb) & msk;
        }
    };
    const int n_small = m < GEN2_FILL ? m : GEN2_FILL;             // keys that enter the two small generations
    const int kreg = lane < n_small ? (int)ms.ord[lane] : 0;       // key j (cell + 1) in lane j
    const unsigned long long hreg = hash_cell1(kreg);
    unsigned long long nocc = ~0ull;                                // complement of the occupancy (one bit per slot), scalar
    unsigned msk_cur = 7u;
    int tidx = 0;                                                   // the table: lane s holds slot s as (list index of its key) + 1, 0 = empty
    // one insert: the first window of the probe sequence decides almost always (straight-line scalar code); the perturbed
    // sequence is the rare path.  The table stores the key's LIST INDEX, so a re-insertion finds the key's hash where the first
    // insertion did (lane j of hreg) - the tuple hash is computed once for both generations.
    auto insert_scalar = [&](int j) {
        const unsigned hl = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)hreg, j);
        const unsigned i = hl & msk_cur;
        const unsigned win = (msk_cur >= 9u && i + 9u <= msk_cur) ? 0x3ffu : 1u;
        const unsigned z = (unsigned)(nocc >> i) & win;
        unsigned slot;
        if (__builtin_expect(z == 0u, 0))
            slot = probe_scalar(((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(hreg >> 32), j) << 32) | hl, msk_cur, ~nocc);
        else slot = i + (unsigned)__builtin_ctz(z);
        const unsigned long long bit = 1ull << slot;
        nocc &= ~bit;
        tidx = (int)azk_sel_mask(bit, (unsigned)(j + 1), (unsigned)tidx);
    };
    const int n_gen1 = n_small < 5 ? n_small : 5;                  // 8-slot table: resize once fill * 5 >= 7 * 3
    for (int j = 0; j < n_gen1; j++) insert_scalar(j);
    if (n_gen1 == 5) {                                              // set_table_resize(so, 20): 32 slots, old table re-inserted in slot order
        const int oldt = tidx;
        unsigned long long old = ~nocc;
        nocc = ~0ull; tidx = 0; msk_cur = 31u;
        while (old) {
            const int s = __builtin_ctzll(old);
            old &= old - 1ull;
            insert_scalar(__builtin_amdgcn_readlane(oldt, s) - 1);
        }
        for (int j = 5; j < n_small; j++) insert_scalar(j);
    }
    const int key_of_slot = __shfl(kreg, tidx > 0 ? tidx - 1 : 0);  // (all lanes take part in the exchange)
    const int tkey = tidx ? key_of_slot : 0;                        // slot s -> its key (cell + 1)
    const unsigned long long occ64 = ~nocc;
    const int below_me = __popcll(occ64 & ((1ull << lane) - 1ull));
    if (m < GEN2_FILL) {                                            // the set never outgrew 32 slots: list(set) = slot order
        if (tkey) moves[below_me] = (int16_t)(tkey - 1);
        azk_wave_sync();
        if (dbgv && lane == 0) { s3 = clock64(); dbgv[0] += s1 - s0; dbgv[1] += s2 - s1; dbgv[2] += s3 - s2; dbgv[4] += m; }
        return m;
    }
    //    4b. From the 128-slot generation on: parallel rounds in LDS.  Up to 64 keys IN ORDER per round, one per lane.  Every
    //        lane probes for its key against the committed table; atomicMin claims on the chosen (first empty) slot expose
    //        clashes; the conflict-free PREFIX commits (a later key may only commit once every earlier key has, otherwise an
    //        earlier key re-probing could have reached its slot first); the rest retry next round.
    unsigned stamp = 0x00ffffffu;     // claims carry a per-round stamp that only decreases: a newer round's atomicMin always beats
                                      // whatever an older round left in the slot, so claims never need to be reset
    // occupancy bitmap of the CURRENT table (one bit per slot; the first-adder bitmap of steps 1-3 is free by now): a probe reads
    // two words of it instead of sixteen table entries
    uint32_t *occ = ms.bits;
    uint16_t *tab = ms.tabA, *other = ms.tabB;
    unsigned mask = 127;                                            // set_table_resize(so, 76)
    int fill = 0, pos = GEN2_FILL;
    int nlist = GEN2_FILL;                                          // re-insertions waiting in moves[0 .. nlist): the 32-slot table in slot order
    if (tkey) moves[below_me] = (int16_t)tkey;
    for (int i = lane; i < 128; i += AZK_WAVE) { tab[i] = 0; ms.claim[i] = 0xffffffffu; }
    if (lane <= 4) occ[lane] = 0u;
    azk_wave_sync();
    auto insert_batch = [&](uint16_t *tb, unsigned msk, const int16_t *list, int count) {
        for (int c0 = 0; c0 < count; c0 += AZK_WAVE) {
            const int idx = c0 + lane;
            const bool have = idx < count;
            const unsigned keyv = have ? (unsigned)list[idx] : 1u;   // cell + 1
            const unsigned long long h = hash_cell1((int)keyv);
            // the first window of the probe sequence is the same in every round (slots i0 .. i0+9, or i0 alone near the table's end):
            // its address, shift and width are fixed per key, and the round reads it straight-line
            const unsigned i0 = (unsigned)h & msk, w0 = i0 >> 5, sh0 = i0 & 31u;
            const unsigned win0 = (i0 + 9u <= msk) ? 0x3ffu : 1u;
            bool placed = !have;
            while (__ballot(!placed) != 0ull) {
                // the ten-slot window from the occupancy bitmap: two words cover bits i .. i+9
                const unsigned long long both0 = ((unsigned long long)occ[w0 + 1] << 32) | (unsigned long long)occ[w0];
                const unsigned z0 = (unsigned)((~both0) >> sh0) & win0;
                unsigned slot = i0 + (unsigned)__ffs((int)z0) - 1u;
                const bool deeper = !placed && z0 == 0u;              // the whole first window is taken: the perturbed sequence (rare)
                if (__ballot(deeper) != 0ull) {
                    if (deeper) {
                        unsigned long long perturb = h;
                        unsigned i = i0;
                        for (;;) {
                            perturb >>= 5;
                            i = (unsigned)((unsigned long long)i * 5 + 1 + perturb) & msk;
                            if (i + 9 <= msk) {
                                const unsigned w = i >> 5, sh = i & 31u;
                                const unsigned long long both = ((unsigned long long)occ[w + 1] << 32) | (unsigned long long)occ[w];
                                const unsigned z = (unsigned)((~both) >> sh) & 0x3ffu;
                                if (z) { slot = i + (unsigned)__ffs((int)z) - 1u; break; }
                            } else if (((occ[i >> 5] >> (i & 31u)) & 1u) == 0u) { slot = i; break; }
                        }
                    }
                }
                const unsigned mine = (stamp << 6) | (unsigned)lane;
                if (!placed) atomicMin(&ms.claim[slot], mine);
                azk_wave_sync();                                      // the claims are applied before the read below (issue order)
                const bool conflict = !placed && ms.claim[placed ? 0u : slot] != mine;
                const unsigned long long cb = __ballot(conflict);
                const int first_bad = cb ? __ffsll((long long)cb) - 1 : AZK_WAVE;
                if (!placed && lane < first_bad) { tb[slot] = (uint16_t)keyv; atomicOr(&occ[slot >> 5], 1u << (slot & 31u)); placed = true; }
                stamp--;
                azk_wave_sync();
            }
        }
    };
    // the keys of one table generation are inserted as ONE ordered batch: [re-insertions in old-table order] + [new
    // candidates up to the next resize threshold]
    while (pos < m || nlist > 0) {
        const int thr = (int)((mask * 3u + 4u) / 5u);             // smallest fill with fill*5 >= mask*3
        int cnt = m - pos;
        if (cnt > thr - (fill + nlist)) cnt = thr - (fill + nlist);
        if (cnt < 0) cnt = 0;
        for (int i = lane; i < cnt; i += AZK_WAVE) moves[nlist + i] = ms.ord[pos + i];
        azk_wave_sync();
        insert_batch(tab, mask, moves, nlist + cnt);
        fill += nlist + cnt; pos += cnt; nlist = 0;
        if ((unsigned)fill * 5u >= mask * 3u) {                   // set_table_resize(so, used * 4)
            unsigned newsize = 8;
            const unsigned minused = (unsigned)fill * 4u;
            while (newsize <= minused) newsize <<= 1;
            for (unsigned base = 0; base <= mask; base += AZK_WAVE) {   // old table in slot order = re-insertion order
                const unsigned i = base + lane;
                const uint16_t kv = i <= mask ? tab[i] : (uint16_t)0;
                const unsigned long long bm = __ballot(kv != 0);
                if (kv) moves[nlist + __popcll(bm & ((1ull << lane) - 1ull))] = (int16_t)kv;   // `moves` doubles as the scratch list
                nlist += __popcll(bm);
            }
            for (unsigned i = lane; i < newsize; i += AZK_WAVE) {       // the table the set grows into starts empty; new claim slots start free
                other[i] = 0;
                if (i > mask) ms.claim[i] = 0xffffffffu;
                if (i <= (newsize >> 5)) occ[i] = 0u;
            }
            azk_wave_sync();
            uint16_t *tmp = tab; tab = other; other = tmp;
            mask = newsize - 1;
            fill = 0;                                             // the new table is empty until the batch above re-inserts
        }
    }
    if (dbgv) s3 = clock64();
    // 5. list(set): table order
    if (mask == 127u) {                                               // the common table: both halves' reads in flight together
        const uint16_t k0 = tab[lane], k1 = tab[AZK_WAVE + lane];
        const unsigned long long b0 = __ballot(k0 != 0), b1 = __ballot(k1 != 0);
        const unsigned long long below = (1ull << lane) - 1ull;
        const int n0 = __popcll(b0);
        if (k0) moves[__popcll(b0 & below)] = (int16_t)(k0 - 1);
        if (k1) moves[n0 + __popcll(b1 & below)] = (int16_t)(k1 - 1);
        n = n0 + __popcll(b1);
    } else
    for (unsigned base = 0; base <= mask; base += AZK_WAVE) {
        const unsigned i = base + lane;
        const uint16_t kv = i <= mask ? tab[i] : (uint16_t)0;
        const unsigned long long bm = __ballot(kv != 0);
        if (kv) moves[n + __popcll(bm & ((1ull << lane) - 1ull))] = (int16_t)(kv - 1);
        n += __popcll(bm);
    }
    azk_wave_sync();
    if (dbgv && lane == 0) { dbgv[0] += s1 - s0; dbgv[1] += s2 - s1; dbgv[2] += s3 - s2; dbgv[3] += clock64() - s3; dbgv[4] += m; }
    return n;
}

__device__ __forceinline__ int azk_valid_moves(const uint8_t *b, const GameDesc &g, int16_t *moves, const MoveScratch &ms) {
    return g.kind == AZK_KIND_GOMOKU ? azk_valid_moves_gomoku(b, g, moves, ms) : azk_valid_moves_small(b, g, moves);
}

// ---- deterministic float32 exp shared bit-for-bit with oracle/az_oracle.c (azo_exp_det) --------------
__device__ __forceinline__ double azk_exp_det64(double x) {
    const double LOG2E = 1.4426950408889634, LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    if (x > 700.0) return __builtin_huge_val();
    if (x < -740.0) return 0.0;
    double n = __builtin_rint(x * LOG2E);
    double r = __builtin_fma(-n, LN2_HI, x);
    r = __builtin_fma(-n, LN2_LO, r);
    double p = 1.0 / 6227020800.0;
    p = __builtin_fma(p, r, 1.0 / 479001600.0);
    p = __builtin_fma(p, r, 1.0 / 39916800.0);
    p = __builtin_fma(p, r, 1.0 / 3628800.0);
    p = __builtin_fma(p, r, 1.0 / 362880.0);
    p = __builtin_fma(p, r, 1.0 / 40320.0);
    p = __builtin_fma(p, r, 1.0 / 5040.0);
    p = __builtin_fma(p, r, 1.0 / 720.0);
    p = __builtin_fma(p, r, 1.0 / 120.0);
    p = __builtin_fma(p, r, 1.0 / 24.0);
    p = __builtin_fma(p, r, 1.0 / 6.0);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    int ni = (int)n, h = ni / 2;
    double s1 = __longlong_as_double((long long)(1023 + h) << 52);
    double s2 = __longlong_as_double((long long)(1023 + (ni - h)) << 52);
    return p * s1 * s2;
}

__device__ __forceinline__ float azk_exp_det(float x) { return (float)azk_exp_det64((double)x); }

// numpy's float32 pairwise summation (numpy/core/src/umath/loops_utils.h.src), n <= 512.
// a[] in LDS.  All lanes call; returns the sum on every lane.
// Recursion: n <= 128 -> one 8-accumulator block; else split at n2 = n/2 - (n/2)%8 and recurse: at most four leaf blocks.
// Lanes 8q .. 8q+7 own block q: lane j runs accumulator r[j] (a[j] + a[j+8] + ... in index order), the combine
// ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) is three DPP exchanges inside the group of eight (float addition commutes, so either side of a
// pair computes the same bits), the block's last len % 8 elements are added in order, and the block sums meet through v_readlane.
// No LDS writes and no single-lane epilogue.
__device__ float azk_pairwise_sum(const float *a, int n) {
    const int lane = azk_lane();
    // leaf blocks (start, len); a block with len 0 does not exist
    int s0 = 0, l0 = n, s1 = 0, l1 = 0, s2 = 0, l2 = 0, s3 = 0, l3 = 0;
    if (n > 128) {
        int n2 = n / 2; n2 -= n2 % 8;
        int ls = 0, ll = n2, rs = n2, rl = n - n2;               // left / right halves
        s0 = ls; l0 = ll; s2 = rs; l2 = rl;
        if (ll > 128) { int h = ll / 2; h -= h % 8; l0 = h; s1 = ls + h; l1 = ll - h; }
        if (rl > 128) { int h = rl / 2; h -= h % 8; l2 = h; s3 = rs + h; l3 = rl - h; }
    }
    const int q = (lane >> 3) & 3, j = lane & 7;                      // (lanes 32..63 repeat the blocks: their results are not read)
    const int ms = q == 0 ? s0 : (q == 1 ? s1 : (q == 2 ? s2 : s3));
    const int ml = q == 0 ? l0 : (q == 1 ? l1 : (q == 2 ? l2 : l3));
    const float *p = a + ms;
    const int lim = ml - (ml % 8);                                    // elements that go through the accumulators (0 when ml < 8)
    // a leaf block has at most 128 elements: 16 strided terms per accumulator and up to 7 in the tail, all loads in flight
    // before the first add (indices clamped into the block; a block of length 0 reads a[0])
    float v[16], tl[7];
#pragma unroll
    for (int u = 0; u < 16; u++) v[u] = p[(8 * u < lim ? 8 * u : 0) + (lim > 0 ? j : 0)];
#pragma unroll
    for (int t = 0; t < 7; t++) tl[t] = p[lim + t < ml ? lim + t : 0];
    float r = v[0];
#pragma unroll
    for (int u = 1; u < 16; u++) r = (8 * u < lim) ? r + v[u] : r;
    r = r + dpp_f32<0xB1>(r);                                         // r0+r1 | r2+r3 | r4+r5 | r6+r7      (quad_perm [1,0,3,2])
    r = r + dpp_f32<0x4E>(r);                                         // (r0+r1)+(r2+r3) | (r4+r5)+(r6+r7)  (quad_perm [2,3,0,1])
    r = r + dpp_f32<0x141>(r);                                        // both halves of the eight            (row_half_mirror)
    float sblk = lim > 0 ? r : 0.f;                                   // numpy: n < 8 starts from 0. and adds the elements in order
#pragma unroll
    for (int t = 0; t < 7; t++) sblk = (lim + t < ml) ? sblk + tl[t] : sblk;
    const float b0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sblk), 0));
    const float b1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sblk), 8));
    const float b2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sblk), 16));
    const float b3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sblk), 24));
    float left = b0;
    if (l1 > 0) left = left + b1;
    float res = left;
    if (l2 > 0) {
        float right = b2;
        if (l3 > 0) right = right + b3;
        res = left + right;
    }
    return res;
}
