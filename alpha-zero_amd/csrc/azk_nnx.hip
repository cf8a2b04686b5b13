// azk_nnx.hip - the fp32-ACCURATE form of the policy-value network's folded cls path (ai/nn.py:5-84 for the row the heads
// read), hand-written for gfx950.  The reference evaluates its network in float32 (nn.py:74-84, mcts.py:46-49); the bf16
// kernels of azk_nn.hip move a few visits per search, so this file holds the same two stages in
// arithmetic that keeps float32 accuracy end to end:
//
//   k_embed_pool_x   boards -> pooled tokens z[n][H][512] (float32).  The structure of k_embed_pool_c (only the tokens a stone
//                    can reach are evaluated, the others enter as precomputed constants; boards pulled from a device queue).
//                    * conv + folded head scores + row mean: the inputs are 0/1, so with the weights split into TWO fp16 terms
//                      (w S = hi + lo, 22 significant bits) every product is exact and v_mfma_f32_16x16x32_f16 accumulates them
//                      in float32 - the im2col GEMM at the fp16 rate, float32 results;
//                    * LayerNorm statistics, normalisation, softmax weights (exp with an extended-precision argument): float32 VALU;
//                    * weighted token sum Z += W^T Xn - Wc^T Xnc: both operands are run-time float32, so it runs on
//                      v_mfma_f32_16x16x4_f32 (exact float32 fma chain), with both operands already in the accumulators' layout.
//   k_gemm_x         the cls-row tail as float32 GEMMs on v_mfma_f32_16x16x4_f32: per-head value projection -> output projection
//                    (+ row statistics) -> LayerNorm2 + MLP up + GELU (erff) -> MLP down + residual (+ row statistics) -> final
//                    LayerNorm + merged heads + tanh.  LayerNorm as in k_tail_gemm: the producer's epilogue leaves per-row partial
//                    (sum, sum of squares), the consumer normalises its A fragments on the fly; affines folded into the weights.
// Everything the host folds (cls query through W_k, LayerNorm affines, softmax constants) is computed in float64 and rounded once.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "azk.h"
#include "azk_launch.h"

namespace {

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

// Sum over the 16 lanes of a DPP row (lanes sharing lane>>4), result in every lane
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    return v;
}

// exp(x) for x <= ~80 with float32 accuracy: x log2(e) carried as hi + lo (the plain product loses |x| ulps of the argument,
// 5e-6 relative at x = -80), v_exp_f32 on hi, first-order correction for lo.
__device__ __forceinline__ float exp_acc(float x) {
    const float L2E_HI = 1.44269502162933349609375f, L2E_LO = 1.92596299112661746e-8f;
    const float hi = x * L2E_HI;
    const float lo = __builtin_fmaf(x, L2E_HI, -hi) + x * L2E_LO;
    const float r = __builtin_amdgcn_exp2f(hi);
    return __builtin_fmaf(r, lo * 0.693147180559945309f, r);
}

// =====================================================================================================
// k_embed_pool_x
// =====================================================================================================
struct EmbedPoolXArgs {
    const void *boards;
    int boards_f32;
    const void *wt_frag;           // conv weight (+ 16 extra columns) x S as fp16 (hi, lo) in MFMA fragment order [33][KS][2][64 lanes] x 16 bytes
    const float *cposT;            // [T + 1][512]  bias + positional term per token; row T (the null token) = 0
    const float *scoreT;           // [T + 1][16]   score constants per token (column 15: row mean); row T: -1e30 in the head columns
    const float *wcT;              // [T + 1][16]   softmax weight of the token taken as an empty-patch token; row T = 0
    const void *xncT;              // [T + 1][128][8] fp16: normalised empty-patch token x 16 as (hi, lo) terms per 4 columns; row T = 0
    const unsigned *wcH;           // [T + 1][16]   -wc x 64 as (hi | lo << 16) fp16 terms
    const float *zall;             // accumulator order [8 waves][4][64 lanes][4]: ZALL[head 4 (lane>>4) + j][64 w + 4 (lane&15) + q]
    const float *lall;             // [16]
    const float *msum, *sref;      // [16]
    float *z;                      // [n][NH][512] float32
    const int *count;
    int *sched;                    // [1]: ticket counter of the board queue; zero between launches
    unsigned long long *wstats;    // optional [2]: boards / 16-token tiles evaluated
    int n, R, Cc, T;
    float eps, wscale_inv;         // 1 / S
    float pscale;                  // unit of zall and of the Z accumulators (64 x 16: the constant tokens' fp16 terms)
    azk_leaf_source src;
};

// One workgroup (four waves, one per SIMD; one workgroup per CU: the fp16 hi / lo weight image is 135 KB of LDS) per board; a
// WAVE owns whole 16-token tiles - all 512 columns of them - so LayerNorm's row statistics never leave the wave and the tile loop
// has no barrier: a tile is 132 fp16 MFMAs (33 column tiles x 2 k-steps x hi / lo) and 256 float32 MFMAs (32 column tiles x
// (4 real + 4 constant) token quads) issued back to back, ~10 k matrix-pipe cycles with the VALU work (bias / positional add,
// squares, normalisation: ~400 instructions) and the token-indexed gathers interleaved.  (Round 3's first form split the COLUMNS
// over eight waves and met at a barrier per tile for the sum of squares: 128 us per launch, four times the matrix-pipe time.)
// Tiles t = wave, wave + 4, ... of the board's compacted dirty-token list; each wave accumulates its own Z[8 heads][512] (128
// accumulator registers) and L, and the four partial sums meet in LDS at the end of the board, added in a fixed wave order
// (deterministic: the same board gives the same bits whatever else is in the batch).
// Column map: column tile ct = 4 g + q covers columns 64 g + 4 (lane&15) + q, so a lane's four accumulators of a token and a
// group g are four consecutive columns (one 16-byte gather / store).  Token patch bits, compaction and leaf ranks: one thread per
// token, exactly as in k_embed_pool_c.
template <int NC, int KSZ, int NH, bool SRC>
__global__ __launch_bounds__(256, 1) void k_embed_pool_x(EmbedPoolXArgs a) {
    constexpr int KS = (NC * KSZ * KSZ + 31) / 32;
    constexpr int D = 512;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4 *alut = (uint4 *)smem;                                  // [256] A fragment (fp16 0 / 1) of 8 patch bits
    const int Tp16 = ((a.T + 15) >> 4) << 4;
    uint4 *pbits = alut + 256;                                    // [Tp16] patch bits of the compacted dirty tokens
    int *dlist = (int *)(pbits + Tp16);                           // [Tp16] their token indices (null token = T past the end)
    int *scan = dlist + Tp16;                                     // [4..7] dirty counts per wave, [8] next board, [9] game, [16..31] class totals
    float *lred = (float *)(scan + 32);                           // [4 waves][16] partial softmax denominators
    f32x4 *zred = (f32x4 *)(lred + 64);                           // [3 waves][8 column tiles][32 lanes] partial Z of one column quarter
    uint4 *rankv = (uint4 *)(zred + 3 * 8 * 32);                  // SRC: [256 threads] ranks of the thread's first eight games, 16 bits each
    uint4 *bimg = rankv + (SRC ? 256 : 0);                        // [33 column tiles][KS][2][64 lanes] weight B fragments

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    int nvalid, my_lo = 0, my_per = 0;
    unsigned long long cb_lo = 0ull, cb_hi = 0ull;
    constexpr int NF = 33 * KS * 2 * 64, PER = (NF + 255) / 256;
    unsigned long long myflags = 0ull;
    if (SRC) {
        my_per = ((((a.src.n_games + 255) >> 8) + 7) >> 3) << 3;
        my_lo = tid * my_per;
        if (my_lo < a.src.flag_bytes) myflags = *(const unsigned long long *)(a.src.leaf_flag + my_lo);
    }
    static_assert(NF % 64 == 0, "the weight image is copied in whole 1 KiB wave pieces");
#pragma unroll
    for (int i = 0; i < PER; i++)
        if (256 * i + 64 * wave < NF)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((const uint4 *)a.wt_frag + tid + 256 * i),
                                             (__attribute__((address_space(3))) void *)(bimg + 256 * i + 64 * wave), 16, 0, 0);
    if (SRC) {
        // leaf ranks: (cost class descending, game index ascending), every workgroup derives the same ones - see k_embed_pool_c
        unsigned long long c_lo = 0ull, c_hi = 0ull;
        for (int w = 0; w < my_per; w += 8)
            if (my_lo + w < a.src.flag_bytes) {
                const unsigned long long f = w == 0 ? myflags : *(const unsigned long long *)(a.src.leaf_flag + my_lo + w);
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const unsigned c = (unsigned)((f >> (8 * q)) & 0xffull);
                    if (c) { if (c <= 4) c_lo += 1ull << (16 * (c - 1)); else c_hi += 1ull << (16 * (c - 5)); }
                }
            }
        unsigned long long i_lo = c_lo, i_hi = c_hi;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned long long v_lo = __shfl_up(i_lo, off), v_hi = __shfl_up(i_hi, off);
            if (lane >= off) { i_lo += v_lo; i_hi += v_hi; }
        }
        unsigned long long *wtot = (unsigned long long *)(scan + 16);           // [4 waves][2]
        if (lane == 63) { wtot[2 * wave] = i_lo; wtot[2 * wave + 1] = i_hi; }
        __syncthreads();
        unsigned long long b_lo = 0ull, b_hi = 0ull, t_lo = 0ull, t_hi = 0ull;
        for (int w = 0; w < 4; w++) {
            if (w < wave) { b_lo += wtot[2 * w]; b_hi += wtot[2 * w + 1]; }
            t_lo += wtot[2 * w]; t_hi += wtot[2 * w + 1];
        }
        const unsigned long long e_lo = b_lo + i_lo - c_lo, e_hi = b_hi + i_hi - c_hi;
        unsigned start = 0;
#pragma unroll
        for (int c = 7; c >= 0; c--) {
            const unsigned tot = (unsigned)(((c < 4 ? t_lo : t_hi) >> (16 * (c & 3))) & 0xffffull);
            const unsigned long long cb = (unsigned long long)(start + (unsigned)(((c < 4 ? e_lo : e_hi) >> (16 * (c & 3))) & 0xffffull)) << (16 * (c & 3));
            if (c < 4) cb_lo |= cb; else cb_hi |= cb;
            start += tot;
        }
        nvalid = (int)start;
        unsigned run = 0;
        unsigned myrank[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const unsigned c = (unsigned)((myflags >> (8 * q)) & 0xffull);
            unsigned r = 0xffffu;
            if (c) {
                const unsigned bsel = (unsigned)(((c <= 4 ? cb_lo : cb_hi) >> (16 * ((c - 1) & 3))) & 0xffffull);
                r = bsel + ((run >> (4 * (c - 1))) & 0xfu);
                run += 1u << (4 * (c - 1));
            }
            myrank[q >> 1] = (q & 1) ? ((myrank[q >> 1] & 0x0000ffffu) | (r << 16)) : ((myrank[q >> 1] & 0xffff0000u) | r);
        }
        rankv[tid] = make_uint4(myrank[0], myrank[1], myrank[2], myrank[3]);
        if (blockIdx.x == 0 && tid == 0) { *a.src.n_leaf = nvalid; if (a.src.cache_stamp) *a.src.cache_stamp += 1u; }
    } else {
        nvalid = a.count ? min(a.n, *a.count) : a.n;
    }
    int board = blockIdx.x;
    int ws_boards = 0, ws_tiles = 0;
    if (board >= nvalid) __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): no LDS-DMA may outlive the workgroup
    if (board < nvalid) {
    union BF { uint4 u; f16x8 v; };
    const uint4 *bfr = bimg + lane;                               // fragment (ct, s, p) at bfr[((ct * KS + s) * 2 + p) * 64]
    {
        unsigned r[4];
#pragma unroll
        for (int i = 0; i < 4; i++) r[i] = (((tid >> (2 * i)) & 1) ? 0x3C00u : 0u) | (((tid >> (2 * i + 1)) & 1) ? 0x3C000000u : 0u);   // fp16 1.0
        alut[tid] = make_uint4(r[0], r[1], r[2], r[3]);
    }
    constexpr int ksz = KSZ, kk = KSZ * KSZ, pad = KSZ / 2;
    const int RC = a.R * a.Cc, T = a.T, ncell = NC * RC;
    const float msum = a.msum[l15], sref = a.sref[l15], lall = a.lall[l15];
    const int colofs = 4 * l15;
    const float sinv = a.wscale_inv;
    int nxt = 0;
    __syncthreads();

    while (board < nvalid) {
        int tv = tid;
        asm volatile("" : "+v"(tv));
        const int lane_b = tv & 63;
        const int tj = tv - 1, tr = tj / a.Cc, tc = tj - tr * a.Cc;
        const bool tlive = tv >= 1 && tv < T;
        unsigned colmask = 0;
#pragma unroll
        for (int kx = 0; kx < ksz; kx++) { const int cc = tc + kx - pad; if (cc >= 0 && cc < a.Cc) colmask |= 1u << kx; }
        int game = 0, player = 0;
        if (SRC) {
            int g = -1;
            const uint4 rk = rankv[tid];
            const unsigned myrank[4] = {rk.x, rk.y, rk.z, rk.w};
#pragma unroll
            for (int q = 0; q < 8; q++) if (((myrank[q >> 1] >> (16 * (q & 1))) & 0xffffu) == (unsigned)board) g = my_lo + q;
            if (my_per > 8) {
                unsigned long long run2 = 0ull;
                for (int w = 0; w < my_per; w++) {
                    const unsigned c = my_lo + w < a.src.flag_bytes ? (unsigned)a.src.leaf_flag[my_lo + w] : 0u;
                    if (!c) continue;
                    const unsigned r = (unsigned)(((c <= 4 ? cb_lo : cb_hi) >> (16 * ((c - 1) & 3))) & 0xffffull) + (unsigned)((run2 >> (8 * (c - 1))) & 0xffull);
                    run2 += 1ull << (8 * (c - 1));
                    if (w >= 8 && r == (unsigned)board) g = my_lo + w;
                }
            }
            if (g >= 0) { scan[9] = g; a.src.leaf_slot[g] = board; }
            __syncthreads();
            game = scan[9];
        }
        unsigned long long plo = 0, phi = 0;
        {
            unsigned wbits = 0;                         // lane i holds bits [32 (i-1), 32 i) of the board bit string (lane 0: zeros)
            constexpr int NQ = 8;
            if (ncell <= 64 * NQ) {
                bool on[NQ];
                if (SRC) {
                    int code[NQ], chq[NQ];
                    const auto *cells = a.src.leaf_cells + (size_t)game * a.src.rc_pad;
#pragma unroll
                    for (int q = 0; q < NQ; q++) {
                        const int e = min(q * 64 + lane_b, ncell - 1);
                        chq[q] = (e >= RC) + (e >= 2 * RC);
                        code[q] = cells[(unsigned)(e - chq[q] * RC)];
                    }
                    player = (a.src.to_move[game] + a.src.leaf_depth[game]) & 1;
#pragma unroll
                    for (int q = 0; q < NQ; q++)
                        on[q] = q * 64 + lane_b < ncell && (chq[q] == 2 ? player != 0 : ((code[q] >> (chq[q] ^ player)) & 1) != 0);
                } else if (a.boards_f32) {
                    float raw[NQ];
                    const float *bp32 = (const float *)a.boards + (size_t)board * ncell;
#pragma unroll
                    for (int q = 0; q < NQ; q++) raw[q] = bp32[(unsigned)min(q * 64 + lane_b, ncell - 1)];
#pragma unroll
                    for (int q = 0; q < NQ; q++) on[q] = q * 64 + lane_b < ncell && raw[q] != 0.0f;
                } else {
                    unsigned short raw[NQ];
                    const unsigned short *bp16 = (const unsigned short *)a.boards + (size_t)board * ncell;
#pragma unroll
                    for (int q = 0; q < NQ; q++) raw[q] = bp16[(unsigned)min(q * 64 + lane_b, ncell - 1)];
#pragma unroll
                    for (int q = 0; q < NQ; q++) on[q] = q * 64 + lane_b < ncell && (raw[q] & 0x7fff) != 0;
                }
#pragma unroll
                for (int q = 0; q < NQ; q++) {
                    const unsigned long long m = __ballot(on[q]);
                    if ((lane_b - 1) >> 1 == q && lane_b >= 1) wbits = ((lane_b - 1) & 1) ? (unsigned)(m >> 32) : (unsigned)m;
                }
            } else {
                if (SRC) player = (a.src.to_move[game] + a.src.leaf_depth[game]) & 1;
                for (int q = 0; q * 64 < ncell; q++) {
                    const int e = q * 64 + lane;
                    bool on = false;
                    if (SRC) {
                        if (e < ncell) {
                            const int ch = (e >= RC) + (e >= 2 * RC), cell = e - ch * RC;
                            const int code = a.src.leaf_cells[(size_t)game * a.src.rc_pad + cell];
                            on = ch == 2 ? player != 0 : ((code >> (ch ^ player)) & 1) != 0;
                        }
                    } else if (e < ncell)
                        on = a.boards_f32 ? ((const float *)a.boards)[(size_t)board * ncell + e] != 0.0f
                                          : (((const unsigned short *)a.boards)[(size_t)board * ncell + e] & 0x7fff) != 0;
                    const unsigned long long m = __ballot(on);
                    if ((lane - 1) >> 1 == q && lane >= 1) wbits = ((lane - 1) & 1) ? (unsigned)(m >> 32) : (unsigned)m;
                }
            }
            // ---- patch bits of this thread's token; dirty = some stone in the patch ----
            unsigned lo[NC * KSZ], hi[NC * KSZ];
#pragma unroll
            for (int ch = 0; ch < NC; ch++)
#pragma unroll
                for (int ky = 0; ky < KSZ; ky++) {
                    const int rr = tr + ky - pad;
                    int off = 32 + ch * RC + (rr < 0 ? 0 : (rr >= a.R ? a.R - 1 : rr)) * a.Cc + (tc - pad);
                    if (!tlive) off = 32;
                    lo[ch * KSZ + ky] = __shfl(wbits, off >> 5); hi[ch * KSZ + ky] = __shfl(wbits, (off >> 5) + 1);
                }
#pragma unroll
            for (int ch = 0; ch < NC; ch++)
#pragma unroll
                for (int ky = 0; ky < KSZ; ky++) {
                    const int rr = tr + ky - pad;
                    int off = 32 + ch * RC + (rr < 0 ? 0 : (rr >= a.R ? a.R - 1 : rr)) * a.Cc + (tc - pad);
                    if (!tlive) off = 32;
                    unsigned bits = __funnelshift_r(lo[ch * KSZ + ky], hi[ch * KSZ + ky], off & 31) & colmask;
                    if (!tlive || rr < 0 || rr >= a.R) bits = 0;
                    const int p0 = ch * kk + ky * ksz;
                    if (p0 < 64) { plo |= (unsigned long long)bits << p0; if (p0 + ksz > 64) phi |= (unsigned long long)bits >> (64 - p0); }
                    else phi |= (unsigned long long)bits << (p0 - 64);
                }
        }
        const bool dirty = (plo | phi) != 0ull;
        const unsigned long long dm = __ballot(dirty);
        if (lane == 0) scan[4 + wave] = __popcll(dm);
        __syncthreads();                                  // (also: every wave is done with the previous board's lists and reduction buffers)
        int dpos = __popcll(dm & ((1ull << lane) - 1ull));
        for (int w = 0; w < wave; w++) dpos += scan[4 + w];
        const int nd = scan[4] + scan[5] + scan[6] + scan[7];
        const int ntile = (nd + 15) >> 4;
        if (dirty) {
            dlist[dpos] = tid;
            pbits[dpos] = make_uint4((unsigned)plo, (unsigned)(plo >> 32), (unsigned)phi, (unsigned)(phi >> 32));
        }
        if (tid < 16 && nd + tid < ntile * 16) { dlist[nd + tid] = T; pbits[nd + tid] = make_uint4(0u, 0u, 0u, 0u); }   // null tokens fill the last tile
        __syncthreads();
        ws_boards += 1; ws_tiles += ntile;                  // (one pair of atomics per workgroup at the very end, never in front of the gathers)
        if (tid == 0) {                                     // next board: the ticket's round trip hides under the tiles
            __builtin_amdgcn_sched_barrier(0);
            nxt = atomicAdd(a.sched, 1);
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- this wave's tiles: t = wave, wave + 4, ...; its own partial Z (8 groups x 4 column tiles) and L ----
        f32x4 Z[8][4];
#pragma unroll
        for (int g = 0; g < 8; g++)
#pragma unroll
            for (int q = 0; q < 4; q++) Z[g][q] = f32x4{0.f, 0.f, 0.f, 0.f};
        float L = 0.f;
        for (int tile = wave; tile < ntile; tile += 4) {
            const int4 tk = *(const int4 *)(dlist + 16 * tile + 4 * l4);
            const int tks[4] = {tk.x, tk.y, tk.z, tk.w};
            unsigned trow[4];                              // byte offset of this lane's 16 bytes in a token's 2 KB table row
#pragma unroll
            for (int r = 0; r < 4; r++) trow[r] = ((unsigned)tks[r] * (unsigned)D + (unsigned)colofs) * 4u;
            f32x4 scn, wcn;
            unsigned wch[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const unsigned osc = ((unsigned)tks[r] * 16u + (unsigned)l15) * 4u;
                scn[r] = *(const float *)((const char *)a.scoreT + osc);
                wcn[r] = *(const float *)((const char *)a.wcT + osc);
                wch[r] = *(const unsigned *)((const char *)a.wcH + osc);
            }
            // Token-indexed table rows (16 bytes per token and group of 64 columns) run through ONE stream of sixteen row sets - the
            // bias + positional rows of groups 0..7, then the constant tokens' normalised rows of groups 0..7 - in a ring of three
            // buffers: a set is requested when the set three places earlier has been consumed, so three gathers are always in flight
            f32x4 cb[3][4];
            auto gather_rows = [&](int j) {
                const char *tab = j < 8 ? (const char *)a.cposT : (const char *)a.xncT;
#pragma unroll
                for (int r = 0; r < 4; r++) cb[j % 3][r] = *(const f32x4 *)(tab + trow[r] + 256u * (unsigned)(j & 7));
            };
            gather_rows(0); gather_rows(1); gather_rows(2);
            // ---- A fragments: 8 patch bits of this lane's token (row lane&15) per k-step -> table (fp16 0 / 1) ----
            const uint4 pb = pbits[tile * 16 + l15];
            const unsigned pw[4] = {pb.x, pb.y, pb.z, pb.w};
            f16x8 afrag[KS];
#pragma unroll
            for (int s = 0; s < KS; s++) {
                BF af;
                af.u = alut[(pw[s] >> (8 * l4)) & 0xffu];
                afrag[s] = af.v;
            }
            // ---- x = conv / S + (bias + positional term) for all 32 column tiles (+ the extra tile: head scores, row mean); the hi and
            //      lo halves of the weights go through the same accumulator: every product is exact (0/1 inputs), the sum is float32.
            //      One wave per SIMD: nobody else hides the LDS latency of the weight fragments, so they run through a ring of three
            //      4-fragment steps issued TWO steps (eight MFMAs) ahead of their use; the four column tiles of a step are independent
            //      accumulator chains. ----
            constexpr int SPG = 2 * KS, NSTEP = 8 * SPG;          // steps (k-step, hi / lo) per group; steps per tile
            f32x4 acc[8][4];
            f32x4 acce = {0.f, 0.f, 0.f, 0.f};
            float ssq[4] = {0.f, 0.f, 0.f, 0.f};
            BF ring[3][4], ext[SPG];
            auto issue = [&](int t) {                             // step t = (group t / SPG, k-step (t % SPG) >> 1, half t & 1)
                const int g = t / SPG, sp = t % SPG;
#pragma unroll
                for (int q = 0; q < 4; q++) ring[t % 3][q].u = bfr[((4 * g + q) * SPG + sp) * 64];
            };
            auto conv_valu = [&](int g) {                         // bias / positional add and squares of group g
#pragma unroll
                for (int r = 0; r < 4; r++)
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const float x = __builtin_fmaf(acc[g][q][r], sinv, cb[g % 3][r][q]);
                        acc[g][q][r] = x;
                        ssq[r] = __builtin_fmaf(x, x, ssq[r]);
                    }
            };
#pragma unroll
            for (int sp = 0; sp < SPG; sp++) ext[sp].u = bfr[(32 * SPG + sp) * 64];
            issue(0); issue(1);
#pragma unroll
            for (int g = 0; g < 8; g++)
#pragma unroll
                for (int q = 0; q < 4; q++) acc[g][q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < NSTEP; t++) {
                const int g = t / SPG, sp = t % SPG;
                if (t + 2 < NSTEP) issue(t + 2);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 4; q++) acc[g][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(afrag[sp >> 1], ring[t % 3][q].v, acc[g][q], 0, 0, 0);
                if (t < SPG) acce = __builtin_amdgcn_mfma_f32_16x16x32_f16(afrag[t >> 1], ext[t].v, acce, 0, 0, 0);     // the extra tile rides along the first group
                if (sp == 0 && g > 0) {
                    conv_valu(g - 1);                             // (the previous group's MFMAs have retired, this group's are in flight)
                    gather_rows(g - 1 + 3);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            conv_valu(7);
            gather_rows(10);
            // ---- LayerNorm statistics of the full rows, inside the wave (mean = column 15 of the extra tile) ----
            float mean[4], rstd[4], shift[4], w[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                acce[r] = __builtin_fmaf(acce[r], sinv, scn[r]);
                mean[r] = __shfl(acce[r], (lane & 48) | 15);
                const float s2 = row16_sum(ssq[r]);
                const float var = __builtin_fmaf(-mean[r], mean[r], s2 * (1.0f / (float)D));
                rstd[r] = 1.0f / sqrtf(fmaxf(var, 0.f) + a.eps);
                shift[r] = -mean[r] * rstd[r];
                // scores (head = lane&15, tokens = rows) and softmax weights against the static reference
                const float sc = rstd[r] * __builtin_fmaf(-mean[r], msum, acce[r]);
                w[r] = exp_acc(sc - sref);
            }
            L += ((w[0] - wcn[0]) + (w[1] - wcn[1])) + ((w[2] - wcn[2]) + (w[3] - wcn[3]));
            // ---- Z += W^T Xn - Wc^T Xnc (in units of pscale).  The stone-touched tokens: both operands are run-time float32, so they
            //      run on v_mfma_f32_16x16x4_f32 - instruction r sums over token 4 (lane>>4) + r of every lane group, A[head = lane&15]
            //      [k = lane>>4] = w, B[k][column = lane&15] = xn, both already where they are.  The same tokens as CONSTANTS: -wc and xnc
            //      are tables, stored as (hi, lo) fp16 terms, and one v_mfma_f32_16x16x32_f16 holds a lane group's four tokens twice in its
            //      eight k-slots - B = (xh0..3, xl0..3) against A1 = (wh0..3, wh0..3) and against A2 = (wl0..3, wl0..3): all four partial
            //      products in two 16-cycle instructions instead of four 32-cycle ones ----
            f32x4 ws;
#pragma unroll
            for (int r = 0; r < 4; r++) ws[r] = w[r] * a.pscale;
            BF A1, A2;
            A1.u.x = __builtin_amdgcn_perm(wch[1], wch[0], 0x05040100u); A1.u.y = __builtin_amdgcn_perm(wch[3], wch[2], 0x05040100u);
            A1.u.z = A1.u.x; A1.u.w = A1.u.y;
            A2.u.x = __builtin_amdgcn_perm(wch[1], wch[0], 0x07060302u); A2.u.y = __builtin_amdgcn_perm(wch[3], wch[2], 0x07060302u);
            A2.u.z = A2.u.x; A2.u.w = A2.u.y;
#pragma unroll
            for (int g = 0; g < 8; g++) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 4; r++)
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const float xn = __builtin_fmaf(acc[g][q][r], rstd[r], shift[r]);       // (x - mean) * rstd
                        Z[g][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(ws[r], xn, Z[g][q], 0, 0, 0);
                    }
                // a gathered row = 16 bytes = (xh0 xh1 | xh2 xh3 | xl0 xl1 | xl2 xl3) of the token's four columns of this group
                uint4 rw[4];
#pragma unroll
                for (int r = 0; r < 4; r++) rw[r] = __builtin_bit_cast(uint4, cb[(8 + g) % 3][r]);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const unsigned sel = (q & 1) ? 0x07060302u : 0x05040100u;          // half q & 1 of each of two dwords
                    const unsigned h0 = q < 2 ? rw[0].x : rw[0].y, h1 = q < 2 ? rw[1].x : rw[1].y, h2 = q < 2 ? rw[2].x : rw[2].y, h3 = q < 2 ? rw[3].x : rw[3].y;
                    const unsigned l0 = q < 2 ? rw[0].z : rw[0].w, l1 = q < 2 ? rw[1].z : rw[1].w, l2 = q < 2 ? rw[2].z : rw[2].w, l3 = q < 2 ? rw[3].z : rw[3].w;
                    BF B1;
                    B1.u.x = __builtin_amdgcn_perm(h1, h0, sel); B1.u.y = __builtin_amdgcn_perm(h3, h2, sel);
                    B1.u.z = __builtin_amdgcn_perm(l1, l0, sel); B1.u.w = __builtin_amdgcn_perm(l3, l2, sel);
                    Z[g][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A1.v, B1.v, Z[g][q], 0, 0, 0);
                    Z[g][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A2.v, B1.v, Z[g][q], 0, 0, 0);
                }
                if (8 + g + 3 < 16) gather_rows(8 + g + 3);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- the four waves' partial sums meet: L first, then Z one column quarter at a time (12 KB of LDS), fixed order ----
        float Lt = L + __shfl_xor(L, 16);
        Lt += __shfl_xor(Lt, 32);
        if (l4 == 0) lred[wave * 16 + l15] = Lt;
        constexpr int HL = 16 * ((NH + 3) / 4);             // lanes holding real heads (lane>>4 < NH / 4)
        // the constant part of this wave's own column quarter (global, L2): requested now, used in its round below (a load inside
        // the round would put a memory round trip between two barriers, four times per board)
        f32x4 zq[2][4];
#pragma unroll
        for (int gg = 0; gg < 2; gg++)
#pragma unroll
            for (int q = 0; q < 4; q++) zq[gg][q] = *((const f32x4 *)a.zall + ((size_t)(2 * wave + gg) * 4 + q) * 64 + lane);
#pragma unroll
        for (int c = 0; c < 4; c++) {
            if (wave != c && lane < HL) {
                const int slot = wave < c ? wave : wave - 1;
#pragma unroll
                for (int gg = 0; gg < 2; gg++)
#pragma unroll
                    for (int q = 0; q < 4; q++) zred[(slot * 8 + gg * 4 + q) * 32 + lane] = Z[2 * c + gg][q];
            }
            __syncthreads();
            if (wave == c && lane < HL) {
                const float Ltot = lall + ((lred[l15] + lred[16 + l15]) + (lred[32 + l15] + lred[48 + l15]));
                float inv[4];
#pragma unroll
                for (int j = 0; j < 4; j++) inv[j] = 1.0f / (__shfl(Ltot, (4 * l4 + j) & 15) * a.pscale);      // (Z runs in units of pscale)
#pragma unroll
                for (int gg = 0; gg < 2; gg++) {
                    f32x4 zs[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        zs[q] = ((Z[2 * c + gg][q] + zred[(0 * 8 + gg * 4 + q) * 32 + lane]) + (zred[(1 * 8 + gg * 4 + q) * 32 + lane] + zred[(2 * 8 + gg * 4 + q) * 32 + lane])) + zq[gg][q];
                    }
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int head = 4 * l4 + j;
                        if (head < NH)
                            *(f32x4 *)(a.z + ((size_t)board * NH + head) * D + 64 * (2 * c + gg) + colofs) =
                                f32x4{zs[0][j] * inv[j], zs[1][j] * inv[j], zs[2][j] * inv[j], zs[3][j] * inv[j]};
                    }
                }
            }
            __syncthreads();
        }
        if (tid == 0) {
            if (nxt == nvalid - 1) a.sched[0] = 0;          // the holder of the last ticket leaves the queue zero for the next launch
            scan[8] = (int)gridDim.x + nxt;
        }
        __syncthreads();
        board = scan[8];
    }
    }
    if (a.wstats != nullptr && tid == 0 && ws_boards) { atomicAdd(a.wstats, (unsigned long long)ws_boards); atomicAdd(a.wstats + 1, (unsigned long long)ws_tiles); }
}

template <int NC, int KSZ, int NH, bool SRC>
int launch_embed_pool_x(const EmbedPoolXArgs &a, hipStream_t st) {
    constexpr int KS = (NC * KSZ * KSZ + 31) / 32;
    const int tp16 = ((a.T + 15) / 16) * 16;
    const int lds = 256 * 16 + tp16 * 16 + tp16 * 4 + 128 + 256 + 3 * 8 * 32 * 16 + (SRC ? 256 * 16 : 0) + 33 * KS * 2 * 64 * 16;     // 156 KB at KS = 2: one workgroup per CU
    if (lds > 160 * 1024) return AZK_ERR_ARG;
    if (azk_set_max_lds((const void *)k_embed_pool_x<NC, KSZ, NH, SRC>, lds) != hipSuccess) return AZK_ERR_HIP;
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    }
    const int blocks = a.n < cus ? a.n : cus;                      // one resident workgroup per CU; each pulls boards until the queue is dry
    k_embed_pool_x<NC, KSZ, NH, SRC><<<blocks, 256, lds, st>>>(a);
    return hipGetLastError() == hipSuccess ? AZK_OK : AZK_ERR_HIP;
}

}  // namespace

static int32_t embed_pool_x_impl(const void *boards_dev, int32_t boards_are_f32, const azk_leaf_source *src, const azk_embed_pool_x_consts *k,
                                 float *z_out_dev, int32_t n, int32_t channels, int32_t rows, int32_t cols,
                                 const int32_t *n_valid_dev, int32_t *sched_dev, void *stream) {
    if ((!boards_dev && !src) || !k || !z_out_dev || !sched_dev) return AZK_ERR_ARG;
    if (!k->wt_frag || !k->cpos_tok || !k->score_tok || !k->wconst_tok || !k->xnconst_tok || !k->z_all || !k->l_all || !k->score_msum || !k->score_ref) return AZK_ERR_ARG;
    if (!k->wconst_h16_tok || !(k->pool_scale > 0.f)) return AZK_ERR_ARG;
    const int ksize = k->ksize, kp = k->kp;
    if (n < 0 || channels < 1 || rows < 1 || cols < 1 || ksize < 1 || (ksize & 1) == 0 || ksize > 7) return AZK_ERR_ARG;
    if (kp != (channels * ksize * ksize + 31) / 32 * 32 || kp > 64) return AZK_ERR_ARG;     // the hi/lo image must fit one CU's LDS
    if (channels * rows * cols > 62 * 32 || k->embed_dim != 512 || !(k->wt_scale > 0.f)) return AZK_ERR_ARG;
    if (rows * cols + 1 > 256) return AZK_ERR_ARG;                 // one thread per token
    if (k->num_heads != 8 && k->num_heads != 4) return AZK_ERR_ARG;
    if (n == 0) return AZK_OK;
    EmbedPoolXArgs a;
    memset(&a, 0, sizeof a);
    a.boards = boards_dev; a.boards_f32 = boards_are_f32; a.wt_frag = k->wt_frag; a.cposT = k->cpos_tok;
    a.scoreT = k->score_tok; a.wcT = k->wconst_tok; a.xncT = k->xnconst_tok; a.zall = k->z_all; a.lall = k->l_all;
    a.msum = k->score_msum; a.sref = k->score_ref; a.z = z_out_dev; a.count = n_valid_dev; a.sched = sched_dev;
    a.wstats = (unsigned long long *)k->work_stats;
    a.n = n; a.R = rows; a.Cc = cols; a.T = rows * cols + 1; a.eps = k->ln_eps; a.wscale_inv = 1.0f / k->wt_scale;
    a.wcH = k->wconst_h16_tok; a.pscale = k->pool_scale;
    if (src) a.src = *src;
    hipStream_t st = (hipStream_t)stream;
    const int nh = k->num_heads;
#define CASE(NC_, KSZ_, NH_) if (channels == NC_ && ksize == KSZ_ && nh == NH_) \
        return src ? launch_embed_pool_x<NC_, KSZ_, NH_, true>(a, st) : launch_embed_pool_x<NC_, KSZ_, NH_, false>(a, st)
    CASE(2, 5, 8); CASE(2, 5, 4); CASE(2, 3, 8); CASE(2, 3, 4); CASE(3, 3, 8); CASE(3, 3, 4);
#undef CASE
    return AZK_ERR_ARG;
}

extern "C" int32_t azk_nnx_embed_pool(const void *boards_dev, int32_t boards_are_f32, const azk_embed_pool_x_consts *consts,
                                      float *z_out_f32_dev, int32_t n, int32_t channels, int32_t rows, int32_t cols,
                                      const int32_t *n_valid_dev, int32_t *sched_dev, void *stream) {
    if (!boards_dev) return AZK_ERR_ARG;
    return embed_pool_x_impl(boards_dev, boards_are_f32, nullptr, consts, z_out_f32_dev, n, channels, rows, cols, n_valid_dev, sched_dev, stream);
}

extern "C" int32_t azk_nnx_embed_pool_leaves(const azk_leaf_source *src, const azk_embed_pool_x_consts *consts, float *z_out_f32_dev,
                                             int32_t *sched_dev, void *stream) {
    if (!src || !src->leaf_flag || !src->leaf_cells || !src->to_move || !src->leaf_depth || !src->leaf_slot || !src->n_leaf) return AZK_ERR_ARG;
    if (src->n_games < 1 || src->rows * src->cols != src->rc || src->flag_bytes < src->n_games) return AZK_ERR_ARG;
    if (src->n_games > AZK_EMBED_POOL_COMPACT_MAX_SLOTS) return AZK_ERR_ARG;
    return embed_pool_x_impl(nullptr, 0, src, consts, z_out_f32_dev, src->n_games, src->planes, src->rows, src->cols, nullptr, sched_dev, stream);
}

// =====================================================================================================
// k_gemm_x: one link of the cls-row tail in float32 on v_mfma_f32_16x16x4_f32 (exact float32 fma chains).
//   C[m][nbatch * N] = op(A) W^T (+ bias) through the epilogues of k_tail_gemm, everything float32 in memory.
//   Wave tile 32 rows x 64 columns (2 x 4 MFMA tiles); a lane's A operand of a 16-wide k-step is ONE 16-byte load (row lane&15,
//   k = 16 s + 4 (lane>>4) .. +3: instruction i of the step takes component i, i.e. sums over k = 16 s + 4 g + i, g = 0..3), the
//   weights come packed the same way: Wp[N/64][K/16][4][64 lanes] x float4, element [g][s][c][lane][i] = W[64 g + 4 (lane&15) + c][16 s + 4 (lane>>4) + i]
//   (a lane's four accumulators of a row are four consecutive output columns: 16-byte stores).  The k loop streams 32 columns
//   of K per iteration, the next iteration's twelve loads in flight under the current 64 MFMAs (2 048 cycles).
//   NWK = 4: the four waves of a workgroup split K (the small GEMMs: enough waves to fill the chip, a quarter of the chain each),
//   partial sums meet in LDS, wave 0 runs the epilogue.  NWK = 1: the four waves are a 2 x 2 block of wave tiles (shared A rows and
//   weight fragments hit in L1).
// =====================================================================================================
namespace {

struct GemmXArgs {
    const float *A; int lda, a_batch;
    const f32x4 *Wp; long long w_batch;          // f32x4 elements between batches
    int M, N, nbatch;                            // N = output columns per batch (multiple of 64; of 128 for NWK = 1)
    const int *count;
    const float *bias;                           // [nbatch * N] or null
    float *out; int ldo;
    const float *resid; int ldr;
    float ln_eps;
    const float *stats_in;                       // AMODE 1: [M][8][2] partial (sum, sum of squares) of every A row (K = 512), left by the producer
    float *stats_out;                            // optional: this GEMM's own partials [M][nbatch * N / 64][2]
    float *logits, *values; int action_dim;
};

enum { X_EPI_PLAIN = 0, X_EPI_GELU = 1, X_EPI_RESID = 2, X_EPI_HEADS = 3 };

template <int EPI, int AMODE, int NWK, int KW>      // KW = columns of K per wave (K = KW * NWK)
__global__ __launch_bounds__(256, 1) void k_gemm_x(GemmXArgs a) {
    constexpr int RT = 2, K = KW * NWK, S16 = K / 16, NCH = KW / 32;
    __shared__ f32x4 kred[NWK > 1 ? (NWK - 1) * RT * 4 * 64 : 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, l4 = lane >> 4;
    const int nvalid = a.count ? min(a.M, *a.count) : a.M;
    const int wk = NWK > 1 ? wave : 0;
    const int wr = NWK > 1 ? 0 : (wave >> 1), wc = NWK > 1 ? 0 : (wave & 1);
    constexpr int WROWS = NWK > 1 ? 16 * RT : 32 * RT, WCOLS = NWK > 1 ? 64 : 128;
    const int rtiles = (nvalid + WROWS - 1) / WROWS, ctiles = a.N / WCOLS;
    const int nitems = rtiles * ctiles * a.nbatch;
    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        if (NWK > 1 && item != (int)blockIdx.x) __syncthreads();      // wave 0 is done with the previous item's partial sums
        const int ct = item % ctiles, r2 = item / ctiles, rt = r2 % rtiles, b = r2 / rtiles;
        const int row0 = rt * WROWS + wr * 16 * RT, g = ct * (WCOLS / 64) + wc;
        const float *ap[RT];
#pragma unroll
        for (int i = 0; i < RT; i++) ap[i] = a.A + (size_t)min(row0 + 16 * i + l15, a.M - 1) * a.lda + (size_t)b * a.a_batch + KW * wk + 4 * l4;
        const f32x4 *bp = a.Wp + (size_t)b * a.w_batch + ((size_t)g * S16 + (size_t)(KW / 16) * wk) * 4 * 64 + lane;
        f32x4 acc[RT][4];
#pragma unroll
        for (int i = 0; i < RT; i++)
#pragma unroll
            for (int c = 0; c < 4; c++) acc[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 st[AMODE == 1 ? RT : 1];
        if (AMODE == 1) {
#pragma unroll
            for (int i = 0; i < RT; i++) st[i] = *((const f32x4 *)(a.stats_in + (size_t)min(row0 + 16 * i + l15, a.M - 1) * 16) + l4);
        }
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        const int col0 = b * a.N + 64 * g + 4 * l15;
        if (a.bias) bv = *(const f32x4 *)(a.bias + col0);
        f32x4 af[2][RT][2], bf[2][2][4];                      // [buffer][row tile][k-step], [buffer][k-step][column tile]
        auto fetch = [&](int buf, int ch) {
#pragma unroll
            for (int s = 0; s < 2; s++) {
#pragma unroll
                for (int i = 0; i < RT; i++) af[buf][i][s] = *(const f32x4 *)(ap[i] + 32 * ch + 16 * s);
#pragma unroll
                for (int c = 0; c < 4; c++) bf[buf][s][c] = bp[((2 * ch + s) * 4 + c) * 64];
            }
        };
        fetch(0, 0);
        float rstd[RT], shift[RT];
        if (AMODE == 1) {                                     // the groups are added in a fixed order: deterministic, no atomics
#pragma unroll
            for (int i = 0; i < RT; i++) {
                float s1 = st[i][0] + st[i][2], s2 = st[i][1] + st[i][3];
                s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
                s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
                const float mean = s1 * (1.0f / 512.0f);
                rstd[i] = 1.0f / sqrtf(fmaxf(__builtin_fmaf(-mean, mean, s2 * (1.0f / 512.0f)), 0.f) + a.ln_eps);
                shift[i] = -mean * rstd[i];
            }
        }
#pragma unroll
        for (int ch = 0; ch < NCH; ch++) {
            const int cur = ch & 1;
            if (ch + 1 < NCH) fetch(cur ^ 1, ch + 1);
            __builtin_amdgcn_sched_barrier(0);                // the next iteration's loads are ISSUED here (left alone, hipcc sinks every load to
                                                              // just before its first use and waits for it there: one exposed round trip per k-step)
#pragma unroll
            for (int s = 0; s < 2; s++) {
                if (AMODE == 1) {
#pragma unroll
                    for (int i = 0; i < RT; i++)
#pragma unroll
                        for (int e = 0; e < 4; e++) af[cur][i][s][e] = __builtin_fmaf(af[cur][i][s][e], rstd[i], shift[i]);
                }
#pragma unroll
                for (int e = 0; e < 4; e++)
#pragma unroll
                    for (int i = 0; i < RT; i++)
#pragma unroll
                        for (int c = 0; c < 4; c++) acc[i][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[cur][i][s][e], bf[cur][s][c][e], acc[i][c], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (row0 >= nvalid) continue;                          // (uniform per wave tile; with NWK > 1 per workgroup)
        if (NWK > 1) {
            if (wave > 0) {
#pragma unroll
                for (int i = 0; i < RT; i++)
#pragma unroll
                    for (int c = 0; c < 4; c++) kred[((wave - 1) * RT * 4 + i * 4 + c) * 64 + lane] = acc[i][c];
            }
            __syncthreads();
            if (wave > 0) continue;
#pragma unroll
            for (int w = 1; w < NWK; w++)
#pragma unroll
                for (int i = 0; i < RT; i++)
#pragma unroll
                    for (int c = 0; c < 4; c++) acc[i][c] += kred[((w - 1) * RT * 4 + i * 4 + c) * 64 + lane];
        }
        f32x4 rr[EPI == X_EPI_RESID ? RT : 1][4];
        if (EPI == X_EPI_RESID) {
#pragma unroll
            for (int i = 0; i < RT; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) rr[i][j] = *(const f32x4 *)(a.resid + (size_t)min(row0 + 16 * i + 4 * l4 + j, a.M - 1) * a.ldr + col0);
        }
        if (EPI == X_EPI_HEADS) {                             // nn.py:82-83
#pragma unroll
            for (int i = 0; i < RT; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int row = row0 + 16 * i + 4 * l4 + j;
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const int col = col0 + c;
                        const float x = acc[i][c][j] + bv[c];
                        if (row < nvalid && col < a.action_dim) a.logits[(size_t)row * a.action_dim + col] = x;
                        if (row < nvalid && col == a.action_dim) a.values[row] = tanhf(x);
                    }
                }
        } else {
            const int ngr = a.nbatch * (a.N >> 6), gr = b * (a.N >> 6) + g;
#pragma unroll
            for (int i = 0; i < RT; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    f32x4 v = {acc[i][0][j] + bv[0], acc[i][1][j] + bv[1], acc[i][2][j] + bv[2], acc[i][3][j] + bv[3]};
                    if (EPI == X_EPI_GELU) {
#pragma unroll
                        for (int c = 0; c < 4; c++) v[c] = 0.5f * v[c] * (1.0f + erff(v[c] * 0.70710678118654752f));          // nn.GELU (erf form)
                    }
                    if (EPI == X_EPI_RESID) v += rr[i][j];
                    const int row = row0 + 16 * i + 4 * l4 + j;
                    if (row < nvalid) *(f32x4 *)(a.out + (size_t)row * a.ldo + col0) = v;
                    if (a.stats_out) {
                        const f32x2 ps = {row16_sum((v[0] + v[1]) + (v[2] + v[3])), row16_sum((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]))};
                        if (l15 == 0 && row < nvalid) *(f32x2 *)(a.stats_out + ((size_t)row * ngr + gr) * 2) = ps;
                    }
                }
        }
    }
}

template <int EPI, int AMODE, int NWK, int KW>
int launch_gemm_x(const GemmXArgs &a, hipStream_t st) {
    constexpr int WROWS = NWK > 1 ? 32 : 64, WCOLS = NWK > 1 ? 64 : 128;
    const long long items = (long long)((a.M + WROWS - 1) / WROWS) * (a.N / WCOLS) * a.nbatch;
    const unsigned blocks = (unsigned)(items < 16384 ? items : 16384);
    k_gemm_x<EPI, AMODE, NWK, KW><<<blocks, 256, 0, st>>>(a);
    return hipGetLastError() == hipSuccess ? AZK_OK : AZK_ERR_HIP;
}
}  // namespace

extern "C" int32_t azk_nnx_gemm(const azk_gemm_x *t, void *stream) {
    if (!t || !t->a_f32 || !t->w_packed || t->m < 0 || t->n_out < 64 || (t->n_out & 63) || t->nbatch < 1) return AZK_ERR_ARG;
    if ((t->k != 512 && t->k != 2048) || t->lda < t->k || (t->lda & 3) || (t->a_batch_stride & 3)) return AZK_ERR_ARG;
    if (t->epilogue < 0 || t->epilogue > 3 || (t->layernorm_a && (t->k != 512 || !t->a_stats))) return AZK_ERR_ARG;
    if (t->epilogue == X_EPI_HEADS ? (!t->logits_out || !t->values_out || t->action_dim + 1 > t->n_out * t->nbatch) : (!t->out_f32 || t->ldo < t->n_out * t->nbatch || (t->ldo & 3)))
        return AZK_ERR_ARG;
    if (t->epilogue == X_EPI_RESID && (!t->resid_f32 || (t->ldr & 3))) return AZK_ERR_ARG;
    if (t->m == 0) return AZK_OK;
    GemmXArgs a;
    memset(&a, 0, sizeof a);
    a.A = t->a_f32; a.lda = t->lda; a.a_batch = t->a_batch_stride; a.Wp = (const f32x4 *)t->w_packed;
    a.w_batch = (long long)(t->n_out / 64) * (t->k / 16) * 4 * 64;
    a.M = t->m; a.N = t->n_out; a.nbatch = t->nbatch; a.count = t->n_valid; a.bias = t->bias; a.out = t->out_f32; a.ldo = t->ldo;
    a.resid = t->resid_f32; a.ldr = t->ldr; a.ln_eps = t->ln_eps; a.logits = t->logits_out; a.values = t->values_out;
    a.action_dim = t->action_dim; a.stats_in = t->a_stats; a.stats_out = t->stats_out;
    hipStream_t st = (hipStream_t)stream;
    if (t->k == 2048) {
        if (t->layernorm_a) return AZK_ERR_ARG;
        if (t->epilogue == X_EPI_RESID) return launch_gemm_x<X_EPI_RESID, 0, 4, 512>(a, st);
        if (t->epilogue == X_EPI_PLAIN) return launch_gemm_x<X_EPI_PLAIN, 0, 4, 512>(a, st);
        return AZK_ERR_ARG;
    }
    const bool wide = t->n_out % 128 == 0 && t->n_out >= 1024;          // enough column groups to fill the chip without splitting K
    if (t->layernorm_a) {
        if (t->epilogue == X_EPI_GELU) return wide ? launch_gemm_x<X_EPI_GELU, 1, 1, 512>(a, st) : launch_gemm_x<X_EPI_GELU, 1, 4, 128>(a, st);
        if (t->epilogue == X_EPI_HEADS) return launch_gemm_x<X_EPI_HEADS, 1, 4, 128>(a, st);
        if (t->epilogue == X_EPI_PLAIN) return launch_gemm_x<X_EPI_PLAIN, 1, 4, 128>(a, st);
        return AZK_ERR_ARG;
    }
    if (t->epilogue == X_EPI_PLAIN) return launch_gemm_x<X_EPI_PLAIN, 0, 4, 128>(a, st);
    if (t->epilogue == X_EPI_GELU) return wide ? launch_gemm_x<X_EPI_GELU, 0, 1, 512>(a, st) : launch_gemm_x<X_EPI_GELU, 0, 4, 128>(a, st);
    if (t->epilogue == X_EPI_RESID) return launch_gemm_x<X_EPI_RESID, 0, 4, 128>(a, st);
    return launch_gemm_x<X_EPI_HEADS, 0, 4, 128>(a, st);
}

// =====================================================================================================
// k_gemm_h: the same links on the fp16 matrix pipe with every operand carried as TWO fp16 terms (x S = hi + lo: 22 significant
// bits, the products hi*hi + hi*lo + lo*hi exact in the float32 accumulator; the dropped lo*lo term is 2^-22 relative).
//   v_mfma_f32_16x16x4_f32 runs at the float32 VECTOR rate: the two wide links of the tail are 32.8 k matrix-pipe cycles per wave
//   on it (31 us each at the clock these loops hold).  Three v_mfma_f32_16x16x32_f16 per 32-wide k-step instead of eight f32
//   instructions of twice the length is 5.3x less pipe time at the same operand bytes (4 B per element either way).
//   Activations travel between the links as (hi, lo) planes written by the PRODUCING epilogue (four VALU per element, once),
//   scaled by 16; the first link reads the float32 z of k_embed_pool_x and splits on the fly (each element is read by one wave only).
//   Weights: (hi, lo) planes of w x 256 in fragment order Wp[N/64][K/32][4][2][64 lanes][8].
//   LayerNorm cannot be applied to split operands on the fly, so it moves into the epilogue: with the row's mean and rstd (from the
//   producer's partial sums, as before) LN(x) W'^T = rstd (x W'^T - mean csum), csum[n] = sum_k W'[n][k] precomputed.
// =====================================================================================================
namespace {

struct GemmHArgs {
    const _Float16 *Ahi, *Alo; const float *Af32; int lda, a_batch;
    const uint4 *Wp; long long w_batch;          // uint4 elements between batches
    int M, N, nbatch;
    const int *count;
    const float *bias, *csum;                    // [nbatch * N]; csum: LayerNorm mode only
    float inv_scale, a_scale;                    // 1 / (16 * 256); 16
    _Float16 *ohi, *olo; float *of32; int ldo;   // output planes (x a_scale) and / or float32
    const float *resid; int ldr;
    float ln_eps;
    const float *stats_in; float *stats_out;
    float *logits, *values; int action_dim;
    int *oflow;                                  // optional sticky flag: a plane value left fp16's range
};

// nn.GELU (erf form), erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7: 3e-7 on a hidden activation, below this path's 22-bit
// operands) - a dozen instructions; erff costs fifty, and the wide link's epilogue runs it on 32 values per lane
__device__ __forceinline__ float gelu_as(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = 1.0f / (1.0f + 0.3275911f * z);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float erf_abs = 1.0f - poly * __expf(-z * z);
    return 0.5f * x * (1.0f + copysignf(erf_abs, x));
}

template <int EPI, int LNA, int AIN, int NWK, int KW>      // AIN 1: float32 A, split on the fly
__global__ __launch_bounds__(256, 1) void k_gemm_h(GemmHArgs a) {
    constexpr int RT = 2, K = KW * NWK, S32 = K / 32, NCH = KW / 32;
    constexpr int RING = NCH >= 4 ? 4 : NCH, AHEAD = RING - 1;        // chunks of loads in flight ahead of the MFMAs
    __shared__ f32x4 kred[NWK > 1 ? (NWK - 1) * RT * 4 * 64 : 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, l4 = lane >> 4;
    // (every kernel argument is wanted in SGPRs HERE: left alone, the compiler fetches the count pointer first, waits for the count, and
    //  only then goes back to the argument segment for the rest - one more dependent scalar round trip before the first load)
    asm volatile("" :: "s"(a.Ahi), "s"(a.Alo), "s"(a.Af32), "s"(a.Wp), "s"(a.bias), "s"(a.csum), "s"(a.ohi), "s"(a.olo), "s"(a.of32), "s"(a.resid),
                 "s"(a.stats_in), "s"(a.stats_out), "s"(a.logits), "s"(a.values), "s"(a.lda), "s"(a.ldo), "s"(a.ldr), "s"(a.N), "s"(a.nbatch), "s"(a.M),
                 "s"(a.a_batch), "s"(a.w_batch), "s"(a.inv_scale), "s"(a.a_scale), "s"(a.ln_eps), "s"(a.action_dim));
    const int nvalid = a.count ? min(a.M, *a.count) : a.M;
    const int wk = NWK > 1 ? wave : 0;
    const int wr = NWK > 1 ? 0 : (wave >> 1), wc = NWK > 1 ? 0 : (wave & 1);
    constexpr int WROWS = NWK > 1 ? 16 * RT : 32 * RT, WCOLS = NWK > 1 ? 64 : 128;
    const int rtiles = (nvalid + WROWS - 1) / WROWS, ctiles = a.N / WCOLS;
    const int nitems = rtiles * ctiles * a.nbatch;
    union HF { uint4 u; f16x8 v; };
    bool oflow_seen = false;                                          // a plane value left fp16's range: reported once, at the end (a.oflow, sticky)
    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        if (NWK > 1 && item != (int)blockIdx.x) __syncthreads();
        const int ct = item % ctiles, r2 = item / ctiles, rt = r2 % rtiles, b = r2 / rtiles;
        const int row0 = rt * WROWS + wr * 16 * RT, g = ct * (WCOLS / 64) + wc;
        size_t aoff[RT];
#pragma unroll
        for (int i = 0; i < RT; i++) aoff[i] = (size_t)min(row0 + 16 * i + l15, a.M - 1) * a.lda + (size_t)b * a.a_batch + KW * wk + 8 * l4;
        const uint4 *bp = a.Wp + (size_t)b * a.w_batch + ((size_t)g * S32 + (size_t)NCH * wk) * 8 * 64 + lane;
        f32x4 acc[RT][4];
#pragma unroll
        for (int i = 0; i < RT; i++)
#pragma unroll
            for (int c = 0; c < 4; c++) acc[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        HF ah[RING][RT], al[RING][RT], bw[RING][4][2];
        f32x4 af[AIN == 1 ? RING : 1][RT][2];
        auto fetch = [&](int ch) {
            const int u = ch % RING;
#pragma unroll
            for (int i = 0; i < RT; i++) {
                if (AIN == 1) { af[u][i][0] = *(const f32x4 *)(a.Af32 + aoff[i] + 32 * ch); af[u][i][1] = *(const f32x4 *)(a.Af32 + aoff[i] + 32 * ch + 4); }
                else { ah[u][i].u = *(const uint4 *)(a.Ahi + aoff[i] + 32 * ch); al[u][i].u = *(const uint4 *)(a.Alo + aoff[i] + 32 * ch); }
            }
#pragma unroll
            for (int c = 0; c < 4; c++)
#pragma unroll
                for (int p = 0; p < 2; p++) bw[u][c][p].u = bp[((ch * 4 + c) * 2 + p) * 64];
        };
        // every operand of the epilogue is requested HERE, in front of the k loop (behind it each of them would be a memory round trip
        // of its own): bias, column sums, the residual rows, and the LayerNorm partial sums of this wave's 32 rows - lane (row lane&15,
        // quarter lane>>4) takes 16 of the row's 64 bytes, the row's mean / rstd then sit in the lanes with lane&15 = row
        const int col0 = b * a.N + 64 * g + 4 * l15;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f}, cs = {0.f, 0.f, 0.f, 0.f};
        if (a.bias) bv = *(const f32x4 *)(a.bias + col0);
        if (LNA) cs = *(const f32x4 *)(a.csum + col0);
        f32x4 st[LNA ? RT : 1];
        if (LNA) {
#pragma unroll
            for (int i = 0; i < RT; i++) st[i] = *((const f32x4 *)(a.stats_in + (size_t)min(row0 + 16 * i + l15, a.M - 1) * 16) + l4);
        }
        f32x4 rr[EPI == X_EPI_RESID ? RT : 1][4];
        if (EPI == X_EPI_RESID && (NWK == 1 || wave == 0)) {
#pragma unroll
            for (int i = 0; i < RT; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) rr[i][j] = *(const f32x4 *)(a.resid + (size_t)min(row0 + 16 * i + 4 * l4 + j, a.M - 1) * a.ldr + col0);
        }
#pragma unroll
        for (int ch = 0; ch < AHEAD && ch < NCH; ch++) fetch(ch);
#pragma unroll
        for (int ch = 0; ch < NCH; ch++) {
            const int u = ch % RING;
            if (ch + AHEAD < NCH) fetch(ch + AHEAD);
            __builtin_amdgcn_sched_barrier(0);                // (the loads AHEAD chunks ahead are issued before this chunk's MFMAs)
            if (AIN == 1) {
#pragma unroll
                for (int i = 0; i < RT; i++)
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        const float x = af[u][i][e >> 2][e & 3] * a.a_scale;
                        oflow_seen |= !(fabsf(x) < 65504.0f);
                        const _Float16 h = (_Float16)x;
                        ah[u][i].v[e] = h;
                        al[u][i].v[e] = (_Float16)(x - (float)h);
                    }
            }
            // term by term over the eight accumulators: eight independent MFMAs between two that touch the same accumulator
#pragma unroll
            for (int term = 0; term < 3; term++)
#pragma unroll
                for (int i = 0; i < RT; i++)
#pragma unroll
                    for (int c = 0; c < 4; c++)
                        acc[i][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(term == 2 ? al[u][i].v : ah[u][i].v, bw[u][c][term == 1 ? 1 : 0].v, acc[i][c], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (row0 >= nvalid) continue;
        if (NWK > 1) {
            if (wave > 0) {
#pragma unroll
                for (int i = 0; i < RT; i++)
#pragma unroll
                    for (int c = 0; c < 4; c++) kred[((wave - 1) * RT * 4 + i * 4 + c) * 64 + lane] = acc[i][c];
            }
            __syncthreads();
            if (wave > 0) continue;
#pragma unroll
            for (int w = 1; w < NWK; w++)
#pragma unroll
                for (int i = 0; i < RT; i++)
#pragma unroll
                    for (int c = 0; c < 4; c++) acc[i][c] += kred[((w - 1) * RT * 4 + i * 4 + c) * 64 + lane];
        }
        // LayerNorm statistics: the groups are added in a fixed order (deterministic); lane (lane&15 = row) ends up with the row's mean
        // and rstd, the epilogue's rows 16 i + 4 (lane>>4) + j fetch them from lane 4 (lane>>4) + j
        float rstd[RT][4], mshift[RT][4];
        if (LNA) {
#pragma unroll
            for (int i = 0; i < RT; i++) {
                float s1 = st[i][0] + st[i][2], s2 = st[i][1] + st[i][3];
                s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
                s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
                const float mean = s1 * (1.0f / 512.0f);
                const float rs = 1.0f / sqrtf(fmaxf(__builtin_fmaf(-mean, mean, s2 * (1.0f / 512.0f)), 0.f) + a.ln_eps);
#pragma unroll
                for (int j = 0; j < 4; j++) { rstd[i][j] = __shfl(rs, 4 * l4 + j); mshift[i][j] = __shfl(mean, 4 * l4 + j); }
            }
        }
        const int ngr = a.nbatch * (a.N >> 6), gr = b * (a.N >> 6) + g;
#pragma unroll
        for (int i = 0; i < RT; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int row = row0 + 16 * i + 4 * l4 + j;
                f32x4 v = {acc[i][0][j], acc[i][1][j], acc[i][2][j], acc[i][3][j]};
                v = v * a.inv_scale;
                if (LNA) v = (v - cs * mshift[i][j]) * rstd[i][j];
                v += bv;
                if (EPI == X_EPI_HEADS) {
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const int col = col0 + c;
                        if (row < nvalid && col < a.action_dim) a.logits[(size_t)row * a.action_dim + col] = v[c];
                        if (row < nvalid && col == a.action_dim) a.values[row] = tanhf(v[c]);
                    }
                    continue;
                }
                if (EPI == X_EPI_GELU) {
#pragma unroll
                    for (int c = 0; c < 4; c++) v[c] = gelu_as(v[c]);
                }
                if (EPI == X_EPI_RESID) v += rr[i][j];
                if (row < nvalid) {
                    if (a.of32) *(f32x4 *)(a.of32 + (size_t)row * a.ldo + col0) = v;
                    if (a.ohi) {
                        union { _Float16 h[4]; uint2 u; } ph, pl;
#pragma unroll
                        for (int c = 0; c < 4; c++) {
                            const float x = v[c] * a.a_scale;
                            oflow_seen |= !(fabsf(x) < 65504.0f);
                            ph.h[c] = (_Float16)x; pl.h[c] = (_Float16)(x - (float)ph.h[c]);
                        }
                        *(uint2 *)(a.ohi + (size_t)row * a.ldo + col0) = ph.u;
                        *(uint2 *)(a.olo + (size_t)row * a.ldo + col0) = pl.u;
                    }
                }
                if (a.stats_out) {
                    const f32x2 ps = {row16_sum((v[0] + v[1]) + (v[2] + v[3])), row16_sum((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]))};
                    if (l15 == 0 && row < nvalid) *(f32x2 *)(a.stats_out + ((size_t)row * ngr + gr) * 2) = ps;
                }
            }
    }
    if (a.oflow && oflow_seen) atomicOr(a.oflow, 1);
}

template <int EPI, int LNA, int AIN, int NWK, int KW>
int launch_gemm_h(const GemmHArgs &a, hipStream_t st) {
    constexpr int WROWS = NWK > 1 ? 32 : 64, WCOLS = NWK > 1 ? 64 : 128;
    const long long items = (long long)((a.M + WROWS - 1) / WROWS) * (a.N / WCOLS) * a.nbatch;
    const unsigned blocks = (unsigned)(items < 16384 ? items : 16384);
    k_gemm_h<EPI, LNA, AIN, NWK, KW><<<blocks, 256, 0, st>>>(a);
    return hipGetLastError() == hipSuccess ? AZK_OK : AZK_ERR_HIP;
}
}  // namespace

extern "C" int32_t azk_nnx_gemm_h(const azk_gemm_h *t, void *stream) {
    if (!t || !t->w_packed || t->m < 0 || t->n_out < 64 || (t->n_out & 63) || t->nbatch < 1) return AZK_ERR_ARG;
    if ((t->k != 512 && t->k != 2048 && t->k != 384) || t->lda < t->k || (t->lda & 7) || (t->a_batch_stride & 7)) return AZK_ERR_ARG;
    const bool af32 = t->a_f32 != nullptr;
    if (!af32 && (!t->a_hi || !t->a_lo)) return AZK_ERR_ARG;
    if (t->epilogue < 0 || t->epilogue > 3 || (t->layernorm_a && (t->k != 512 || !t->a_stats || !t->col_sums))) return AZK_ERR_ARG;
    if (t->epilogue == X_EPI_HEADS ? (!t->logits_out || !t->values_out || t->action_dim + 1 > t->n_out * t->nbatch)
                                   : ((!t->out_f32 && !(t->out_hi && t->out_lo)) || t->ldo < t->n_out * t->nbatch || (t->ldo & 3))) return AZK_ERR_ARG;
    if ((t->out_hi != nullptr) != (t->out_lo != nullptr)) return AZK_ERR_ARG;
    if (t->epilogue == X_EPI_RESID && (!t->resid_f32 || (t->ldr & 3))) return AZK_ERR_ARG;
    if (!(t->a_scale > 0.f) || !(t->w_scale > 0.f)) return AZK_ERR_ARG;
    if (t->m == 0) return AZK_OK;
    GemmHArgs a;
    memset(&a, 0, sizeof a);
    a.Ahi = (const _Float16 *)t->a_hi; a.Alo = (const _Float16 *)t->a_lo; a.Af32 = t->a_f32; a.lda = t->lda; a.a_batch = t->a_batch_stride;
    a.Wp = (const uint4 *)t->w_packed; a.w_batch = (long long)(t->n_out / 64) * (t->k / 32) * 8 * 64;
    a.M = t->m; a.N = t->n_out; a.nbatch = t->nbatch; a.count = t->n_valid; a.bias = t->bias; a.csum = t->col_sums;
    a.inv_scale = 1.0f / (t->a_scale * t->w_scale); a.a_scale = t->a_scale;
    a.ohi = (_Float16 *)t->out_hi; a.olo = (_Float16 *)t->out_lo; a.of32 = t->out_f32; a.ldo = t->ldo; a.resid = t->resid_f32; a.ldr = t->ldr;
    a.ln_eps = t->ln_eps; a.stats_in = t->a_stats; a.stats_out = t->stats_out; a.logits = t->logits_out; a.values = t->values_out; a.action_dim = t->action_dim;
    a.oflow = t->overflow_flag;
    hipStream_t st = (hipStream_t)stream;
    const int ln = t->layernorm_a ? 1 : 0;
    if (t->k == 2048) {
        if (ln || af32) return AZK_ERR_ARG;
        if (t->epilogue == X_EPI_RESID) return launch_gemm_h<X_EPI_RESID, 0, 0, 4, 512>(a, st);
        if (t->epilogue == X_EPI_PLAIN) return launch_gemm_h<X_EPI_PLAIN, 0, 0, 4, 512>(a, st);
        return AZK_ERR_ARG;
    }
    if (t->k == 384)                                                   // azk_nnx_embed_fold's float32 rows (AZK_EMBED_FOLD_ROW) against [D_t; U_all; M_h]
        return af32 && !ln && t->epilogue == X_EPI_PLAIN ? launch_gemm_h<X_EPI_PLAIN, 0, 1, 4, 96>(a, st) : AZK_ERR_ARG;
    const bool wide = t->n_out % 128 == 0 && t->n_out >= 1024;
    if (af32) {                                                        // the first link: float32 z, split on the fly
        if (ln || t->epilogue != X_EPI_PLAIN) return AZK_ERR_ARG;
        return launch_gemm_h<X_EPI_PLAIN, 0, 1, 4, 128>(a, st);
    }
    if (ln) {
        if (t->epilogue == X_EPI_GELU) return wide ? launch_gemm_h<X_EPI_GELU, 1, 0, 1, 512>(a, st) : launch_gemm_h<X_EPI_GELU, 1, 0, 4, 128>(a, st);
        if (t->epilogue == X_EPI_HEADS) return launch_gemm_h<X_EPI_HEADS, 1, 0, 4, 128>(a, st);
        return AZK_ERR_ARG;
    }
    if (t->epilogue == X_EPI_PLAIN) return launch_gemm_h<X_EPI_PLAIN, 0, 0, 4, 128>(a, st);
    if (t->epilogue == X_EPI_GELU) return wide ? launch_gemm_h<X_EPI_GELU, 0, 0, 1, 512>(a, st) : launch_gemm_h<X_EPI_GELU, 0, 0, 4, 128>(a, st);
    if (t->epilogue == X_EPI_RESID) return launch_gemm_h<X_EPI_RESID, 0, 0, 4, 128>(a, st);
    return launch_gemm_h<X_EPI_HEADS, 0, 0, 4, 128>(a, st);
}
