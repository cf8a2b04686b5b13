// azk_nn.hip - hand-written CDNA4 kernels for the policy-value network's token embedding (ai/nn.py:5-36):
//   tokens[n, 0, :]   = cls_token + pos_embedding[0]
//   tokens[n, 1+j, :] = Conv2d(C -> D, k x k, stride 1, 'same')(board)[:, r, c] + pos_embedding[1+j]      j = r*cols + c
// lowered to an im2col GEMM on the matrix cores (v_mfma_f32_16x16x32_bf16, fp32 accumulate).
//
// Structure (one wavefront = one 16-token x D output tile, no workgroup barriers in the main loop):
//   * the conv weight [D][KP] is staged ONCE per workgroup into LDS in MFMA-fragment order, so every B-fragment
//     read is a conflict-free, lane-linear ds_read_b128;
//   * the board is a bit string held across the wave's lanes (one ballot per 64 cells); each lane assembles the
//     k*k*C-bit patch of its token with funnel shifts and expands its 8 k-values to a bf16 A fragment - the
//     im2col matrix never exists in memory;
//   * bias + positional embedding enter as the accumulator's initial value (coalesced fp32 loads);
//   * the column -> accumulator map is permuted so each lane ends up with 8 consecutive columns per group:
//     LayerNorm statistics need only a 16-lane butterfly, and stores are 16 B per lane, 256 B contiguous.
// The kernel is bound by its HBM writes (T*D*2 bytes per board per output), not by MFMA.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>

#include "azk.h"
#include "azk_launch.h"
#include "azk_tail_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

struct EmbedArgs {
    const void *boards;        // [n][C][R][Cc] bf16 or f32, values 0/1
    int boards_f32;
    const __hip_bfloat16 *wt;  // [D][KP] conv weight, k index = ch*k*k + ky*k + kx, zero padded
    const float *cpos;         // [T][D]: row 0 = cls + pos[0]; row 1+j = conv bias + pos[1+j]
    const float *ln_w, *ln_b;  // [D] LayerNorm affine (used when xhat != nullptr)
    __hip_bfloat16 *x;         // [n][T][D] tokens (may be null)
    __hip_bfloat16 *xhat;      // [n][T][D] LayerNorm(tokens) (may be null)
    const float *mtab;         // scores variant: [T][16] per-token additive term of the 16 folded score columns (m'_h . cpos[t]), or null
    const float *msum;         // scores variant: [16] sum_d m'_h[d]
    float *scores;             // [n][NH][Tp] (Tp = 16*ceil(T/16)) xn_t . m'_h, written when mtab != null
    int nh;
    const int *count;          // optional device-side number of valid boards (<= n): rows beyond it are skipped
    int n, C, R, Cc, ksz, T;
    float eps;
    int ablate;                // debug only (AZK_EMBED_ABLATE): 1 no cpos loads, 2 no stores, 4 no MFMA, 8 no patch build
};

typedef __attribute__((ext_vector_type(8))) float f32x8;

// Sum over the 16 lanes of a DPP row (lanes sharing lane>>4), result in every lane: quad_perm [1,0,3,2], quad_perm
// [2,3,0,1], row_half_mirror, row_mirror - four v_add_f32 with a DPP operand instead of four ds_bpermute round trips.
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    return v;
}

// fp32 -> bf16 (round to nearest even) as plain vector casts: hipcc lowers them to v_cvt_pk_bf16_f32
__device__ __forceinline__ uint4 pack8(const float *v) {
    f32x8 f;
#pragma unroll
    for (int q = 0; q < 8; q++) f[q] = v[q];
    union { bf16x8 b; uint4 u; } r;
    r.b = __builtin_convertvector(f, bf16x8);
    return r.u;
}

// NG = D / 128 column groups (each lane owns 8 consecutive columns per group); KS = KP / 32 k-steps.
template <int NG, int KS, bool WANT_X, bool WANT_XHAT, int NH>
__global__ __launch_bounds__(256, 2) void k_embed(EmbedArgs a) {
    constexpr int D = 128 * NG, KP = 32 * KS, NACC = 8 * NG;
    constexpr int NTILE = NACC + (NH > 0 ? 1 : 0);     // NH > 0: one extra 16-column tile = the folded head-score columns
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4 *bimg = (uint4 *)smem;                       // [NTILE][KS][64 lanes] 16-byte B fragments
    float *lnw = (float *)(smem + NTILE * KS * 64 * 16);  // [D] LayerNorm weight, then [D] bias (affine variant only)
    float *lnb = lnw + D;
    constexpr bool AFFINE = WANT_XHAT && NH == 0;      // NH > 0 emits the plain normalised tokens (affine folded by the caller)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    // work item = (board, group of consecutive 16-token tiles): a board's tiles are spread over `groups` wavefronts so
    // the chip stays busy when only part of the batch is live
    const int tiles_per_leaf_ = (a.T + 15) >> 4;
    const int groups = tiles_per_leaf_ >= 6 ? 3 : 1, tiles_per_group = (tiles_per_leaf_ + groups - 1) / groups;
    const int nvalid = a.count ? min(a.n, *a.count) : a.n;
    const int nitems = nvalid * groups;
    if ((int)blockIdx.x * 4 >= nitems) return;                  // nothing for this workgroup: skip the weight staging too

    // ---- stage the weight in fragment order: fragment (acc, s) of lane l = wt[col(acc, l)][32 s + 8 (l>>4) .. +8] ----
    for (int f = tid; f < NTILE * KS * 64; f += 256) {
        const int l = f & 63, s = (f >> 6) % KS, acc = (f >> 6) / KS;
        const int col = acc < NACC ? 128 * (acc >> 3) + 8 * (l & 15) + (acc & 7) : D + (l & 15);   // weight rows D..D+15: score columns
        bimg[f] = *(const uint4 *)(a.wt + (size_t)col * KP + 32 * s + 8 * (l >> 4));
    }
    if (AFFINE)
        for (int i = tid; i < D; i += 256) { lnw[i] = a.ln_w[i]; lnb[i] = a.ln_b[i]; }
    __syncthreads();

    const int RC = a.R * a.Cc, T = a.T, ksz = a.ksz, kk = ksz * ksz, pad = ksz / 2, ncell = a.C * RC;
    const int tiles_per_leaf = (T + 15) >> 4;
    const int nwaves = gridDim.x * 4;


    // the board's bit string is built once per item, then the wave walks its group of 16-token tiles
    for (int item = blockIdx.x * 4 + wave; item < nitems; item += nwaves) {
        const int leaf = item / groups, grp = item - leaf * groups;
        const int tile_lo = grp * tiles_per_group;
        const int tile_hi = min(tiles_per_leaf, tile_lo + tiles_per_group);
        unsigned wbits = 0;                             // lane i holds bits [32 (i-1), 32 i) of the board bit string (lane 0: zeros)
        if (!(a.ablate & 8)) {
            for (int q = 0; q * 64 < ncell; q++) {
                const int e = q * 64 + lane;
                bool on = false;
                if (e < ncell)
                    on = a.boards_f32 ? ((const float *)a.boards)[(size_t)leaf * ncell + e] != 0.0f
                                      : (((const unsigned short *)a.boards)[(size_t)leaf * ncell + e] & 0x7fff) != 0;
                const unsigned long long m = __ballot(on);
                if ((lane - 1) >> 1 == q && lane >= 1) wbits = ((lane - 1) & 1) ? (unsigned)(m >> 32) : (unsigned)m;
            }
        }
      for (int tile = tile_lo; tile < tile_hi; tile++) {
        // ---- this lane's token (A-fragment row l&15) and its patch bits ----
        const int t = tile * 16 + l15;
        unsigned long long plo = 0, phi = 0;
        {
            const int j = t - 1, r = j / a.Cc, c = j - r * a.Cc;
            const bool live = t >= 1 && t < T;
            unsigned colmask = 0;
            for (int kx = 0; kx < ksz; kx++) { const int cc = c + kx - pad; if (cc >= 0 && cc < a.Cc) colmask |= 1u << kx; }
            for (int ch = 0; ch < ((a.ablate & 8) ? 0 : a.C); ch++)
                for (int ky = 0; ky < ksz; ky++) {
                    const int rr = r + ky - pad;
                    // every lane takes part in the shuffles; dead rows contribute zero bits
                    const int off = 32 + ch * RC + (rr < 0 ? 0 : (rr >= a.R ? a.R - 1 : rr)) * a.Cc + (c - pad);
                    const int wi = off >> 5, sh = off & 31;
                    const unsigned lo = __shfl(wbits, wi), hi = __shfl(wbits, wi + 1);
                    unsigned bits = __funnelshift_r(lo, hi, sh) & colmask;
                    if (!live || rr < 0 || rr >= a.R) bits = 0;
                    const int p0 = ch * kk + ky * ksz;
                    if (p0 < 64) { plo |= (unsigned long long)bits << p0; if (p0 + ksz > 64) phi |= (unsigned long long)bits >> (64 - p0); }
                    else phi |= (unsigned long long)bits << (p0 - 64);
                }
        }
        bf16x8 afrag[KS];
#pragma unroll
        for (int s = 0; s < KS; s++) {
            const int b0 = 32 * s + 8 * l4;
            const unsigned byte = (unsigned)((b0 < 64 ? (plo >> b0) : (phi >> (b0 - 64))) & 0xff);
            union { bf16x8 v; unsigned short h[8]; } u;
#pragma unroll
            for (int q = 0; q < 8; q++) u.h[q] = ((byte >> q) & 1) ? 0x3F80 : 0;
            afrag[s] = u.v;
        }
        // ---- accumulators start at bias + positional embedding (C/D map: col = lane&15 -> permuted column,
        //      row = 4 (lane>>4) + reg) ----
        f32x4 acc[NACC];
        int trow[4];
#pragma unroll
        for (int r4 = 0; r4 < 4; r4++) { const int tt = tile * 16 + 4 * l4 + r4; trow[r4] = tt < T ? tt : T - 1; }
#pragma unroll
        for (int g = 0; g < NG; g++)
#pragma unroll
            for (int r4 = 0; r4 < 4; r4++) {
                const float *src = a.cpos + ((a.ablate & 1) ? 0 : (size_t)trow[r4] * D) + 128 * g + 8 * l15;
                const f32x4 c0 = *(const f32x4 *)src, c1 = *(const f32x4 *)(src + 4);
                acc[g * 8 + 0][r4] = c0[0]; acc[g * 8 + 1][r4] = c0[1]; acc[g * 8 + 2][r4] = c0[2]; acc[g * 8 + 3][r4] = c0[3];
                acc[g * 8 + 4][r4] = c1[0]; acc[g * 8 + 5][r4] = c1[1]; acc[g * 8 + 6][r4] = c1[2]; acc[g * 8 + 7][r4] = c1[3];
            }
        f32x4 acce;                                                   // NH > 0: raw scores x_t . m'_h for head = lane&15, tokens 4 (lane>>4) + r
        if (NH > 0) {
#pragma unroll
            for (int r4 = 0; r4 < 4; r4++) acce[r4] = a.mtab[(size_t)trow[r4] * 16 + l15];
#pragma unroll
            for (int s = 0; s < KS; s++) {
                union { uint4 u; bf16x8 v; } bf;
                bf.u = bimg[(NACC * KS + s) * 64 + lane];
                acce = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag[s], bf.v, acce, 0, 0, 0);
            }
        }
        // ---- MFMA: acc[n] (16 x 16) += A (16 x KP) * B (KP x 16) ----
        if (!(a.ablate & 4))
#pragma unroll
        for (int n = 0; n < NACC; n++) {
#pragma unroll
            for (int s = 0; s < KS; s++) {
                union { uint4 u; bf16x8 v; } bf;
                bf.u = bimg[(n * KS + s) * 64 + lane];
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag[s], bf.v, acc[n], 0, 0, 0);
            }
            if ((n & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // keep B-fragment prefetch to 4 accumulators (VGPR budget)
        }
        // ---- epilogue: rows 4 (lane>>4) + r4, this lane's columns 128 g + 8 (lane&15) + q ----
        float mean[4], rstd[4];
        if (WANT_XHAT) {
#pragma unroll
            for (int r4 = 0; r4 < 4; r4++) {
                float s = 0.f;
#pragma unroll
                for (int n = 0; n < NACC; n++) s += acc[n][r4];
                s = row16_sum(s);
                mean[r4] = s * (1.0f / (float)D);
                float ss = 0.f;
#pragma unroll
                for (int n = 0; n < NACC; n++) { const float dl = acc[n][r4] - mean[r4]; ss += dl * dl; }
                ss = row16_sum(ss);
                rstd[r4] = rsqrtf(ss * (1.0f / (float)D) + a.eps);
            }
        }
#pragma unroll
        for (int g = 0; g < NG; g++) {
            f32x4 w0, w1, b0, b1;
            if (AFFINE) {
                w0 = *(const f32x4 *)(lnw + 128 * g + 8 * l15); w1 = *(const f32x4 *)(lnw + 128 * g + 8 * l15 + 4);
                b0 = *(const f32x4 *)(lnb + 128 * g + 8 * l15); b1 = *(const f32x4 *)(lnb + 128 * g + 8 * l15 + 4);
            }
#pragma unroll
            for (int r4 = 0; r4 < 4; r4++) {
                const int tt = tile * 16 + 4 * l4 + r4;
                const bool ok = tt < T && !((a.ablate & 2) && tt != 7777);
                const size_t orow = ((size_t)leaf * T + (tt < T ? tt : 0)) * D;
                float v[8];
#pragma unroll
                for (int q = 0; q < 8; q++) v[q] = acc[g * 8 + q][r4];
                if (WANT_X && ok) *(uint4 *)(a.x + orow + 128 * g + 8 * l15) = pack8(v);
                if (WANT_XHAT) {
                    if (AFFINE) {
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            v[q] = (v[q] - mean[r4]) * rstd[r4] * w0[q] + b0[q];
                            v[q + 4] = (v[q + 4] - mean[r4]) * rstd[r4] * w1[q] + b1[q];
                        }
                    } else {
#pragma unroll
                        for (int q = 0; q < 8; q++) v[q] = (v[q] - mean[r4]) * rstd[r4];
                    }
                    if (ok) *(uint4 *)(a.xhat + orow + 128 * g + 8 * l15) = pack8(v);
                }
            }
        }
        if (NH > 0 && l15 < NH) {
            // xn = (x - mean) * rstd  =>  xn . m' = rstd * (x . m' - mean * sum(m'))
            const int Tp = tiles_per_leaf * 16;
            const float ms = a.msum[l15];
#pragma unroll
            for (int r4 = 0; r4 < 4; r4++)
                a.scores[((size_t)leaf * NH + l15) * Tp + tile * 16 + 4 * l4 + r4] = rstd[r4] * (acce[r4] - mean[r4] * ms);
        }
      }
    }
}

template <int NG, int KS, bool WX, bool WH, int NH>
int launch_embed2(const EmbedArgs &a, hipStream_t st) {
    constexpr int NACC = 8 * NG;
    const int lds = (NACC + (NH > 0 ? 1 : 0)) * KS * 64 * 16 + (NH > 0 ? 0 : 2 * 128 * NG * 4);
    const int tiles = (a.T + 15) >> 4, groups = tiles >= 6 ? 3 : 1;
    long long blocks = ((long long)a.n * groups + 3) / 4;       // one wavefront per (board, tile group); idle workgroups exit at once
    if (blocks > 4096) blocks = 4096;
    if (azk_set_max_lds((const void *)k_embed<NG, KS, WX, WH, NH>, lds) != hipSuccess) return AZK_ERR_HIP;
    k_embed<NG, KS, WX, WH, NH><<<(unsigned)blocks, 256, lds, st>>>(a);
    return hipGetLastError() == hipSuccess ? AZK_OK : AZK_ERR_HIP;
}

template <int NG, int KS>
int launch_embed(const EmbedArgs &a, hipStream_t st) {
    if (a.mtab) {                                     // scores ride along with xhat (folded cls attention, shared query)
        if (!a.xhat || a.x) return AZK_ERR_ARG;
        if (a.nh == 8) return launch_embed2<NG, KS, false, true, 8>(a, st);
        if (a.nh == 4) return launch_embed2<NG, KS, false, true, 4>(a, st);
        return AZK_ERR_ARG;
    }
    if (a.x && a.xhat) return launch_embed2<NG, KS, true, true, 0>(a, st);
    if (a.xhat) return launch_embed2<NG, KS, false, true, 0>(a, st);
    return launch_embed2<NG, KS, true, false, 0>(a, st);
}

}  // namespace

static int32_t patch_embed_impl(const void *boards_dev, int32_t boards_are_f32, const void *wt_bf16_dev,
                                const float *cpos_dev, const float *ln_w_dev, const float *ln_b_dev,
                                void *x_out_bf16_dev, void *xhat_out_bf16_dev, int32_t n, int32_t channels,
                                int32_t rows, int32_t cols, int32_t ksize, int32_t kp, int32_t embed_dim,
                                float ln_eps, const float *mtab_dev, const float *msum_dev, float *scores_dev, int32_t num_heads,
                                const int32_t *n_valid_dev, void *stream) {
    if (!boards_dev || !wt_bf16_dev || !cpos_dev || (!x_out_bf16_dev && !xhat_out_bf16_dev)) return AZK_ERR_ARG;
    if (xhat_out_bf16_dev && !mtab_dev && (!ln_w_dev || !ln_b_dev)) return AZK_ERR_ARG;   // the scores variant has no affine
    if (n < 0 || channels < 1 || rows < 1 || cols < 1 || ksize < 1 || (ksize & 1) == 0 || ksize > 7) return AZK_ERR_ARG;
    if (kp < channels * ksize * ksize || kp % 32 != 0 || kp > 128) return AZK_ERR_ARG;
    if (channels * rows * cols > 62 * 32) return AZK_ERR_ARG;         // the board bit string lives in one wave's lanes
    if (n == 0) return AZK_OK;
    EmbedArgs a;
    a.boards = boards_dev; a.boards_f32 = boards_are_f32; a.wt = (const __hip_bfloat16 *)wt_bf16_dev; a.cpos = cpos_dev;
    a.ln_w = ln_w_dev; a.ln_b = ln_b_dev; a.x = (__hip_bfloat16 *)x_out_bf16_dev; a.xhat = (__hip_bfloat16 *)xhat_out_bf16_dev;
    a.mtab = mtab_dev; a.msum = msum_dev; a.scores = scores_dev; a.nh = num_heads; a.count = n_valid_dev;
    if ((mtab_dev != nullptr) != (scores_dev != nullptr)) return AZK_ERR_ARG;
    { const char *ab = getenv("AZK_EMBED_ABLATE"); a.ablate = ab ? atoi(ab) : 0; }
    a.n = n; a.C = channels; a.R = rows; a.Cc = cols; a.ksz = ksize; a.T = rows * cols + 1; a.eps = ln_eps;
    hipStream_t st = (hipStream_t)stream;
    const int ks = kp / 32;
#define CASE(NG_, KS_) if (embed_dim == 128 * NG_ && ks == KS_) return launch_embed<NG_, KS_>(a, st)
    CASE(4, 2); CASE(4, 1); CASE(4, 3);
    CASE(2, 2); CASE(2, 1); CASE(2, 3);
    CASE(1, 2); CASE(1, 1); CASE(1, 3);
#undef CASE
    return AZK_ERR_ARG;   // unsupported (embed_dim, kp): the caller keeps its generic path
}

extern "C" int32_t azk_nn_patch_embed(const void *boards_dev, int32_t boards_are_f32, const void *wt_bf16_dev,
                                      const float *cpos_dev, const float *ln_w_dev, const float *ln_b_dev,
                                      void *x_out_bf16_dev, void *xhat_out_bf16_dev, int32_t n, int32_t channels,
                                      int32_t rows, int32_t cols, int32_t ksize, int32_t kp, int32_t embed_dim,
                                      float ln_eps, void *stream) {
    return patch_embed_impl(boards_dev, boards_are_f32, wt_bf16_dev, cpos_dev, ln_w_dev, ln_b_dev, x_out_bf16_dev,
                            xhat_out_bf16_dev, n, channels, rows, cols, ksize, kp, embed_dim, ln_eps, nullptr, nullptr, nullptr, 0, nullptr, stream);
}

extern "C" int32_t azk_nn_patch_embed_scores(const void *boards_dev, int32_t boards_are_f32, const void *wt_bf16_dev,
                                             const float *cpos_dev, const float *ln_w_dev, const float *ln_b_dev,
                                             void *xhat_out_bf16_dev, const float *score_cpos_dev, const float *score_msum_dev,
                                             float *scores_out_dev, int32_t num_heads, int32_t n, int32_t channels, int32_t rows, int32_t cols,
                                             int32_t ksize, int32_t kp, int32_t embed_dim, float ln_eps,
                                             const int32_t *n_valid_dev, void *stream) {
    if (!score_cpos_dev || !score_msum_dev || !scores_out_dev) return AZK_ERR_ARG;
    return patch_embed_impl(boards_dev, boards_are_f32, wt_bf16_dev, cpos_dev, ln_w_dev, ln_b_dev, nullptr,
                            xhat_out_bf16_dev, n, channels, rows, cols, ksize, kp, embed_dim, ln_eps, score_cpos_dev, score_msum_dev,
                            scores_out_dev, num_heads, n_valid_dev, stream);
}

// =====================================================================================================
// cls-row attention of the LAST block, folded (ai/nn.py:52-56 restricted to the row nn.py:80 reads).
// With q = Wq LN1(x)[cls] + bq fixed per board, the scores against every token are
//     s[h][t] = scale * q_h . (Wk_h xhat_t + bk_h) = xhat_t . m_h + c_h,   m_h = scale * Wk_h^T q_h,  c_h = scale * q_h . bk_h
// and the head outputs are Wv_h (sum_t softmax_t(s[h])[t] xhat_t) + bv_h, so K and V are never formed:
// this kernel streams xhat once (online softmax, flash-style running max / sum per wave) and emits
//     z[b][h][:] = sum_t softmax_t(s[b][h][:])[t] * xhat[b][t][:]            ([n][H][D])
// The tiny per-board GEMMs around it (m_h, Wv_h z_h, out-proj, MLP, heads) stay in the caller.
// One workgroup (4 waves) per board; wave w takes tokens t = w (mod 4); lane l owns CPL = D/64 columns.
// HBM-bound: reads T*D*2 bytes per board once.
// =====================================================================================================
namespace {

struct ClsAttnArgs {
    const __hip_bfloat16 *xhat;   // [n][T][D]
    const float *m;               // [n or 1][H][D]  (already multiplied by the softmax scale)
    const float *c;               // [n or 1][H]
    long long m_stride, c_stride; // elements between boards (0 = shared by every board)
    __hip_bfloat16 *z;            // [n][H][D]
    int n, T;
};

template <int CPL, int NH>
__global__ __launch_bounds__(256) void k_cls_attn(ClsAttnArgs a) {
    constexpr int D = 64 * CPL;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *zpart = (float *)smem;                    // [4 waves][NH][D]
    float *mlpart = (float *)(smem + 4 * NH * D * 4); // [4][NH] running max, then [4][NH] running sum
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    const float *mp = a.m + (size_t)b * a.m_stride, *cp = a.c + (size_t)b * a.c_stride;
    float mh[NH][CPL], ch[NH];
#pragma unroll
    for (int h = 0; h < NH; h++) {
        ch[h] = cp[h];
#pragma unroll
        for (int q = 0; q < CPL; q++) mh[h][q] = mp[h * D + lane * CPL + q];
    }
    float run_m[NH], run_l[NH], zacc[NH][CPL];
#pragma unroll
    for (int h = 0; h < NH; h++) {
        run_m[h] = -3.0e38f; run_l[h] = 0.f;
#pragma unroll
        for (int q = 0; q < CPL; q++) zacc[h][q] = 0.f;
    }
    const unsigned short *base = (const unsigned short *)a.xhat + (size_t)b * a.T * D + lane * CPL;
    for (int t = wave; t < a.T; t += 4) {
        float xv[CPL];
        if (CPL == 8) {
            const uint4 raw = *(const uint4 *)(base + (size_t)t * D);
            xv[0] = __uint_as_float(raw.x << 16); xv[1] = __uint_as_float(raw.x & 0xffff0000u);
            xv[2] = __uint_as_float(raw.y << 16); xv[3] = __uint_as_float(raw.y & 0xffff0000u);
            xv[4] = __uint_as_float(raw.z << 16); xv[5] = __uint_as_float(raw.z & 0xffff0000u);
            xv[6] = __uint_as_float(raw.w << 16); xv[7] = __uint_as_float(raw.w & 0xffff0000u);
        } else {
#pragma unroll
            for (int q = 0; q < CPL; q++) xv[q] = __uint_as_float((unsigned)base[(size_t)t * D + q] << 16);
        }
        float s[NH];
#pragma unroll
        for (int h = 0; h < NH; h++) {
            float p = 0.f;
#pragma unroll
            for (int q = 0; q < CPL; q++) p += xv[q] * mh[h][q];
            s[h] = p;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
#pragma unroll
            for (int h = 0; h < NH; h++) s[h] += __shfl_xor(s[h], off);
#pragma unroll
        for (int h = 0; h < NH; h++) {
            const float sc = s[h] + ch[h];
            const float nm = fmaxf(run_m[h], sc);
            const float alpha = __expf(run_m[h] - nm), p = __expf(sc - nm);
            run_m[h] = nm;
            run_l[h] = run_l[h] * alpha + p;
#pragma unroll
            for (int q = 0; q < CPL; q++) zacc[h][q] = zacc[h][q] * alpha + p * xv[q];
        }
    }
    // ---- combine the four waves' partial (max, sum, z) ----
#pragma unroll
    for (int h = 0; h < NH; h++) {
        if (lane == 0) { mlpart[wave * NH + h] = run_m[h]; mlpart[4 * NH + wave * NH + h] = run_l[h]; }
#pragma unroll
        for (int q = 0; q < CPL; q++) zpart[(wave * NH + h) * D + lane * CPL + q] = zacc[h][q];
    }
    __syncthreads();
    for (int i = tid; i < NH * D; i += 256) {
        const int h = i / D, col = i - h * D;
        float M = mlpart[h];
#pragma unroll
        for (int w = 1; w < 4; w++) M = fmaxf(M, mlpart[w * NH + h]);
        float L = 0.f, Z = 0.f;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const float e = __expf(mlpart[w * NH + h] - M);
            L += mlpart[4 * NH + w * NH + h] * e;
            Z += zpart[(w * NH + h) * D + col] * e;
        }
        a.z[((size_t)b * NH + h) * D + col] = __float2bfloat16(Z / L);
    }
}

template <int CPL, int NH>
int launch_cls_attn(const ClsAttnArgs &a, hipStream_t st) {
    constexpr int D = 64 * CPL;
    const int lds = 4 * NH * D * 4 + 8 * NH * 4;
    if (azk_set_max_lds((const void *)k_cls_attn<CPL, NH>, lds) != hipSuccess) return AZK_ERR_HIP;
    k_cls_attn<CPL, NH><<<a.n, 256, lds, st>>>(a);
    return hipGetLastError() == hipSuccess ? AZK_OK : AZK_ERR_HIP;
}

}  // namespace

extern "C" int32_t azk_nn_cls_attention(const void *xhat_bf16_dev, const float *m_dev, const float *c_dev,
                                        int32_t per_board_m, void *z_out_bf16_dev, int32_t n, int32_t tokens,
                                        int32_t embed_dim, int32_t num_heads, void *stream) {
    if (!xhat_bf16_dev || !m_dev || !c_dev || !z_out_bf16_dev || n < 0 || tokens < 1) return AZK_ERR_ARG;
    if (n == 0) return AZK_OK;
    ClsAttnArgs a;
    a.xhat = (const __hip_bfloat16 *)xhat_bf16_dev; a.m = m_dev; a.c = c_dev; a.z = (__hip_bfloat16 *)z_out_bf16_dev;
    a.m_stride = per_board_m ? (long long)num_heads * embed_dim : 0; a.c_stride = per_board_m ? num_heads : 0;
    a.n = n; a.T = tokens;
    hipStream_t st = (hipStream_t)stream;
#define CASE(CPL_, NH_) if (embed_dim == 64 * CPL_ && num_heads == NH_) return launch_cls_attn<CPL_, NH_>(a, st)
    CASE(8, 8); CASE(4, 8); CASE(4, 4); CASE(2, 4); CASE(2, 8); CASE(8, 4);
#undef CASE
    return AZK_ERR_ARG;
}

// =====================================================================================================
// k_cls_pool: the streaming half of the folded cls attention when the scores already exist (emitted by k_embed for the
// depth-1 case, where the cls query is a constant of the weights):  a = softmax_t(scores[b][h][:] + c[h]),
// z[b][h][:] = sum_t a[h][t] * xhat[b][t][:].  No cross-lane reduction in the token loop: each lane owns 8 (CPL)
// columns, reads its 16 bytes of every token row and the token's NH weights (one broadcast LDS read).
// One workgroup per board, 4 waves interleave tokens, 4 tokens in flight per wave.  HBM-read-bound.
// =====================================================================================================
namespace {

struct ClsPoolArgs {
    const __hip_bfloat16 *xhat;   // [n][T][D]
    const float *scores;          // [n][NH][Tp]
    const float *c;               // [NH]
    __hip_bfloat16 *z;            // [n][NH][D]
    int n, T, Tp;
    const int *count;             // optional device-side number of valid boards
    int ablate;                   // debug only (AZK_POOL_ABLATE): 1 no softmax phase, 2 no token loop, 4 no combine
};

template <int CPL>
__device__ __forceinline__ void load_row(const unsigned short *p, float *xv) {
    if (CPL == 8) {
        const uint4 raw = *(const uint4 *)p;
        xv[0] = __uint_as_float(raw.x << 16); xv[1] = __uint_as_float(raw.x & 0xffff0000u);
        xv[2] = __uint_as_float(raw.y << 16); xv[3] = __uint_as_float(raw.y & 0xffff0000u);
        xv[4] = __uint_as_float(raw.z << 16); xv[5] = __uint_as_float(raw.z & 0xffff0000u);
        xv[6] = __uint_as_float(raw.w << 16); xv[7] = __uint_as_float(raw.w & 0xffff0000u);
    } else if (CPL == 4) {
        const uint2 raw = *(const uint2 *)p;
        xv[0] = __uint_as_float(raw.x << 16); xv[1] = __uint_as_float(raw.x & 0xffff0000u);
        xv[2] = __uint_as_float(raw.y << 16); xv[3] = __uint_as_float(raw.y & 0xffff0000u);
    } else {
        const unsigned raw = *(const unsigned *)p;
        xv[0] = __uint_as_float(raw << 16); xv[1] = __uint_as_float(raw & 0xffff0000u);
    }
}

template <int CPL, int NH>
__global__ __launch_bounds__(256) void k_cls_pool(ClsPoolArgs a) {
    constexpr int D = 64 * CPL;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;                                // one workgroup (4 waves) per board: wave w takes tokens 8 (4 i + w) .. +8
    if (b >= (a.count ? min(a.n, *a.count) : a.n)) return;
    float *aw = (float *)smem;                               // [Tp][NH] softmax weights
    float *zpart = aw + (size_t)a.Tp * NH;                   // [2][NH][D] partial sums handed between waves
    const float *sp = a.scores + (size_t)b * NH * a.Tp;
    // ---- softmax over tokens: wave w handles heads w, w + 4 (scores are tiny: NH * T floats) ----
    if (!(a.ablate & 1)) {
        for (int h = wave; h < NH; h += 4) {
            const float ch = a.c[h];
            float e[4], mx = -3.0e38f;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int t = lane + 64 * k;
                e[k] = t < a.T ? sp[h * a.Tp + t] + ch : -3.0e38f;
                mx = fmaxf(mx, e[k]);
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < 4; k++) { e[k] = (lane + 64 * k) < a.T ? __expf(e[k] - mx) : 0.f; sum += e[k]; }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
            const float inv = 1.0f / sum;
#pragma unroll
            for (int k = 0; k < 4; k++) { const int t = lane + 64 * k; if (t < a.Tp) aw[t * NH + h] = e[k] * inv; }
        }
    }
    __syncthreads();
    // ---- weighted token sum: each lane owns CPL columns; 8-row register sets, two in flight per wave ----
    float zacc[NH][CPL];
#pragma unroll
    for (int h = 0; h < NH; h++)
#pragma unroll
        for (int q = 0; q < CPL; q++) zacc[h][q] = 0.f;
    const unsigned short *base = (const unsigned short *)a.xhat + (size_t)b * a.T * D + lane * CPL;
    const int T = (a.ablate & 2) ? 0 : a.T;
    // raw rows stay in registers exactly as loaded (no conversion at fetch time, so nothing waits on a load until its row
    // is consumed and a whole 8-row set stays in flight behind the one being used)
    typedef typename std::conditional<CPL == 8, uint4, typename std::conditional<CPL == 4, uint2, unsigned>::type>::type raw_t;
    auto consume = [&](const raw_t (&x)[8], int t0) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            float xv[CPL];
            if constexpr (CPL == 8) {
                xv[0] = __uint_as_float(x[k].x << 16); xv[1] = __uint_as_float(x[k].x & 0xffff0000u);
                xv[2] = __uint_as_float(x[k].y << 16); xv[3] = __uint_as_float(x[k].y & 0xffff0000u);
                xv[4] = __uint_as_float(x[k].z << 16); xv[5] = __uint_as_float(x[k].z & 0xffff0000u);
                xv[6] = __uint_as_float(x[k].w << 16); xv[7] = __uint_as_float(x[k].w & 0xffff0000u);
            } else if constexpr (CPL == 4) {
                xv[0] = __uint_as_float(x[k].x << 16); xv[1] = __uint_as_float(x[k].x & 0xffff0000u);
                xv[2] = __uint_as_float(x[k].y << 16); xv[3] = __uint_as_float(x[k].y & 0xffff0000u);
            } else {
                xv[0] = __uint_as_float(x[k] << 16); xv[1] = __uint_as_float(x[k] & 0xffff0000u);
            }
            float w[NH];
#pragma unroll
            for (int h = 0; h < NH; h += 4) {
                const f32x4 w4 = *(const f32x4 *)(aw + (t0 + k) * NH + h);
                w[h] = w4[0]; w[h + 1] = w4[1]; w[h + 2] = w4[2]; w[h + 3] = w4[3];
            }
#pragma unroll
            for (int h = 0; h < NH; h++)
#pragma unroll
                for (int q = 0; q < CPL; q++) zacc[h][q] += w[h] * xv[q];
        }
    };
    // rows past T are clamped to the last row; their weights aw[t >= T] are exactly 0 (Tp padding), so they add nothing
    auto fetch = [&](raw_t (&x)[8], int t0) {
#pragma unroll
        for (int k = 0; k < 8; k++) { const int t = t0 + k < a.T ? t0 + k : a.T - 1; x[k] = *(const raw_t *)(base + (size_t)t * D); }
    };
    raw_t xa[8], xb[8];
    const int t_first = 8 * wave;                             // this wave's 8-token chunks: t_first, t_first + 32, ...
    if (t_first < T) fetch(xa, t_first);
    for (int t = t_first; t < T; t += 64) {
        if (t + 32 < T) fetch(xb, t + 32);
        consume(xa, t);
        if (t + 64 < T) fetch(xa, t + 64);
        if (t + 32 < T) consume(xb, t + 32);
    }
    // ---- combine the four waves: 3,2 -> LDS ; 1,0 add theirs and 1 -> LDS ; 0 adds, packs, stores ----
    if (!(a.ablate & 4)) {
        if (wave >= 2) {
#pragma unroll
            for (int h = 0; h < NH; h++)
#pragma unroll
                for (int q = 0; q < CPL; q++) zpart[((wave - 2) * NH + h) * D + q * 64 + lane] = zacc[h][q];
        }
        __syncthreads();
        if (wave < 2) {
#pragma unroll
            for (int h = 0; h < NH; h++)
#pragma unroll
                for (int q = 0; q < CPL; q++) zacc[h][q] += zpart[(wave * NH + h) * D + q * 64 + lane];
        }
        __syncthreads();
        if (wave == 1) {
#pragma unroll
            for (int h = 0; h < NH; h++)
#pragma unroll
                for (int q = 0; q < CPL; q++) zpart[h * D + q * 64 + lane] = zacc[h][q];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int h = 0; h < NH; h++) {
#pragma unroll
                for (int q = 0; q < CPL; q++) zacc[h][q] += zpart[h * D + q * 64 + lane];
                unsigned short *dst = (unsigned short *)a.z + ((size_t)b * NH + h) * D + lane * CPL;
                if (CPL == 8) {
                    *(uint4 *)dst = pack8(zacc[h]);
                } else {
#pragma unroll
                    for (int q = 0; q < CPL; q++) dst[q] = __bfloat16_as_ushort(__float2bfloat16(zacc[h][q]));
                }
            }
        }
    }
}

template <int CPL, int NH>
int launch_cls_pool(const ClsPoolArgs &a, hipStream_t st) {
    const int lds = (a.Tp * NH + 2 * NH * 64 * CPL) * 4;
    if (azk_set_max_lds((const void *)k_cls_pool<CPL, NH>, 64 * 1024) != hipSuccess) return AZK_ERR_HIP;
    if (lds > 64 * 1024) return AZK_ERR_ARG;
    k_cls_pool<CPL, NH><<<a.n, 256, lds, st>>>(a);
    return hipGetLastError() == hipSuccess ? AZK_OK : AZK_ERR_HIP;
}

}  // namespace

extern "C" int32_t azk_nn_cls_pool(const void *xhat_bf16_dev, const float *scores_dev, const float *c_dev,
                                   void *z_out_bf16_dev, int32_t n, int32_t tokens, int32_t embed_dim,
                                   int32_t num_heads, const int32_t *n_valid_dev, void *stream) {
    if (!xhat_bf16_dev || !scores_dev || !c_dev || !z_out_bf16_dev || n < 0 || tokens < 1) return AZK_ERR_ARG;
    if (n == 0) return AZK_OK;
    ClsPoolArgs a;
    a.xhat = (const __hip_bfloat16 *)xhat_bf16_dev; a.scores = scores_dev; a.c = c_dev; a.z = (__hip_bfloat16 *)z_out_bf16_dev;
    a.n = n; a.T = tokens; a.Tp = (tokens + 15) / 16 * 16; a.count = n_valid_dev;
    { const char *ab = getenv("AZK_POOL_ABLATE"); a.ablate = ab ? atoi(ab) : 0; }
    hipStream_t st = (hipStream_t)stream;
#define CASE(CPL_, NH_) if (embed_dim == 64 * CPL_ && num_heads == NH_) return launch_cls_pool<CPL_, NH_>(a, st)
    CASE(8, 8); CASE(4, 8); CASE(4, 4); CASE(2, 4); CASE(2, 8); CASE(8, 4);
#undef CASE
    return AZK_ERR_ARG;
}


// =====================================================================================================
// k_heads_finalize: the merged policy/value head GEMM output [n][ld] (bf16; columns [0, A) logits, column A the raw
// value) -> logits float32 [n][A] and values float32 [n] = tanh(raw) (nn.py:82-83), one launch instead of three.
// =====================================================================================================
namespace {
__global__ void k_heads_finalize(const unsigned short *__restrict__ out, int ld, int A, int n, float *__restrict__ logits,
                                 float *__restrict__ values, const int *count) {
    const int nvalid = count ? min(n, *count) : n;
    const long long total = (long long)nvalid * (A + 1);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(i / (A + 1)), col = (int)(i - (long long)row * (A + 1));
        const float v = __uint_as_float((unsigned)out[(size_t)row * ld + col] << 16);
        if (col < A) logits[(size_t)row * A + col] = v;
        else values[row] = tanhf(v);
    }
}
}  // namespace

extern "C" int32_t azk_nn_heads_finalize(const void *heads_bf16_dev, int32_t ld, int32_t action_dim, int32_t n,
                                         float *logits_out_dev, float *values_out_dev, const int32_t *n_valid_dev,
                                         void *stream) {
    if (!heads_bf16_dev || !logits_out_dev || !values_out_dev || ld < action_dim + 1 || n < 0) return AZK_ERR_ARG;
    if (n == 0) return AZK_OK;
    k_heads_finalize<<<1024, 256, 0, (hipStream_t)stream>>>((const unsigned short *)heads_bf16_dev, ld, action_dim, n, logits_out_dev,
                                                           values_out_dev, n_valid_dev);
    return hipGetLastError() == hipSuccess ? AZK_OK : AZK_ERR_HIP;
}


// =====================================================================================================
// k_ln_rows: LayerNorm over the rows of a bf16 matrix [n][D] (nn.LayerNorm: biased variance, eps inside the sqrt,
// fp32 statistics), one wave per row, 16-byte loads/stores; optionally also writes x + add_bias back in place (the
// residual operand of the following GEMM, nn.py:59-60: the mlp.3 bias joins the residual before the product is added).
// =====================================================================================================
namespace {
__device__ __forceinline__ float wave64_sum(float v) {
    v = row16_sum(v);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

template <int VPL>   // values per lane: D = 64 * VPL
__global__ __launch_bounds__(256) void k_ln_rows(unsigned short *__restrict__ x, const float *__restrict__ w, const float *__restrict__ b,
                                                 float eps, unsigned short *__restrict__ y, const float *__restrict__ add_bias, int n,
                                                 const int *count) {
    constexpr int D = 64 * VPL;
    const int nvalid = count ? min(n, *count) : n;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= nvalid) return;
    unsigned short *xr = x + (size_t)row * D + lane * VPL;
    float v[VPL];
    if constexpr (VPL == 8) {
        const uint4 raw = *(const uint4 *)xr;
        const unsigned u[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
        for (int q = 0; q < 4; q++) { v[2 * q] = __uint_as_float(u[q] << 16); v[2 * q + 1] = __uint_as_float(u[q] & 0xffff0000u); }
    } else {
#pragma unroll
        for (int q = 0; q < VPL; q++) v[q] = __uint_as_float((unsigned)xr[q] << 16);
    }
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < VPL; q++) s += v[q];
    const float mean = wave64_sum(s) * (1.0f / D);
    float ss = 0.f;
#pragma unroll
    for (int q = 0; q < VPL; q++) { const float d = v[q] - mean; ss += d * d; }
    const float rstd = rsqrtf(wave64_sum(ss) * (1.0f / D) + eps);
    float o[VPL], r[VPL];
#pragma unroll
    for (int q = 0; q < VPL; q++) {
        o[q] = (v[q] - mean) * rstd * w[lane * VPL + q] + b[lane * VPL + q];
        if (add_bias) r[q] = v[q] + add_bias[lane * VPL + q];
    }
    unsigned short *yr = y + (size_t)row * D + lane * VPL;
    if constexpr (VPL == 8) {
        *(uint4 *)yr = pack8(o);
        if (add_bias) *(uint4 *)xr = pack8(r);
    } else {
#pragma unroll
        for (int q = 0; q < VPL; q++) {
            yr[q] = __builtin_bit_cast(unsigned short, (__bf16)o[q]);
            if (add_bias) xr[q] = __builtin_bit_cast(unsigned short, (__bf16)r[q]);
        }
    }
}
}  // namespace

extern "C" int32_t azk_nn_layernorm_rows(void *x_bf16_dev, const float *w_dev, const float *b_dev, float eps, void *y_bf16_dev,
                                         const float *add_bias_dev, int32_t n, int32_t embed_dim, const int32_t *n_valid_dev,
                                         void *stream) {
    if (!x_bf16_dev || !w_dev || !b_dev || !y_bf16_dev || n < 0) return AZK_ERR_ARG;
    if (embed_dim != 128 && embed_dim != 256 && embed_dim != 512) return AZK_ERR_ARG;
    if (n == 0) return AZK_OK;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((n + 3) / 4), block(256);
    unsigned short *x = (unsigned short *)x_bf16_dev, *y = (unsigned short *)y_bf16_dev;
    if (embed_dim == 512) k_ln_rows<8><<<grid, block, 0, st>>>(x, w_dev, b_dev, eps, y, add_bias_dev, n, n_valid_dev);
    else if (embed_dim == 256) k_ln_rows<4><<<grid, block, 0, st>>>(x, w_dev, b_dev, eps, y, add_bias_dev, n, n_valid_dev);
    else k_ln_rows<2><<<grid, block, 0, st>>>(x, w_dev, b_dev, eps, y, add_bias_dev, n, n_valid_dev);
    return hipGetLastError() == hipSuccess ? AZK_OK : AZK_ERR_HIP;
}


// =====================================================================================================
// k_embed_pool: patch embedding + LayerNorm1 + folded cls attention (scores, softmax over tokens, weighted token sum)
// in ONE kernel - the normalised tokens never leave the CU (nn.py:13-36, 52-56 restricted to the cls query).
//   One workgroup (4 waves) per board; wave w owns the 128 output columns [128 w, 128 w + 128) of every 16-token tile
//   (8 MFMA accumulators) and recomputes the 16 extra columns (heads' raw scores, and column 15 = the row mean).  Per tile:
//     x tile (MFMA 16x16x32; accumulators start at bias + positional embedding; A fragments = a 256-entry LDS table
//     indexed by 8 patch bits)
//     -> per-row sum of squares over the wave's columns -> LDS -> one barrier -> LayerNorm statistics of the full row
//     -> scores s[t][h] = rstd_t (x_t . m'_h - mean_t sum(m'_h)); weights w = exp(s - ref_h): ref_h is either a static
//        upper bound (|s| <= sqrt(D) |m'_h|, used when it cannot underflow) or the running maximum (online softmax)
//     -> Z[h][cols] += sum_t w[t][h] xn[t][cols] as MFMA 16x16x16: the A operand (weights: head = lane&15, tokens
//        4 (lane>>4) + r) and the B operand (normalised tile: column = lane&15, same tokens) are exactly the C/D layout
//        the score and x accumulators already have, so nothing moves between lanes.
//   Output z[b][h][:] = Z[h][:] / L[h]  ([n][H][D] bf16).  No HBM traffic besides the board, the (L2-resident) constants
//   and 8 KB of output per board.  The per-token constants are padded to whole tiles by the caller: padding rows of
//   cpos are 0 and padding rows of the score columns are -1e30, which makes their softmax weight exactly 0.
// =====================================================================================================
namespace {

typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

// four floats -> the 4 x bf16 operand of v_mfma_f32_16x16x16_bf16 with two v_cvt_pk_bf16_f32
__device__ __forceinline__ s16x4 pack4_bf16(f32x2 lo, f32x2 hi) {
    const u32x2 p = {__builtin_bit_cast(unsigned, __builtin_convertvector(lo, bf16x2)), __builtin_bit_cast(unsigned, __builtin_convertvector(hi, bf16x2))};
    return __builtin_bit_cast(s16x4, p);
}

struct EmbedPoolArgs {
    const void *boards;
    int boards_f32;
    const __hip_bfloat16 *wt;   // [D + 16][KP]: conv weight rows, then the 16 extra rows (head scores; row 15 = column mean)
    const float *cpos;          // accumulator order [tiles][4 waves][8][64 lanes][4 rows]: cpos[16 tile + 4 (lane>>4) + r][128 wave + 8 (lane&15) + q], rows >= T zero
    const float *mtab;          // accumulator order [tiles][64 lanes][4 rows]: score constants [16 tile + 4 (lane>>4) + r][lane&15]; rows >= T: -1e30 in the head columns
    const float *msum;          // [16]
    const float *sref;          // [16] static per-head reference (upper bound of the scores) or null = running maximum
    __hip_bfloat16 *z;          // [n][NH][D]
    const int *count;
    int n, C, R, Cc, ksz, T;
    float eps;
    azk_leaf_source src;        // SRC variant only
};

// SRC: the boards are the engine's pending leaves (azk_leaf_source): the kernel builds the prefix over the leaf flags itself
// (board j = the j-th flagged game, ascending game order = azk_step_gather's order), reads the cell codes of that game,
// records the slot for the next expansion and publishes the leaf count - no compaction launch, no evaluator batch.
template <int KS, int NH, bool STATIC_REF, bool SRC>
__global__ __launch_bounds__(256, 2) void k_embed_pool(EmbedPoolArgs a) {
    constexpr int D = 512, KP = 32 * KS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4 *alut = (uint4 *)smem;                                  // [256] A fragment of 8 patch bits (bit q -> bf16 1.0 in slot q)
    const int Tp16 = ((a.T + 15) >> 4) * 16;
    float *part = (float *)(alut + 256);                          // [2 parities][16 rows][4 waves] partial sums of squares
    uint4 *pbits = (uint4 *)(part + 128);                         // [Tp] patch bits per token (<= 128 bits)
    int *scan = (int *)(pbits + Tp16);                            // SRC: [4] wave totals, [16] games of this workgroup's next leaves

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    int nvalid, my_first = 0, my_count = 0, my_lo = 0, my_per = 0;
    if (SRC) {
        // exclusive prefix of the leaf flags over the workgroup's 256 threads (thread t owns games [t per, (t+1) per))
        my_per = ((((a.src.n_games + 255) >> 8) + 7) >> 3) << 3;
        my_lo = tid * my_per;
        for (int w = 0; w < my_per; w += 8)
            if (my_lo + w < a.src.flag_bytes) {
                const unsigned long long f = *(const unsigned long long *)(a.src.leaf_flag + my_lo + w);
                my_count += __popcll((((f & 0x7f7f7f7f7f7f7f7full) + 0x7f7f7f7f7f7f7f7full) | f) & 0x8080808080808080ull);   // non-zero flag bytes
            }
        int incl = my_count;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off); if (lane >= off) incl += v; }
        if (lane == 63) scan[wave] = incl;
        __syncthreads();
        int before = 0;
        for (int w = 0; w < wave; w++) before += scan[w];
        my_first = before + incl - my_count;
        nvalid = scan[0] + scan[1] + scan[2] + scan[3];
        if (blockIdx.x == 0 && tid == 0) { *a.src.n_leaf = nvalid; if (a.src.cache_stamp) *a.src.cache_stamp += 1u; }
        __syncthreads();
    } else {
        nvalid = a.count ? min(a.n, *a.count) : a.n;
    }
    if ((int)blockIdx.x >= nvalid) return;

    // this wave's weight fragments (its 8 column tiles + the extra tile) live in registers for the whole kernel: 9 * KS * 4
    // VGPRs instead of 9 * KS LDS reads per token tile.  Fragment (tile, s) of lane l = wt[col(tile, l)][32 s + 8 (l>>4) .. +8]
    union BF { uint4 u; bf16x8 v; };
    BF bw[8][KS], be[KS];
#pragma unroll
    for (int q = 0; q < 8; q++)
#pragma unroll
        for (int s = 0; s < KS; s++) bw[q][s].u = *(const uint4 *)(a.wt + (size_t)(128 * wave + 8 * l15 + q) * KP + 32 * s + 8 * l4);
#pragma unroll
    for (int s = 0; s < KS; s++) be[s].u = *(const uint4 *)(a.wt + (size_t)(D + l15) * KP + 32 * s + 8 * l4);
    {
        unsigned r[4];
#pragma unroll
        for (int i = 0; i < 4; i++) r[i] = (((tid >> (2 * i)) & 1) ? 0x3F80u : 0u) | (((tid >> (2 * i + 1)) & 1) ? 0x3F800000u : 0u);
        alut[tid] = make_uint4(r[0], r[1], r[2], r[3]);
    }
    __syncthreads();

    const int RC = a.R * a.Cc, T = a.T, ksz = a.ksz, kk = ksz * ksz, pad = ksz / 2, ncell = a.C * RC;
    const int tiles = (T + 15) >> 4;
    const float msum = a.msum[l15];
    const float sref = STATIC_REF ? a.sref[l15] : 0.f;
    const bool headlane = l15 < NH;
    const f32x4 *cbase = (const f32x4 *)a.cpos + (size_t)wave * 8 * 64 + lane;
    const f32x4 *mbase = (const f32x4 *)a.mtab + lane;
    int par = 0;

    for (int leaf0 = blockIdx.x; leaf0 < nvalid; leaf0 += 16 * gridDim.x) {
      if (SRC) {
          // the games behind this workgroup's next (up to 16) leaves, resolved in one pass: the thread whose flag range holds
          // the leaf-th flagged game finds it, records its slot for the next expansion and posts the game index
          __syncthreads();
          for (int k = 0; k < 16; k++) {
              const int lf = leaf0 + k * (int)gridDim.x;
              if (lf >= nvalid) break;
              if (lf >= my_first && lf < my_first + my_count) {
                  int kk = lf - my_first, g = my_lo;
                  for (int w = 0; w < my_per; w++) {
                      const int f = a.src.leaf_flag[my_lo + w];
                      if (f && kk-- == 0) { g = my_lo + w; break; }
                  }
                  scan[4 + k] = g;
                  a.src.leaf_slot[g] = lf;
              }
          }
          __syncthreads();
      }
      for (int k16 = 0; k16 < 16; k16++) {
        const int leaf = leaf0 + k16 * (int)gridDim.x;
        if (leaf >= nvalid) break;
        int game = 0, player = 0;
        if (SRC) {
            game = scan[4 + k16];
            player = (a.src.to_move[game] + a.src.leaf_depth[game]) & 1;     // node.currentPlayer at the leaf
        }
        unsigned wbits = 0;                             // lane i holds bits [32 (i-1), 32 i) of the board bit string (lane 0: zeros)
        for (int q = 0; q * 64 < ncell; q++) {
            const int e = q * 64 + lane;
            bool on = false;
            if (SRC) {
                if (e < ncell) {                                     // canonical planes from the cell codes (gomoku.py:34-40; 3-plane: mcts.py:126-137)
                    const int ch = (e >= RC) + (e >= 2 * RC), cell = e - ch * RC;
                    const int code = a.src.leaf_cells[(size_t)game * a.src.rc_pad + cell];
                    on = ch == 2 ? player != 0 : ((code >> (ch ^ player)) & 1) != 0;
                }
            } else if (e < ncell)
                on = a.boards_f32 ? ((const float *)a.boards)[(size_t)leaf * ncell + e] != 0.0f
                                  : (((const unsigned short *)a.boards)[(size_t)leaf * ncell + e] & 0x7fff) != 0;
            const unsigned long long m = __ballot(on);
            if ((lane - 1) >> 1 == q && lane >= 1) wbits = ((lane - 1) & 1) ? (unsigned)(m >> 32) : (unsigned)m;
        }
        // ---- patch bits of every token, once per board: token = wave * 64 + lane (+ 256 per round) ----
        for (int t0 = 0; t0 < tiles * 16; t0 += 256) {
            const int t = t0 + wave * 64 + lane;
            unsigned long long plo = 0, phi = 0;
            const int j = t - 1, r = j / a.Cc, c = j - r * a.Cc;
            const bool live = t >= 1 && t < T;
            unsigned colmask = 0;
            for (int kx = 0; kx < ksz; kx++) { const int cc = c + kx - pad; if (cc >= 0 && cc < a.Cc) colmask |= 1u << kx; }
            for (int ch = 0; ch < a.C; ch++)
                for (int ky = 0; ky < ksz; ky++) {
                    const int rr = r + ky - pad;
                    // every lane takes part in the shuffles; dead rows contribute zero bits
                    int off = 32 + ch * RC + (rr < 0 ? 0 : (rr >= a.R ? a.R - 1 : rr)) * a.Cc + (c - pad);
                    if (!live) off = 32;
                    const int wi = off >> 5, sh = off & 31;
                    const unsigned lo = __shfl(wbits, wi), hi = __shfl(wbits, wi + 1);
                    unsigned bits = __funnelshift_r(lo, hi, sh) & colmask;
                    if (!live || rr < 0 || rr >= a.R) bits = 0;
                    const int p0 = ch * kk + ky * ksz;
                    if (p0 < 64) { plo |= (unsigned long long)bits << p0; if (p0 + ksz > 64) phi |= (unsigned long long)bits >> (64 - p0); }
                    else phi |= (unsigned long long)bits << (p0 - 64);
                }
            if (t < tiles * 16) pbits[t] = make_uint4((unsigned)plo, (unsigned)(plo >> 32), (unsigned)phi, (unsigned)(phi >> 32));
        }
        __syncthreads();
        f32x4 Z[8];
#pragma unroll
        for (int q = 0; q < 8; q++) Z[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        float M = -INFINITY, L = 0.f;                   // per head (lane&15 < NH); L is this lane>>4 group's share

        const f32x4 *cp = cbase;
        const f32x4 *mp = mbase;
        for (int tile = 0; tile < tiles; tile++, cp += 4 * 8 * 64, mp += 64) {
            // ---- accumulators: rows 4 (lane>>4) + r4, columns 128 wave + 8 (lane&15) + q; the constants are stored in
            //      this very order, so each accumulator is one 16-byte load, 1 KB contiguous per wave ----
            f32x4 acc[8], acce = *mp;
#pragma unroll
            for (int q = 0; q < 8; q++) acc[q] = cp[q * 64];
            // ---- A fragments: 8 patch bits of this lane's token (row lane&15) per k-step -> table ----
            const uint4 pb = pbits[tile * 16 + l15];
            const unsigned pw[4] = {pb.x, pb.y, pb.z, pb.w};
            bf16x8 afrag[KS];
#pragma unroll
            for (int s = 0; s < KS; s++) {
                union { uint4 u; bf16x8 v; } af;
                af.u = alut[(pw[s] >> (8 * l4)) & 0xffu];
                afrag[s] = af.v;
            }
#pragma unroll
            for (int s = 0; s < KS; s++) acce = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag[s], be[s].v, acce, 0, 0, 0);
#pragma unroll
            for (int s = 0; s < KS; s++)
#pragma unroll
                for (int q = 0; q < 8; q++) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag[s], bw[q][s].v, acc[q], 0, 0, 0);
            // ---- LayerNorm statistics of the full rows.  The mean is GEMM column 15 of the extra tile; only the sum of
            //      squares needs this wave's 128 columns -> LDS -> all four waves ----
            float mean[4];
#pragma unroll
            for (int r4 = 0; r4 < 4; r4++) mean[r4] = __shfl(acce[r4], (lane & 48) | 15);
            f32x2 q01 = {0.f, 0.f}, q23 = {0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const f32x2 lo = {acc[q][0], acc[q][1]}, hi = {acc[q][2], acc[q][3]};
                q01 = __builtin_elementwise_fma(lo, lo, q01);
                q23 = __builtin_elementwise_fma(hi, hi, q23);
            }
            // the row pairs (0,1) and (2,3) travel as float2 from here on: every step below works on adjacent register pairs
            // (packed-math operands), nothing has to be shuffled into place
            const f32x2 pq01 = {row16_sum(q01[0]), row16_sum(q01[1])}, pq23 = {row16_sum(q23[0]), row16_sum(q23[1])};
            f32x2 *part2 = (f32x2 *)part;                             // [parity][8 row pairs][4 waves]
            if (l15 == 0) {
                part2[(par * 8 + 2 * l4) * 4 + wave] = pq01;
                part2[(par * 8 + 2 * l4 + 1) * 4 + wave] = pq23;
            }
            __syncthreads();
            const f32x4 *pp = (const f32x4 *)(part2 + (par * 8 + 2 * l4) * 4);
            const f32x4 a0 = pp[0], a1 = pp[1], b0 = pp[2], b1 = pp[3];   // pair (0,1): waves 0,1 | 2,3; pair (2,3): likewise
            const f32x2 s01 = (f32x2{a0[0], a0[1]} + f32x2{a0[2], a0[3]}) + (f32x2{a1[0], a1[1]} + f32x2{a1[2], a1[3]});
            const f32x2 s23 = (f32x2{b0[0], b0[1]} + f32x2{b0[2], b0[3]}) + (f32x2{b1[0], b1[1]} + f32x2{b1[2], b1[3]});
            const f32x2 mean01 = {mean[0], mean[1]}, mean23 = {mean[2], mean[3]};
            const f32x2 invD = {1.0f / (float)D, 1.0f / (float)D};
            const f32x2 v01 = __builtin_elementwise_fma(-mean01, mean01, s01 * invD), v23 = __builtin_elementwise_fma(-mean23, mean23, s23 * invD);
            const f32x2 r01 = {__builtin_amdgcn_rsqf(fmaxf(v01[0], 0.f) + a.eps), __builtin_amdgcn_rsqf(fmaxf(v01[1], 0.f) + a.eps)};
            const f32x2 r23 = {__builtin_amdgcn_rsqf(fmaxf(v23[0], 0.f) + a.eps), __builtin_amdgcn_rsqf(fmaxf(v23[1], 0.f) + a.eps)};
            const f32x2 h01 = -mean01 * r01, h23 = -mean23 * r23;    // xn = x * rstd + shift
            par ^= 1;
            // ---- scores (head = lane&15, tokens 4 (lane>>4) + r4) and softmax weights; lanes >= NH carry harmless finite
            //      values into rows of Z that are never stored ----
            const f32x2 ms2 = {msum, msum};
            const f32x2 sc01 = r01 * __builtin_elementwise_fma(-mean01, ms2, f32x2{acce[0], acce[1]});
            const f32x2 sc23 = r23 * __builtin_elementwise_fma(-mean23, ms2, f32x2{acce[2], acce[3]});
            const float sc[4] = {sc01[0], sc01[1], sc23[0], sc23[1]};
            float w[4];
            if (STATIC_REF) {
#pragma unroll
                for (int r4 = 0; r4 < 4; r4++) w[r4] = __expf(sc[r4] - sref);
            } else {
                float tmax = fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3]));
                tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
                tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
                const float Mn = fmaxf(M, tmax);
                const float f = __expf(M - Mn);                       // M = -inf on the first tile: f = 0 (Z and L are 0)
#pragma unroll
                for (int r4 = 0; r4 < 4; r4++) w[r4] = __expf(sc[r4] - Mn);
                L *= f;
                M = Mn;
                if (__ballot(headlane && f != 1.0f) != 0ull) {        // some head's running maximum moved: rescale its Z rows
                    float fr[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) fr[j] = l4 < 2 ? __shfl(f, 4 * l4 + j) : 1.0f;   // rows of Z = heads 4 (lane>>4) + j
#pragma unroll
                    for (int q = 0; q < 8; q++)
#pragma unroll
                        for (int j = 0; j < 4; j++) Z[q][j] *= fr[j];
                }
            }
            L += (w[0] + w[1]) + (w[2] + w[3]);
            // ---- Z += W^T Xn : A = weights (bf16), B = normalised tile (bf16), both already in operand layout ----
            const s16x4 wa = pack4_bf16(f32x2{w[0], w[1]}, f32x2{w[2], w[3]});
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const f32x2 lo = {acc[q][0], acc[q][1]}, hi = {acc[q][2], acc[q][3]};
                const f32x2 vlo = __builtin_elementwise_fma(lo, r01, h01), vhi = __builtin_elementwise_fma(hi, r23, h23);   // (x - mean) * rstd
                Z[q] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wa, pack4_bf16(vlo, vhi), Z[q], 0, 0, 0);
            }
        }
        // ---- z[b][h][:] = Z[h][:] / L[h] ----
        float Lt = L + __shfl_xor(L, 16);
        Lt += __shfl_xor(Lt, 32);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int head = 4 * l4 + j;
            const float Lh = __shfl(Lt, head & 15);
            if (head < NH) {
                const float inv = 1.0f / Lh;
                float v[8];
#pragma unroll
                for (int q = 0; q < 8; q++) v[q] = Z[q][j] * inv;
                *(uint4 *)(a.z + ((size_t)leaf * NH + head) * D + 128 * wave + 8 * l15) = pack8(v);
            }
        }
      }
    }
}

template <int KS, int NH, bool SR, bool SRC>
int launch_embed_pool2(const EmbedPoolArgs &a, hipStream_t st) {
    const int lds = 256 * 16 + 512 + ((a.T + 15) / 16) * 16 * 16 + 96;
    if (azk_set_max_lds((const void *)k_embed_pool<KS, NH, SR, SRC>, lds) != hipSuccess) return AZK_ERR_HIP;
    const int blocks = a.n < 512 ? a.n : 512;                      // two resident workgroups per CU, each walks its boards
    k_embed_pool<KS, NH, SR, SRC><<<blocks, 256, lds, st>>>(a);
    return hipGetLastError() == hipSuccess ? AZK_OK : AZK_ERR_HIP;
}

template <int KS, int NH>
int launch_embed_pool(const EmbedPoolArgs &a, hipStream_t st) {
    if (a.src.leaf_flag) return a.sref ? launch_embed_pool2<KS, NH, true, true>(a, st) : launch_embed_pool2<KS, NH, false, true>(a, st);
    return a.sref ? launch_embed_pool2<KS, NH, true, false>(a, st) : launch_embed_pool2<KS, NH, false, false>(a, st);
}
}  // namespace

static int32_t embed_pool_impl(const void *boards_dev, int32_t boards_are_f32, const azk_leaf_source *src, const void *wt_ext_bf16_dev,
                               const float *cpos_frag_dev, const float *score_frag_dev, const float *score_msum_dev,
                               const float *score_ref_dev, void *z_out_bf16_dev, int32_t num_heads, int32_t n, int32_t channels,
                               int32_t rows, int32_t cols, int32_t ksize, int32_t kp, int32_t embed_dim, float ln_eps,
                               const int32_t *n_valid_dev, void *stream) {
    if ((!boards_dev && !src) || !wt_ext_bf16_dev || !cpos_frag_dev || !score_frag_dev || !score_msum_dev || !z_out_bf16_dev) return AZK_ERR_ARG;
    if (n < 0 || channels < 1 || rows < 1 || cols < 1 || ksize < 1 || (ksize & 1) == 0 || ksize > 7) return AZK_ERR_ARG;
    if (kp < channels * ksize * ksize || kp % 32 != 0 || kp > 96) return AZK_ERR_ARG;
    if (channels * rows * cols > 62 * 32 || embed_dim != 512) return AZK_ERR_ARG;      // one column group per wave, four waves
    if (num_heads != 8 && num_heads != 4) return AZK_ERR_ARG;
    if (n == 0) return AZK_OK;
    EmbedPoolArgs a;
    memset(&a, 0, sizeof a);
    a.boards = boards_dev; a.boards_f32 = boards_are_f32; a.wt = (const __hip_bfloat16 *)wt_ext_bf16_dev; a.cpos = cpos_frag_dev;
    a.mtab = score_frag_dev; a.msum = score_msum_dev; a.sref = score_ref_dev; a.z = (__hip_bfloat16 *)z_out_bf16_dev;
    a.count = n_valid_dev;
    a.n = n; a.C = channels; a.R = rows; a.Cc = cols; a.ksz = ksize; a.T = rows * cols + 1; a.eps = ln_eps;
    if (src) a.src = *src;
    hipStream_t st = (hipStream_t)stream;
    const int ks = kp / 32;
#define CASE(KS_, NH_) if (ks == KS_ && num_heads == NH_) return launch_embed_pool<KS_, NH_>(a, st)
    CASE(2, 8); CASE(1, 8); CASE(3, 8); CASE(2, 4); CASE(1, 4); CASE(3, 4);
#undef CASE
    return AZK_ERR_ARG;
}

extern "C" int32_t azk_nn_embed_pool(const void *boards_dev, int32_t boards_are_f32, const void *wt_ext_bf16_dev,
                                     const float *cpos_frag_dev, const float *score_frag_dev, const float *score_msum_dev,
                                     const float *score_ref_dev, void *z_out_bf16_dev, int32_t num_heads, int32_t n,
                                     int32_t channels, int32_t rows, int32_t cols, int32_t ksize, int32_t kp, int32_t embed_dim,
                                     float ln_eps, const int32_t *n_valid_dev, void *stream) {
    if (!boards_dev) return AZK_ERR_ARG;
    return embed_pool_impl(boards_dev, boards_are_f32, nullptr, wt_ext_bf16_dev, cpos_frag_dev, score_frag_dev, score_msum_dev,
                           score_ref_dev, z_out_bf16_dev, num_heads, n, channels, rows, cols, ksize, kp, embed_dim, ln_eps, n_valid_dev, stream);
}

extern "C" int32_t azk_nn_embed_pool_leaves(const azk_leaf_source *src, const void *wt_ext_bf16_dev, const float *cpos_frag_dev,
                                            const float *score_frag_dev, const float *score_msum_dev, const float *score_ref_dev,
                                            void *z_out_bf16_dev, int32_t num_heads, int32_t ksize, int32_t kp, int32_t embed_dim,
                                            float ln_eps, void *stream) {
    if (!src || !src->leaf_flag || !src->leaf_cells || !src->to_move || !src->leaf_depth || !src->leaf_slot || !src->n_leaf) return AZK_ERR_ARG;
    if (src->n_games < 1 || src->rows * src->cols != src->rc || src->flag_bytes < src->n_games) return AZK_ERR_ARG;
    return embed_pool_impl(nullptr, 0, src, wt_ext_bf16_dev, cpos_frag_dev, score_frag_dev, score_msum_dev, score_ref_dev,
                           z_out_bf16_dev, num_heads, src->n_games, src->planes, src->rows, src->cols, ksize, kp, embed_dim, ln_eps, nullptr, stream);
}

// =====================================================================================================
// k_embed_pool_c: k_embed_pool that only computes the tokens a stone can reach.
//   A token whose k x k patch holds no stone is a constant of the weights: x_t = cpos[t], so its normalised row xn_t, its
//   head scores and - with the static softmax reference - its weights w_t[h] = exp(s_t[h] - ref[h]) do not depend on the
//   board.  With  ZALL[h] = sum_t wc_t[h] xnc_t  and  LALL[h] = sum_t wc_t[h]  over ALL tokens taken as empty-patch tokens,
//       Z[h] = ZALL[h] + sum_{t dirty} (w_t[h] xn_t - wc_t[h] xnc_t),     L[h] = LALL[h] + sum_{t dirty} (w_t[h] - wc_t[h])
//   exactly (the softmax reference is the same constant on both sides).  On a 15x15 board with ~20 stones ~100 of the 226
//   tokens are dirty: 7 sixteen-token tiles instead of 15.  Per board: patch bits of all tokens (one per thread), the
//   dirty ones compacted through LDS (ballot + prefix), then the tile loop of k_embed_pool over the compacted list with
//   the per-token constants GATHERED by token index ([token][...] tables, L2 resident); the subtraction rides on the same
//   MFMA: v_mfma_f32_16x16x32_bf16 with k-slots 0..3 of a lane group = its four tokens (A: w, B: xn) and k-slots 4..7 = the
//   same tokens as constants (A: -wc, B: xnc).
//   Scheduling: a workgroup's first board is blockIdx.x; further boards come from a device-side queue head (one atomic per
//   board, issued behind the first tile's loads so its round trip hides under the tile), because boards now differ in
//   cost.  Exactly n_valid tickets are drawn per launch (every workgroup with a board draws until one fails), so the
//   workgroup holding ticket n_valid - 1 knows the queue is finished and leaves the counter zero for the next launch.
// =====================================================================================================
namespace {

struct EmbedPoolCArgs {
    const void *boards;
    int boards_f32;
    const void *wt_frag;           // conv weight (+ the 16 extra columns) in MFMA fragment order [33][KS][64] x 16 bytes
    const float *cposT;            // [T + 1][D]   bias + positional term per token; row T (the null token) = 0
    const float *scoreT;           // [T + 1][16]  score constants per token (column 15: row mean); row T: -1e30 in the head columns
    const float *wcT;              // [T + 1][16]  softmax weight of the token taken as an empty-patch token; row T = 0
    const __hip_bfloat16 *xncT;    // [T + 1][D]   normalised empty-patch token (bf16); row T = 0
    const float *zall;             // accumulator order [4 waves][8][64 lanes][4]: ZALL[head 4 (lane>>4) + j][128 w + 8 (lane&15) + q]
    const float *lall;             // [16]
    const float *msum, *sref;      // [16]
    __hip_bfloat16 *z;             // [n][NH][D]
    const int *count;
    int *sched;                    // [1]: ticket counter of the board queue; zero between launches
    long long *dbg;                // debug only (AZK_EMBED_POOL_STAMPS): [8] cycle sums per phase, wave 0 of every workgroup
    unsigned long long *wstats;    // optional [2]: boards evaluated, 16-token tiles evaluated (fire-and-forget atomics, one pair per board)
    int n, R, Cc, T;
    float eps;
    azk_leaf_source src;
};

template <int NC, int KSZ, int NH, bool SRC>
__global__ __launch_bounds__(256, 2) void k_embed_pool_c(EmbedPoolCArgs a) {
    constexpr int KS = (NC * KSZ * KSZ + 31) / 32;
    constexpr int D = 512, KP = 32 * KS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4 *alut = (uint4 *)smem;                                  // [256] A fragment of 8 patch bits
    float *part = (float *)(alut + 256);                          // [2 parities][16 rows][4 waves] partial sums of squares
    const int Tp16 = ((a.T + 15) >> 4) << 4;
    uint4 *pbits = (uint4 *)(part + 128);                         // [Tp16] patch bits of the compacted dirty tokens
    int *dlist = (int *)(pbits + Tp16);                           // [Tp16] their token indices (null token = T past the end)
    int *scan = dlist + Tp16;                                     // [4] SRC wave totals, [4] dirty counts per wave, [8] next board, [9] game
    uint4 *rankv = (uint4 *)(scan + 32);                          // SRC: [256 threads] ranks of the thread's first eight games, 16 bits each (scan[16..31]: class totals of the four waves)
    uint4 *bimg = rankv + (SRC ? 256 : 0);                        // [33 column tiles][KS][64 lanes] weight B fragments                           // [33 column tiles][KS][64 lanes] weight B fragments (the wave's 8 tiles + the extra one)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    // SRC: a non-zero leaf flag is 1 + the leaf's cost class (0..7, by stone count).  Board j of the launch is the j-th flagged
    // game in the order (class descending, game index ascending): the stone-heavy boards - the ones with the most tokens to
    // evaluate - are handed out first, the light ones fill the gaps at the end (longest-processing-time-first; the queue is
    // dynamic).  Every workgroup derives the same ranks: per-class counts of its threads' games (thread t owns games
    // [t per, (t+1) per)), an exclusive scan over the 256 threads with the eight 16-bit counters packed in two 64-bit words.
    int nvalid, my_lo = 0, my_per = 0;
    unsigned long long cb_lo = 0ull, cb_hi = 0ull;                // rank of this thread's first game of each class, 16 bits each (classes 0-3 / 4-7)
    // The launch's fixed cost is a chain of round trips (leaf flags -> ranks -> weights -> first board): the flag words and the
    // whole conv weight image (wt_frag, MFMA fragment order [33 column tiles][KS][64 lanes] x 16 bytes: fragment (tile, s) of lane l
    // = wt[col(tile, l)][32 s + 8 (l>>4) .. +8], column tile 32 = the extra columns) are requested together, the flags first so
    // that the rank arithmetic waits for them alone; the image goes to LDS once the ranks are done.
    constexpr int NF = 33 * KS * 64, PER = (NF + 255) / 256;
    unsigned long long myflags = 0ull;
    if (SRC) {
        my_per = ((((a.src.n_games + 255) >> 8) + 7) >> 3) << 3;
        my_lo = tid * my_per;
        if (my_lo < a.src.flag_bytes) myflags = *(const unsigned long long *)(a.src.leaf_flag + my_lo);
    }
    // (LDS-DMA: a wave instruction moves 64 x 16 contiguous bytes, no registers; NF is a multiple of 64, whole wave pieces only)
    static_assert(NF % 64 == 0, "the weight image is copied in whole 1 KiB wave pieces");
#pragma unroll
    for (int i = 0; i < PER; i++)
        if (256 * i + 64 * wave < NF)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((const uint4 *)a.wt_frag + tid + 256 * i),
                                             (__attribute__((address_space(3))) void *)(bimg + 256 * i + 64 * wave), 16, 0, 0);
    if (SRC) {
        unsigned long long c_lo = 0ull, c_hi = 0ull;              // classes 0-3 / 4-7, 16 bits each
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
        const unsigned long long e_lo = b_lo + i_lo - c_lo, e_hi = b_hi + i_hi - c_hi;   // exclusive prefix over lower threads, per class
        unsigned start = 0;
#pragma unroll
        for (int c = 7; c >= 0; c--) {                            // class 7 (most stones) first
            const unsigned tot = (unsigned)(((c < 4 ? t_lo : t_hi) >> (16 * (c & 3))) & 0xffffull);
            const unsigned long long cb = (unsigned long long)(start + (unsigned)(((c < 4 ? e_lo : e_hi) >> (16 * (c & 3))) & 0xffffull)) << (16 * (c & 3));
            if (c < 4) cb_lo |= cb; else cb_hi |= cb;
            start += tot;
        }
        nvalid = (int)start;
        unsigned run = 0;                                         // games of each class seen so far in this thread: 4 bits each
        unsigned myrank[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};   // 0xffff: no game
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
        rankv[tid] = make_uint4(myrank[0], myrank[1], myrank[2], myrank[3]);      // read back by the same thread only
        if (blockIdx.x == 0 && tid == 0) { *a.src.n_leaf = nvalid; if (a.src.cache_stamp) *a.src.cache_stamp += 1u; }
    } else {
        nvalid = a.count ? min(a.n, *a.count) : a.n;
    }
    int board = blockIdx.x;
#ifdef AZK_EP_STAMPS          // diagnosis build only (make EXTRA=-DAZK_EP_STAMPS): the stamps cost registers, the product kernel has none
    const bool stamp = a.dbg != nullptr && tid == 0;
    long long tp = stamp ? clock64() : 0, tacc[5] = {0, 0, 0, 0, 0}, nt_acc = 0, nb_acc = 0;     // sums stay in registers until the end
#define AZK_STAMP(i) do { if (stamp) { const long long tn_ = clock64(); tacc[i] += tn_ - tp; tp = tn_; } } while (0)
#else
    constexpr bool stamp = false;
    long long nt_acc = 0, nb_acc = 0;
#define AZK_STAMP(i) do { } while (0)
#endif
    int ws_boards = 0, ws_tiles = 0;
    if (board >= nvalid) __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): no LDS-DMA may outlive the workgroup
    if (board < nvalid) {                                         // (workgroups without a board go straight to the sign-off below)
    union BF { uint4 u; bf16x8 v; };
    const uint4 *bwv = bimg + (size_t)wave * 8 * KS * 64 + lane, *bev = bimg + (size_t)32 * KS * 64 + lane;
    {
        unsigned r[4];
#pragma unroll
        for (int i = 0; i < 4; i++) r[i] = (((tid >> (2 * i)) & 1) ? 0x3F80u : 0u) | (((tid >> (2 * i + 1)) & 1) ? 0x3F800000u : 0u);
        alut[tid] = make_uint4(r[0], r[1], r[2], r[3]);
    }
    constexpr int ksz = KSZ, kk = KSZ * KSZ, pad = KSZ / 2;
    const int RC = a.R * a.Cc, T = a.T, ncell = NC * RC;
    const float msum = a.msum[l15], sref = a.sref[l15], lall = a.lall[l15];
    const int colofs = 128 * wave + 8 * l15;
    int par = 0, nxt = 0;
    __syncthreads();
    AZK_STAMP(0);                                                 // prologue: weights staged

    while (board < nvalid) {
        // Everything below that depends only on the thread index (where the token's patch rows sit in the board bit string, the
        // cell each lane fetches, the cross-lane read addresses) is recomputed per board from an opaque copy of the index: left
        // to the compiler these ~60 values are hoisted out of the board loop, live through the tile loop, and the spills they
        // cause are reloaded between the board's loads - one memory round trip per reload.
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
            // the game behind board `board`: the thread that owns the game with that rank posts it and records the slot the next
            // expansion reads (its first eight games' ranks sit in registers; engines with more than 2048 slots walk the rest)
            int g = -1;
            const uint4 rk = rankv[tid];
            const unsigned myrank[4] = {rk.x, rk.y, rk.z, rk.w};
#pragma unroll
            for (int q = 0; q < 8; q++) if (((myrank[q >> 1] >> (16 * (q & 1))) & 0xffffu) == (unsigned)board) g = my_lo + q;
            if (my_per > 8) {
                unsigned long long run2 = 0ull;                    // games of each class seen so far: 8 bits each
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
        // the workgroup's Z starts at the constant part (fetched here, under the board's own loads)
        f32x4 Z[8];
        if (l4 < (NH + 3) / 4) {
            const f32x4 *zp = (const f32x4 *)a.zall + (size_t)wave * 8 * 64 + lane;
#pragma unroll
            for (int q = 0; q < 8; q++) Z[q] = zp[q * 64];
        } else {
#pragma unroll
            for (int q = 0; q < 8; q++) Z[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        unsigned wbits = 0;                             // lane i holds bits [32 (i-1), 32 i) of the board bit string (lane 0: zeros)
        constexpr int NQ = 8;                           // boards of up to 512 plane cells: every load of the board in ONE round trip
        if (ncell <= 64 * NQ) {
            bool on[NQ];
            if (SRC) {
                // canonical planes from the cell codes (gomoku.py:34-40; 3-plane: mcts.py:126-137); the code loads do not
                // depend on the side to move, so they travel together with the two words that give it
                int code[NQ], chq[NQ];
                const auto *cells = a.src.leaf_cells + (size_t)game * a.src.rc_pad;       // uniform base + 32-bit lane offsets
#pragma unroll
                for (int q = 0; q < NQ; q++) {
                    const int e = min(q * 64 + lane_b, ncell - 1);
                    chq[q] = (e >= RC) + (e >= 2 * RC);
                    code[q] = cells[(unsigned)(e - chq[q] * RC)];
                }
                player = (a.src.to_move[game] + a.src.leaf_depth[game]) & 1;     // node.currentPlayer at the leaf
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
        AZK_STAMP(1);                                             // board resolved, loaded, bit string built
        // ---- patch bits of this thread's token; dirty = some stone in the patch ----
        unsigned long long plo = 0, phi = 0;
        {
            // compile-time trip counts: all 2 NC KSZ cross-lane reads of the bit string are issued together
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
                    constexpr int dummy = 0; (void)dummy;
                    const int p0 = ch * kk + ky * ksz;
                    if (p0 < 64) { plo |= (unsigned long long)bits << p0; if (p0 + ksz > 64) phi |= (unsigned long long)bits >> (64 - p0); }
                    else phi |= (unsigned long long)bits << (p0 - 64);
                }
        }
        const bool dirty = (plo | phi) != 0ull;
        const unsigned long long dm = __ballot(dirty);
        if (lane == 0) scan[4 + wave] = __popcll(dm);
        __syncthreads();                                  // (also: every wave is done with the previous board's lists)
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

        AZK_STAMP(2);                                             // patch bits + compaction
        if (stamp) nt_acc += ntile;
        ws_boards += 1; ws_tiles += ntile;                  // (uniform; one pair of atomics per workgroup at the very end: an atomic here sits in
                                                            //  the vmcnt queue in front of the tile's gathers, which wait for it - measured +7 us per launch)
        float L = 0.f;                                    // per head (lane&15 < NH): this lane>>4 group's share of sum (w - wc)
        // The per-token constants are GATHERED (by token index, L2) and every tile would wait a full round trip for them, so they
        // run one phase ahead: the conv MFMAs start from zero and the constants are added behind them; the registers they leave
        // are refilled with the NEXT tile's constants before the statistics / pooling phase, and the constant rows of the pooling
        // (needed last) are refetched right behind their use.  Same registers, the round trip under the other phase's arithmetic.
        f32x4 c0[4], c1[4], scn, wcn;
        uint4 xr[4];
        auto gather_a = [&](int t) {
            const int4 tk = *(const int4 *)(dlist + 16 * t + 4 * l4);
            const int tks[4] = {tk.x, tk.y, tk.z, tk.w};
#pragma unroll
            for (int r = 0; r < 4; r++) {
                // 32-bit BYTE offsets from the (uniform) table bases: the loads take the base from SGPRs, one VGPR per token and table
                const unsigned orow = ((unsigned)tks[r] * (unsigned)D + (unsigned)colofs) * 4u, osc = ((unsigned)tks[r] * 16u + (unsigned)l15) * 4u;
                c0[r] = *(const f32x4 *)((const char *)a.cposT + orow); c1[r] = *(const f32x4 *)((const char *)a.cposT + orow + 16);
                scn[r] = *(const float *)((const char *)a.scoreT + osc);
                wcn[r] = *(const float *)((const char *)a.wcT + osc);
            }
        };
        auto gather_x = [&](int t) {
            const int4 tk = *(const int4 *)(dlist + 16 * t + 4 * l4);
            const int tks[4] = {tk.x, tk.y, tk.z, tk.w};
#pragma unroll
            for (int r = 0; r < 4; r++) xr[r] = *(const uint4 *)((const char *)a.xncT + ((unsigned)tks[r] * (unsigned)D + (unsigned)colofs) * 2u);
        };
        if (ntile > 0) { gather_a(0); gather_x(0); }
        for (int tile = 0; tile < ntile; tile++) {
            if (tile == 0 && tid == 0) {                  // next board: the round trip hides under this tile
                __builtin_amdgcn_sched_barrier(0);
                nxt = atomicAdd(a.sched, 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- A fragments: 8 patch bits of this lane's token (row lane&15) per k-step -> table ----
            const uint4 pb = pbits[tile * 16 + l15];
            const unsigned pw[4] = {pb.x, pb.y, pb.z, pb.w};
            bf16x8 afrag[KS];
#pragma unroll
            for (int s = 0; s < KS; s++) {
                union { uint4 u; bf16x8 v; } af;
                af.u = alut[(pw[s] >> (8 * l4)) & 0xffu];
                afrag[s] = af.v;
            }
            f32x4 acc[8];
            f32x4 acce = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 8; q++) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            // k-step outermost: nine independent accumulator chains per step, the B fragments stream from LDS
#pragma unroll
            for (int s = 0; s < KS; s++) {
                { BF b; b.u = bev[s * 64]; acce = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag[s], b.v, acce, 0, 0, 0); }
#pragma unroll
                for (int q = 0; q < 8; q++) { BF b; b.u = bwv[(q * KS + s) * 64]; acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag[s], b.v, acc[q], 0, 0, 0); }
                if (s + 1 < KS) __builtin_amdgcn_sched_barrier(0);    // one k-step's fragments in flight at a time (VGPR budget)
            }
            // ---- the gathered constants (bias + positional term, score constants), then the next tile's gathers into their registers ----
#pragma unroll
            for (int r = 0; r < 4; r++) {
                acc[0][r] += c0[r][0]; acc[1][r] += c0[r][1]; acc[2][r] += c0[r][2]; acc[3][r] += c0[r][3];
                acc[4][r] += c1[r][0]; acc[5][r] += c1[r][1]; acc[6][r] += c1[r][2]; acc[7][r] += c1[r][3];
            }
            acce += scn;
            const f32x4 wc = wcn;
            __builtin_amdgcn_sched_barrier(0);
            const int tnext = min(tile + 1, ntile - 1);          // (the last tile refetches itself: no branch around loads)
            gather_a(tnext);
            __builtin_amdgcn_sched_barrier(0);
            // ---- LayerNorm statistics of the full rows (mean = GEMM column 15 of the extra tile) ----
            float mean[4];
#pragma unroll
            for (int r4 = 0; r4 < 4; r4++) mean[r4] = __shfl(acce[r4], (lane & 48) | 15);
            f32x2 q01 = {0.f, 0.f}, q23 = {0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const f32x2 lo = {acc[q][0], acc[q][1]}, hi = {acc[q][2], acc[q][3]};
                q01 = __builtin_elementwise_fma(lo, lo, q01);
                q23 = __builtin_elementwise_fma(hi, hi, q23);
            }
            const f32x2 pq01 = {row16_sum(q01[0]), row16_sum(q01[1])}, pq23 = {row16_sum(q23[0]), row16_sum(q23[1])};
            f32x2 *part2 = (f32x2 *)part;                             // [parity][8 row pairs][4 waves]
            if (l15 == 0) {
                part2[(par * 8 + 2 * l4) * 4 + wave] = pq01;
                part2[(par * 8 + 2 * l4 + 1) * 4 + wave] = pq23;
            }
            __syncthreads();
            const f32x4 *pp = (const f32x4 *)(part2 + (par * 8 + 2 * l4) * 4);
            const f32x4 a0 = pp[0], a1 = pp[1], b0 = pp[2], b1 = pp[3];
            const f32x2 s01 = (f32x2{a0[0], a0[1]} + f32x2{a0[2], a0[3]}) + (f32x2{a1[0], a1[1]} + f32x2{a1[2], a1[3]});
            const f32x2 s23 = (f32x2{b0[0], b0[1]} + f32x2{b0[2], b0[3]}) + (f32x2{b1[0], b1[1]} + f32x2{b1[2], b1[3]});
            const f32x2 mean01 = {mean[0], mean[1]}, mean23 = {mean[2], mean[3]};
            const f32x2 invD = {1.0f / (float)D, 1.0f / (float)D};
            const f32x2 v01 = __builtin_elementwise_fma(-mean01, mean01, s01 * invD), v23 = __builtin_elementwise_fma(-mean23, mean23, s23 * invD);
            const f32x2 r01 = {__builtin_amdgcn_rsqf(fmaxf(v01[0], 0.f) + a.eps), __builtin_amdgcn_rsqf(fmaxf(v01[1], 0.f) + a.eps)};
            const f32x2 r23 = {__builtin_amdgcn_rsqf(fmaxf(v23[0], 0.f) + a.eps), __builtin_amdgcn_rsqf(fmaxf(v23[1], 0.f) + a.eps)};
            const f32x2 h01 = -mean01 * r01, h23 = -mean23 * r23;    // xn = x * rstd + shift
            par ^= 1;
            // ---- scores (head = lane&15, tokens = rows) and softmax weights against the static reference ----
            const f32x2 ms2 = {msum, msum};
            const f32x2 sc01 = r01 * __builtin_elementwise_fma(-mean01, ms2, f32x2{acce[0], acce[1]});
            const f32x2 sc23 = r23 * __builtin_elementwise_fma(-mean23, ms2, f32x2{acce[2], acce[3]});
            float w[4];
            w[0] = __expf(sc01[0] - sref); w[1] = __expf(sc01[1] - sref); w[2] = __expf(sc23[0] - sref); w[3] = __expf(sc23[1] - sref);
            L += ((w[0] - wc[0]) + (w[1] - wc[1])) + ((w[2] - wc[2]) + (w[3] - wc[3]));
            // ---- Z += W^T Xn - Wc^T Xnc as ONE 16x16x32 MFMA per column: k-slots 0..3 actual, 4..7 constant ----
            union { bf16x8 v; s16x4 h[2]; } wa;
            wa.h[0] = pack4_bf16(f32x2{w[0], w[1]}, f32x2{w[2], w[3]});
            wa.h[1] = pack4_bf16(f32x2{-wc[0], -wc[1]}, f32x2{-wc[2], -wc[3]});
            const unsigned xw[4][4] = {{xr[0].x, xr[0].y, xr[0].z, xr[0].w}, {xr[1].x, xr[1].y, xr[1].z, xr[1].w},
                                       {xr[2].x, xr[2].y, xr[2].z, xr[2].w}, {xr[3].x, xr[3].y, xr[3].z, xr[3].w}};
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const f32x2 lo = {acc[q][0], acc[q][1]}, hi = {acc[q][2], acc[q][3]};
                const f32x2 vlo = __builtin_elementwise_fma(lo, r01, h01), vhi = __builtin_elementwise_fma(hi, r23, h23);   // (x - mean) * rstd
                union { bf16x8 v; struct { s16x4 h; unsigned c01, c23; } p; } xb;
                xb.p.h = pack4_bf16(vlo, vhi);
                const unsigned sel = (q & 1) ? 0x07060302u : 0x05040100u;                  // bf16 element q of each token's 16-byte row
                xb.p.c01 = __builtin_amdgcn_perm(xw[1][q >> 1], xw[0][q >> 1], sel);
                xb.p.c23 = __builtin_amdgcn_perm(xw[3][q >> 1], xw[2][q >> 1], sel);
                Z[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa.v, xb.v, Z[q], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            gather_x(tnext);
        }
        AZK_STAMP(3);                                             // tile loop
        if (ntile == 0 && tid == 0) nxt = atomicAdd(a.sched, 1);
        // ---- z[b][h][:] = (ZALL + Z)[h][:] / (LALL + L)[h] ----
        float Lt = L + __shfl_xor(L, 16);
        Lt += __shfl_xor(Lt, 32);
        Lt += lall;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int head = 4 * l4 + j;
            const float Lh = __shfl(Lt, head & 15);
            if (head < NH) {
                const float inv = 1.0f / Lh;
                float v[8];
#pragma unroll
                for (int q = 0; q < 8; q++) v[q] = Z[q][j] * inv;
                *(uint4 *)(a.z + ((size_t)board * NH + head) * D + colofs) = pack8(v);
            }
        }
        if (tid == 0) {
            // every workgroup that got a board draws tickets until one fails, so exactly nvalid tickets are drawn per launch:
            // whoever holds the last one (nvalid - 1) knows the queue is finished for this launch and leaves it zero
            if (nxt == nvalid - 1) a.sched[0] = 0;
            scan[8] = (int)gridDim.x + nxt;
        }
        __syncthreads();
        board = scan[8];
        AZK_STAMP(4);                                             // epilogue + next board known
        if (stamp) nb_acc += 1;
    }
    }
    if (a.wstats != nullptr && tid == 0 && ws_boards) { atomicAdd(a.wstats, (unsigned long long)ws_boards); atomicAdd(a.wstats + 1, (unsigned long long)ws_tiles); }
#undef AZK_STAMP
#ifdef AZK_EP_STAMPS
    if (stamp) {
        const long long wg_total = tacc[0] + tacc[1] + tacc[2] + tacc[3] + tacc[4];
        atomicMax((unsigned long long *)a.dbg + 5, (unsigned long long)wg_total);       // the busiest workgroup of any launch
        for (int i = 0; i < 5; i++) atomicAdd((unsigned long long *)a.dbg + i, (unsigned long long)tacc[i]);
        atomicAdd((unsigned long long *)a.dbg + 6, (unsigned long long)nt_acc);
        atomicAdd((unsigned long long *)a.dbg + 7, (unsigned long long)nb_acc);
    }
#endif
    (void)nt_acc; (void)nb_acc;
}

template <int NC, int KSZ, int NH, bool SRC>
int launch_embed_pool_c2(const EmbedPoolCArgs &a, hipStream_t st) {
    constexpr int KS = (NC * KSZ * KSZ + 31) / 32;
    const int tp16 = ((a.T + 15) / 16) * 16;
    const int lds = 256 * 16 + 512 + tp16 * 16 + tp16 * 4 + 128 + (SRC ? 256 * 16 : 0) + 33 * KS * 64 * 16;      // 77 KB at KS = 2: two workgroups per CU
    if (azk_set_max_lds((const void *)k_embed_pool_c<NC, KSZ, NH, SRC>, lds) != hipSuccess) return AZK_ERR_HIP;
    const int blocks = a.n < 512 ? a.n : 512;                      // two resident workgroups per CU; each pulls boards until the queue is dry
    k_embed_pool_c<NC, KSZ, NH, SRC><<<blocks, 256, lds, st>>>(a);
    return hipGetLastError() == hipSuccess ? AZK_OK : AZK_ERR_HIP;
}
}  // namespace

static int32_t embed_pool_c_impl(const void *boards_dev, int32_t boards_are_f32, const azk_leaf_source *src, const azk_embed_pool_consts *k,
                                 void *z_out_bf16_dev, int32_t n, int32_t channels, int32_t rows, int32_t cols,
                                 const int32_t *n_valid_dev, int32_t *sched_dev, void *stream) {
    if ((!boards_dev && !src) || !k || !z_out_bf16_dev || !sched_dev) return AZK_ERR_ARG;
    if (!k->wt_frag || !k->cpos_tok || !k->score_tok || !k->wconst_tok || !k->xnconst_tok || !k->z_all || !k->l_all || !k->score_msum || !k->score_ref) return AZK_ERR_ARG;
    const int ksize = k->ksize, kp = k->kp;
    if (n < 0 || channels < 1 || rows < 1 || cols < 1 || ksize < 1 || (ksize & 1) == 0 || ksize > 7) return AZK_ERR_ARG;
    if (kp < channels * ksize * ksize || kp % 32 != 0 || kp > 96) return AZK_ERR_ARG;
    if (channels * rows * cols > 62 * 32 || k->embed_dim != 512) return AZK_ERR_ARG;
    if (rows * cols + 1 > 256) return AZK_ERR_ARG;                 // one thread per token
    if (k->num_heads != 8 && k->num_heads != 4) return AZK_ERR_ARG;
    if (n == 0) return AZK_OK;
    EmbedPoolCArgs a;
    memset(&a, 0, sizeof a);
    a.boards = boards_dev; a.boards_f32 = boards_are_f32; a.wt_frag = k->wt_frag; a.cposT = k->cpos_tok;
    a.scoreT = k->score_tok; a.wcT = k->wconst_tok; a.xncT = (const __hip_bfloat16 *)k->xnconst_tok; a.zall = k->z_all; a.lall = k->l_all;
    a.msum = k->score_msum; a.sref = k->score_ref; a.z = (__hip_bfloat16 *)z_out_bf16_dev; a.count = n_valid_dev; a.sched = sched_dev;
    a.wstats = (unsigned long long *)k->work_stats;
    a.n = n; a.R = rows; a.Cc = cols; a.T = rows * cols + 1; a.eps = k->ln_eps;
    if (src) a.src = *src;
    {
        static long long *dbg_buf = nullptr;
        const char *ds = getenv("AZK_EMBED_POOL_STAMPS");
        if (ds && atoi(ds)) {
            if (!dbg_buf && (hipMalloc((void **)&dbg_buf, 64) != hipSuccess || hipMemset(dbg_buf, 0, 64) != hipSuccess)) return AZK_ERR_HIP;
            a.dbg = dbg_buf;
            if (atoi(ds) == 2) {          // print-and-reset request
                long long h[8];
                if (hipMemcpy(h, dbg_buf, 64, hipMemcpyDeviceToHost) != hipSuccess) return AZK_ERR_HIP;
                fprintf(stderr, "[embed_pool_c stamps] busiest workgroup %lld cycles | ", h[5]);
                fprintf(stderr, "[embed_pool_c stamps] boards %lld tiles %lld | cycles per board: prologue(total) %lld, load %.0f, patch+compact %.0f, tiles %.0f (%.0f per tile), epilogue %.0f\n",
                        h[7], h[6], h[0], (double)h[1] / (double)(h[7] ? h[7] : 1), (double)h[2] / (double)(h[7] ? h[7] : 1), (double)h[3] / (double)(h[7] ? h[7] : 1),
                        (double)h[3] / (double)(h[6] ? h[6] : 1), (double)h[4] / (double)(h[7] ? h[7] : 1));
                if (hipMemset(dbg_buf, 0, 64) != hipSuccess) return AZK_ERR_HIP;
            }
        }
    }
    hipStream_t st = (hipStream_t)stream;
    const int nh = k->num_heads;
    if (kp != (channels * ksize * ksize + 31) / 32 * 32) return AZK_ERR_ARG;
#define CASE(NC_, KSZ_, NH_) if (channels == NC_ && ksize == KSZ_ && nh == NH_) \
        return src ? launch_embed_pool_c2<NC_, KSZ_, NH_, true>(a, st) : launch_embed_pool_c2<NC_, KSZ_, NH_, false>(a, st)
    CASE(2, 5, 8); CASE(2, 5, 4); CASE(3, 5, 8); CASE(3, 5, 4); CASE(2, 3, 8); CASE(2, 3, 4); CASE(3, 3, 8); CASE(3, 3, 4);
#undef CASE
    return AZK_ERR_ARG;
}

extern "C" int32_t azk_nn_embed_pool_compact(const void *boards_dev, int32_t boards_are_f32, const azk_embed_pool_consts *consts,
                                             void *z_out_bf16_dev, int32_t n, int32_t channels, int32_t rows, int32_t cols,
                                             const int32_t *n_valid_dev, int32_t *sched_dev, void *stream) {
    if (!boards_dev) return AZK_ERR_ARG;
    return embed_pool_c_impl(boards_dev, boards_are_f32, nullptr, consts, z_out_bf16_dev, n, channels, rows, cols, n_valid_dev, sched_dev, stream);
}

extern "C" int32_t azk_nn_embed_pool_compact_leaves(const azk_leaf_source *src, const azk_embed_pool_consts *consts, void *z_out_bf16_dev,
                                                    int32_t *sched_dev, void *stream) {
    if (!src || !src->leaf_flag || !src->leaf_cells || !src->to_move || !src->leaf_depth || !src->leaf_slot || !src->n_leaf) return AZK_ERR_ARG;
    if (src->n_games < 1 || src->rows * src->cols != src->rc || src->flag_bytes < src->n_games) return AZK_ERR_ARG;
    // the leaf ranks travel as 16-bit per-class counters with 0xffff = "no game" (and 8-bit per-thread run counters): more pending-leaf
    // slots than this would wrap them silently - refuse, the caller keeps azk_nn_embed_pool_leaves / azk_step_gather for such engines
    if (src->n_games > AZK_EMBED_POOL_COMPACT_MAX_SLOTS) return AZK_ERR_ARG;
    return embed_pool_c_impl(nullptr, 0, src, consts, z_out_bf16_dev, src->n_games, src->planes, src->rows, src->cols, nullptr, sched_dev, stream);
}

// =====================================================================================================
// k_embed_fold: embedding + cls pooling without the token rows (include/azk.h azk_nn_embed_fold; pvnet.PolicyValueNet.fold_u).
//   x_t = Wc p_t + cpos_t with p_t the token's 0/1 patch, so LayerNorm1's variance is a quadratic form of <= 64 bits, the cls scores
//   are linear in them, and the value-projected pooled row u_h = (1/L_h) sum_t a_t[h] (M_h p_t + D_t[h]) is linear in x_t: what a
//   board contributes is, per head, one weight per token, 1/L, and the pooled patch sum_t a_t p_t / L - 384 bf16 per head, which the
//   tail's first GEMM multiplies with [D_t; U_all; M_h].  No conv, no D-wide normalisation, no gather of D-wide rows.
//   Board queue, leaf ranks, board bits, patch bits and the compaction of the stone-touched ("dirty") tokens are k_embed_pool_c's.
//   Tile loop: a WAVE owns 16-token tiles (tile = wave, wave + 4, ...), nothing between the waves until the board's sums meet:
//     Y = P G (16 x 64, fp16 hi + lo terms: exact products with the 0/1 patch), E = P S (scores), both v_mfma_f32_16x16x32_f16;
//     var_t = (sum_k p_tk (Y_tk + u2_tk) + n_t) / D from the accumulator layout (DPP row sum); w = exp(rstd (E + sc) - ref);
//     a = w rstd; b = a - aconst to LDS; pooled patch: Pw[h][k] += a_t[h] p_tk on v_mfma_f32_16x16x32_bf16 with k-slots 0..3 of a
//     lane group = its four tokens' a as bf16 hi, slots 4..7 = the bf16 remainder (B: the patch bits twice).
//   Board end: L and Pw of the four waves meet in LDS, every thread scales ITS token's eight weights by 1 / L into the output image
//   (LDS, bf16), the pooled patch and the three 1/L slots follow, the image leaves with 16-byte stores.
// =====================================================================================================
namespace {
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

struct EmbedFoldArgs {
    const void *boards;
    int boards_f32;
    const uint4 *gfrag;            // [2 (hi, lo)][4][2][64]
    const uint4 *efrag;            // [2][2][64]
    const float *u2T, *scoreT, *wcT, *lall, *sref, *inv_scales;
    void *out;                     // [n][NH][FOLD_ROW] bf16 (EX: float32)
    const int *count;
    int *sched;
    unsigned long long *wstats;
    long long *dbg;                // debug only (AZK_EP_STAMPS build + AZK_EMBED_POOL_STAMPS): [8] cycle sums per phase, wave 0 of every workgroup
    int n, R, Cc, T;
    float eps;
    azk_leaf_source src;
};

__device__ __forceinline__ unsigned bf16_rne(float v) {
    const unsigned u = __float_as_uint(v);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

constexpr int FOLD_ROW = AZK_EMBED_FOLD_ROW;
constexpr int FOLD_MAX_SLOTS = AZK_EMBED_FOLD_MAX_SLOTS;

// exp(x) for x <= ~80 with float32 accuracy: x log2(e) carried as hi + lo, v_exp_f32 on hi, first-order correction for lo (as azk_nnx.hip)
__device__ __forceinline__ float fold_exp_acc(float x) {
    const float L2E_HI = 1.44269502162933349609375f, L2E_LO = 1.92596299112661746e-8f;
    const float hi = x * L2E_HI;
    const float lo = __builtin_fmaf(x, L2E_HI, -hi) + x * L2E_LO;
    const float r = __builtin_amdgcn_exp2f(hi);
    return __builtin_fmaf(r, lo * 0.693147180559945309f, r);
}

// EX: the float32-accurate form (azk_nnx_embed_fold, the fp32 line): correctly rounded rsqrt, exp with an extended-precision argument, the
// pooled patch on v_mfma_f32_16x16x4_f32 (float32 weights against the 0 / 1 patch: exact products), float32 rows out (1 / L in one slot).
template <int NC, int KSZ, int NH, bool SRC, bool EX>
__global__ __launch_bounds__(256, 2) void k_embed_fold(EmbedFoldArgs a) {
    static_assert(NC * KSZ * KSZ <= 64, "the patch is one 64-bit word");
    constexpr int D = 512;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4 *alut = (uint4 *)smem;                                  // [256] A fragment (fp16 0 / 1) of 8 patch bits
    const int Tp16 = ((a.T + 15) >> 4) << 4;
    uint2 *pbits = (uint2 *)(alut + 256);                         // [Tp16] patch bits of the compacted dirty tokens
    int *dlist = (int *)(pbits + Tp16);                           // [Tp16] their token indices (null token = T past the end)
    int *scan = dlist + Tp16;                                     // [4 dirty counts per wave at 4..7], [10] next ticket, [16..31] class totals of the rank scan
    float *lred = (float *)(scan + 32);                           // [4 waves][8 heads]
    float *lall_s = lred + 32;                                    // [8] l_all
    float *bw = lall_s + 8;                                       // [Tp16][8]  a - aconst per dirty token and head
    float *pwred = bw + Tp16 * 8;                                 // [4 waves][8 heads][64]
    unsigned short *gor = (unsigned short *)(pwred + 4 * 8 * 64); // SRC: [n_games] game of rank r (= of the launch's r-th board)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    int nvalid;
    // the quadratic form's and the score columns' B fragments live in registers for the whole launch (80 VGPRs, 20 KB per wave once).
    // Measured against keeping them in LDS (154 VGPRs: three workgroups per CU, or two tiles per wave at a time): 86.5 vs 92.4 / 90.1 ms
    // per move - every MFMA then waits for its ds_read.
    uint4 gfr[16], efr[4];
#pragma unroll
    for (int i = 0; i < 16; i++) gfr[i] = a.gfrag[i * 64 + lane];
#pragma unroll
    for (int i = 0; i < 4; i++) efr[i] = a.efrag[i * 64 + lane];
    if (SRC) {
        // A non-zero leaf flag is 1 + the leaf's cost class (0..7, by stone count).  Board r of the launch is the r-th flagged game in
        // the order (class descending, game ascending): the stone-heavy boards are handed out first, the light ones fill the gaps at
        // the end.  Every workgroup derives the same ranks - per-class counts of its threads' games (thread t owns games [t per, (t+1)
        // per)), an exclusive scan over the 256 threads with the eight 16-bit counters packed in two 64-bit words - and keeps the
        // inverse (game of rank r) in LDS: a board's game is then one LDS read, whoever asks.
        const int my_per = ((((a.src.n_games + 255) >> 8) + 7) >> 3) << 3, my_lo = tid * my_per;
        unsigned long long c_lo = 0ull, c_hi = 0ull;              // classes 0-3 / 4-7, 16 bits each
        for (int w = 0; w < my_per; w += 8)
            if (my_lo + w < a.src.flag_bytes) {
                const unsigned long long f = *(const unsigned long long *)(a.src.leaf_flag + my_lo + w);
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const unsigned c = (unsigned)((f >> (8 * q)) & 0xffull);
                    if (c) { if (c <= 4) c_lo += 1ull << (16 * (c - 1)); else c_hi += 1ull << (16 * (c - 5)); }
                }
            }
        // inclusive wave scan of the eight packed 16-bit counters on DPP (row_shr 1, 2, 4, 8, then row_bcast 15 / 31: six dependent
        // v_add per dword instead of six ds_bpermute round trips per 64-bit word - ~2 k cycles at the head of every launch).  The fields
        // never carry into each other (a count is at most the slot count, < 65536), so the four dwords scan independently.
        unsigned long long i_lo, i_hi;
        {
            int w4[4] = {(int)(unsigned)c_lo, (int)(unsigned)(c_lo >> 32), (int)(unsigned)c_hi, (int)(unsigned)(c_hi >> 32)};
#pragma unroll
            for (int q = 0; q < 4; q++) {
                int v = w4[q];
                v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);
                v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);
                v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);
                v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);
                v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);
                v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);
                w4[q] = v;
            }
            i_lo = ((unsigned long long)(unsigned)w4[1] << 32) | (unsigned)w4[0];
            i_hi = ((unsigned long long)(unsigned)w4[3] << 32) | (unsigned)w4[2];
        }
        unsigned long long *wtot = (unsigned long long *)(scan + 16);           // [4 waves][2]
        if (lane == 63) { wtot[2 * wave] = i_lo; wtot[2 * wave + 1] = i_hi; }
        __syncthreads();
        unsigned long long b_lo = 0ull, b_hi = 0ull, t_lo = 0ull, t_hi = 0ull;
        for (int w = 0; w < 4; w++) {
            if (w < wave) { b_lo += wtot[2 * w]; b_hi += wtot[2 * w + 1]; }
            t_lo += wtot[2 * w]; t_hi += wtot[2 * w + 1];
        }
        const unsigned long long e_lo = b_lo + i_lo - c_lo, e_hi = b_hi + i_hi - c_hi;   // exclusive prefix over lower threads, per class
        unsigned long long cb_lo = 0ull, cb_hi = 0ull;            // rank of this thread's next game of each class, 16 bits each (classes 0-3 / 4-7)
        unsigned start = 0;
#pragma unroll
        for (int c = 7; c >= 0; c--) {                            // class 7 (most stones) first
            const unsigned long long cb = (unsigned long long)(start + (unsigned)(((c < 4 ? e_lo : e_hi) >> (16 * (c & 3))) & 0xffffull)) << (16 * (c & 3));
            if (c < 4) cb_lo |= cb; else cb_hi |= cb;
            start += (unsigned)(((c < 4 ? t_lo : t_hi) >> (16 * (c & 3))) & 0xffffull);
        }
        nvalid = (int)start;
        for (int w = 0; w < my_per; w += 8)
            if (my_lo + w < a.src.flag_bytes) {
                const unsigned long long f = *(const unsigned long long *)(a.src.leaf_flag + my_lo + w);
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const unsigned c = (unsigned)((f >> (8 * q)) & 0xffull);
                    const bool lo = c <= 4;
                    const int sh = 16 * ((c - 1) & 3);
                    if (c) gor[(unsigned)(((lo ? cb_lo : cb_hi) >> sh) & 0xffffull)] = (unsigned short)(my_lo + w + q);
                    const unsigned long long inc = c ? 1ull << sh : 0ull;
                    cb_lo += lo ? inc : 0ull; cb_hi += lo ? 0ull : inc;
                }
            }
        if (blockIdx.x == 0 && tid == 0) { *a.src.n_leaf = nvalid; if (a.src.cache_stamp) *a.src.cache_stamp += 1u; }
    } else {
        nvalid = a.count ? min(a.n, *a.count) : a.n;
    }
    int board = blockIdx.x;
    int ws_boards = 0, ws_tiles = 0;
#ifdef AZK_EP_STAMPS
    const bool stamp = a.dbg != nullptr && tid == 0;
    long long tp = stamp ? clock64() : 0, tacc[5] = {0, 0, 0, 0, 0};
#define AZK_FSTAMP(i) do { if (stamp) { const long long tn_ = clock64(); tacc[i] += tn_ - tp; tp = tn_; } } while (0)
#else
#define AZK_FSTAMP(i) do { } while (0)
#endif
    if (board < nvalid) {
    {
        unsigned r[4];
#pragma unroll
        for (int i = 0; i < 4; i++) r[i] = (((tid >> (2 * i)) & 1) ? 0x3C00u : 0u) | (((tid >> (2 * i + 1)) & 1) ? 0x3C000000u : 0u);
        alut[tid] = make_uint4(r[0], r[1], r[2], r[3]);
    }
    constexpr int ksz = KSZ, kk = KSZ * KSZ, pad = KSZ / 2;
    const int RC = a.R * a.Cc, T = a.T, ncell = NC * RC;
    const float sref = a.sref[l15];
    if (tid < 8) lall_s[tid] = a.lall[tid];
    const float invD = 1.0f / (float)D, ginv = a.inv_scales[0], einv = a.inv_scales[1];
    __syncthreads();
    AZK_FSTAMP(0);                                                // launch prologue: ranks, fragments staged

    // Boards of up to 512 plane cells: every load of a board is issued at once (one round trip), and the NEXT board's loads are issued
    // as soon as its ticket is known - behind the tiles' barrier, under the output phase - so a board starts with its cells on hand.
    constexpr int NQ = 8;
    const bool fast = ncell <= 64 * NQ;
    int codeN[NQ], tmN = 0, ldN = 0, game = 0;
    auto load_cells = [&](int g) {                                // SRC: cell codes (one byte per cell) + the two words that give the side to move
        const auto *cells = a.src.leaf_cells + (size_t)g * a.src.rc_pad;           // uniform base + 32-bit lane offsets
        int lv = tid;                                             // (opaque: the eight offsets are recomputed per call, not kept across the board loop)
        asm volatile("" : "+v"(lv));
        lv &= 63;
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const int e = min(q * 64 + lv, ncell - 1);
            codeN[q] = cells[(unsigned)(e - ((e >= RC) + (e >= 2 * RC)) * RC)];
        }
        tmN = a.src.to_move[g]; ldN = a.src.leaf_depth[g];
    };
    if (SRC) { game = gor[board]; if (fast) load_cells(game); }

    while (board < nvalid) {
        // Everything below that depends only on the thread index is recomputed per board from an opaque copy of the index: hoisted out of
        // the board loop these values live through the tile loop and spill.
        int tv = tid;
        asm volatile("" : "+v"(tv));
        const int lane_b = tv & 63;
        const int tj = tv - 1, tr = tj / a.Cc, tc = tj - tr * a.Cc;
        const bool tlive = tv >= 1 && tv < T;
        int player = 0;
        if (SRC && tid == 0) a.src.leaf_slot[game] = board;       // the slot the next expansion reads this game's outputs from
        unsigned wbits = 0;                             // lane i holds bits [32 (i-1), 32 i) of the board bit string (lane 0: zeros)
        if (fast) {
            bool on[NQ];
            if (SRC) {
                // canonical planes from the cell codes (gomoku.py:34-40; 3-plane: mcts.py:126-137)
                player = (tmN + ldN) & 1;                         // node.currentPlayer at the leaf
#pragma unroll
                for (int q = 0; q < NQ; q++) {
                    const int e = min(q * 64 + lane_b, ncell - 1), chq = (e >= RC) + (e >= 2 * RC);
                    on[q] = q * 64 + lane_b < ncell && (chq == 2 ? player != 0 : ((codeN[q] >> (chq ^ player)) & 1) != 0);
                }
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
        AZK_FSTAMP(1);                                            // board loaded, bit string built
        // ---- patch bits of this thread's token; dirty = some stone in the patch ----
        // Row words first: lane L < NC R holds plane L / R, row L % R of the board with two zero bits on either side (two cross-lane reads
        // of the bit string + a funnel shift, once); a token then takes its KSZ bits of each of its NC KSZ rows with ONE cross-lane read
        // and a shift - no column mask, the margins are zero.
        unsigned long long plo = 0;
        {
            const int chL = lane_b / a.R, rL = lane_b - chL * a.R;
            const int offL = 32 + chL * RC + rL * a.Cc - 2;
            const unsigned loL = __shfl(wbits, offL >> 5), hiL = __shfl(wbits, (offL >> 5) + 1);
            const unsigned roww = lane_b < NC * a.R ? (__funnelshift_r(loL, hiL, offL & 31) & (((1u << a.Cc) - 1u) << 2)) : 0u;
            unsigned rw[NC * KSZ];
#pragma unroll
            for (int ch = 0; ch < NC; ch++)
#pragma unroll
                for (int ky = 0; ky < KSZ; ky++) {
                    const int rr = tr + ky - pad;
                    rw[ch * KSZ + ky] = __shfl(roww, ch * a.R + (rr < 0 ? 0 : (rr >= a.R ? a.R - 1 : rr)));
                }
#pragma unroll
            for (int ch = 0; ch < NC; ch++)
#pragma unroll
                for (int ky = 0; ky < KSZ; ky++) {
                    const int rr = tr + ky - pad;
                    unsigned bits = (rw[ch * KSZ + ky] >> (tc + 2 - pad)) & ((1u << KSZ) - 1u);
                    if (!tlive || rr < 0 || rr >= a.R) bits = 0;
                    plo |= (unsigned long long)bits << (ch * kk + ky * ksz);
                }
        }
        const bool dirty = plo != 0ull;
        const unsigned long long dm = __ballot(dirty);
        if (lane == 0) scan[4 + wave] = __popcll(dm);
        __syncthreads();                                  // (also: every wave is done with the previous board's lists and sums)
        int dpos = __popcll(dm & ((1ull << lane) - 1ull));
        for (int w = 0; w < wave; w++) dpos += scan[4 + w];
        const int nd = scan[4] + scan[5] + scan[6] + scan[7];
        const int ntile = (nd + 15) >> 4;
        if (dirty) {
            dlist[dpos] = tid;
            pbits[dpos] = make_uint2((unsigned)plo, (unsigned)(plo >> 32));
        }
        if (tid < 16 && nd + tid < ntile * 16) { dlist[nd + tid] = T; pbits[nd + tid] = make_uint2(0u, 0u); }   // null tokens fill the last tile
        __syncthreads();
        ws_boards += 1; ws_tiles += ntile;
        AZK_FSTAMP(2);                                            // patch bits + compaction

        // ---- the wave's tiles: wave, wave + 4, ... ----
        float L = 0.f;                                    // per head (lane&15 < NH): this lane group's share of sum (w - wconst)
        f32x4 Pw[4];
#pragma unroll
        for (int q = 0; q < 4; q++) Pw[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        // the per-token constants are gathered (L2) one tile ahead: the round trip runs under the previous tile's arithmetic
        f32x4 utn[4];
        float scnn[4], wcnn[4], ntn[4], rcn[4];
        auto gather = [&](int t) {
            const int4 tk = *(const int4 *)(dlist + 16 * t + 4 * l4);
            const int tks[4] = {tk.x, tk.y, tk.z, tk.w};
#pragma unroll
            for (int r = 0; r < 4; r++) {
                // (the cross term u2_t . p_t is summed over the lane group like the quadratic form, but with ITS OWN column split: lane
                //  l15 takes columns 4 l15 .. 4 l15 + 3 - one 16-byte load per token instead of four 4-byte ones)
                const unsigned ou = ((unsigned)tks[r] * 64u + 4u * (unsigned)l15) * 4u, os = ((unsigned)tks[r] * 16u + (unsigned)l15) * 4u;
                utn[r] = *(const f32x4 *)((const char *)a.u2T + ou);
                scnn[r] = *(const float *)((const char *)a.scoreT + os);
                wcnn[r] = *(const float *)((const char *)a.wcT + os);
                // the token's n_t and constant rstd sit in column 15 of the same rows: loaded by every lane of the group (one line, a
                // tile ahead) instead of being broadcast from lane 15 through two ds_bpermute round trips on the tile's critical path
                ntn[r] = *(const float *)((const char *)a.scoreT + (unsigned)tks[r] * 64u + 60u);
                rcn[r] = *(const float *)((const char *)a.wcT + (unsigned)tks[r] * 64u + 60u);
            }
        };
        if (wave < ntile) gather(wave);
        for (int tile = wave; tile < ntile; tile += 4) {
            const int base = 16 * tile;
            // patch bits of this lane group's four tokens (rows 4 l4 + r)
            const int4 pq0 = *(const int4 *)(pbits + base + 4 * l4), pq1 = *(const int4 *)(pbits + base + 4 * l4 + 2);
            const unsigned prx[4] = {(unsigned)pq0.x, (unsigned)pq0.z, (unsigned)pq1.x, (unsigned)pq1.z};
            const unsigned pry[4] = {(unsigned)pq0.y, (unsigned)pq0.w, (unsigned)pq1.y, (unsigned)pq1.w};
            // the gathered rows are consumed at once (their registers take the next tile's): this lane's share of u2_t . p_t
            float cross[4], scn[4], wcn[4], ntv[4], rcv[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const unsigned nib = (l15 < 8 ? prx[r] : pry[r]) >> (4 * (l15 & 7));                   // patch bits 4 l15 .. 4 l15 + 3
                float c = 0.f;
#pragma unroll
                for (int cidx = 0; cidx < 4; cidx++) c = fmaf((float)((nib >> cidx) & 1u), utn[r][cidx], c);
                cross[r] = c; scn[r] = scnn[r]; wcn[r] = wcnn[r]; ntv[r] = ntn[r]; rcv[r] = rcn[r];
            }
            __builtin_amdgcn_sched_barrier(0);
            gather(tile + 4 < ntile ? tile + 4 : tile);          // (the last tile refetches itself: no branch around loads)
            __builtin_amdgcn_sched_barrier(0);
            const uint2 pa = pbits[base + l15];
            union { uint4 u; f16x8 v; } af[2];
            af[0].u = alut[(pa.x >> (8 * l4)) & 0xffu];
            af[1].u = alut[(pa.y >> (8 * l4)) & 0xffu];
            f32x4 Y[4], E = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; q++) Y[q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++) {
#pragma unroll
                for (int hl = 0; hl < 2; hl++) {
                    union { uint4 u; f16x8 v; } b;
                    b.u = efr[hl * 2 + s2];
                    E = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[s2].v, b.v, E, 0, 0, 0);
                }
#pragma unroll
                for (int q = 0; q < 4; q++)
#pragma unroll
                    for (int hl = 0; hl < 2; hl++) {
                        union { uint4 u; f16x8 v; } b;
                        b.u = gfr[(hl * 4 + q) * 2 + s2];
                        Y[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[s2].v, b.v, Y[q], 0, 0, 0);
                    }
            }
            // ---- per token (row 4 l4 + r): variance from the quadratic form, scores, weights ----
            // patch bit of token r at column 16 q + l15 as a float (0 / 1): multiplies the quadratic form's column and, as its upper half, IS the
            // bf16 B operand of the pooled patch (one v_perm per pair of tokens instead of compare / select chains)
            float bitf[4][4];
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int q = 0; q < 4; q++) bitf[r][q] = (float)(((q < 2 ? prx[r] : pry[r]) >> (16 * (q & 1) + l15)) & 1u);
            float av[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                float qd = 0.f;
#pragma unroll
                for (int q = 0; q < 4; q++) qd = fmaf(bitf[r][q], Y[q][r], qd);
                qd = fmaf(qd, ginv, cross[r]);
                qd = row16_sum(qd);
                const float e = fmaf(E[r], einv, scn[r]);                        // head lanes: the raw score
                const float var = fmaxf((qd + ntv[r]) * invD, 0.f) + a.eps;
                const float rstd = EX ? 1.0f / sqrtf(var) : __builtin_amdgcn_rsqf(var);
                const float w = EX ? fold_exp_acc(rstd * e - sref) : __expf(rstd * e - sref);   // (0 beyond the heads: their reference is +1e30)
                av[r] = w * rstd;
                L += w - wcn[r];
                if (l15 < NH) bw[(base + 4 * l4 + r) * 8 + l15] = av[r] - wcn[r] * rcv[r];
            }
            // ---- pooled patch: Pw[h][k] += a_t[h] p_tk ----
            if (EX) {
                // float32: one v_mfma_f32_16x16x4_f32 per token of the lane group and column tile (A: a of token 4 l4 + r for head l15,
                // B: that token's patch bit at column 16 q + l15; the k index of both is the lane group)
#pragma unroll
                for (int r = 0; r < 4; r++)
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        Pw[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r], bitf[r][q], Pw[q], 0, 0, 0);
                    }
            } else {
                // bf16: a as hi + remainder in the eight k-slots of the lane group (B: the patch bits twice)
                union { bf16x8 v; s16x4 h[2]; } wa;
                wa.h[0] = pack4_bf16(f32x2{av[0], av[1]}, f32x2{av[2], av[3]});
                {
                    const u32x2 hh = __builtin_bit_cast(u32x2, wa.h[0]);
                    const float r0 = av[0] - __uint_as_float(hh[0] << 16), r1 = av[1] - __uint_as_float(hh[0] & 0xffff0000u);
                    const float r2 = av[2] - __uint_as_float(hh[1] << 16), r3 = av[3] - __uint_as_float(hh[1] & 0xffff0000u);
                    wa.h[1] = pack4_bf16(f32x2{r0, r1}, f32x2{r2, r3});
                }
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    // (bf16 of a float is its upper half: 1.0f -> 0x3F80)
                    const unsigned b01 = __builtin_amdgcn_perm(__float_as_uint(bitf[1][q]), __float_as_uint(bitf[0][q]), 0x07060302u);
                    const unsigned b23 = __builtin_amdgcn_perm(__float_as_uint(bitf[3][q]), __float_as_uint(bitf[2][q]), 0x07060302u);
                    union { uint4 u; bf16x8 v; } pb;
                    pb.u = make_uint4(b01, b23, b01, b23);
                    Pw[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa.v, pb.v, Pw[q], 0, 0, 0);
                }
            }
        }
        // next board: one ticket per board, drawn by the wave with the fewest tiles behind its last one (a returning atomic is waited
        // for where it is issued: in wave 0 it sat on the board's critical path)
        if (tid == 192) scan[10] = atomicAdd(a.sched, 1);
        AZK_FSTAMP(3);                                            // wave 0's tiles
        // ---- the waves' sums meet ----
        {
            float Lw = L + __shfl_xor(L, 16);
            Lw += __shfl_xor(Lw, 32);
            if (lane < 8) lred[wave * 8 + lane] = lane < NH ? Lw : 0.f;
            if (l4 < 2) {
#pragma unroll
                for (int q = 0; q < 4; q++)
#pragma unroll
                    for (int j = 0; j < 4; j++) pwred[(wave * 8 + 4 * l4 + j) * 64 + 16 * q + l15] = Pw[q][j];
            }
        }
        __syncthreads();
        // the next board is known to every thread now: its loads go out under this board's output phase
        const int nxt = scan[10], nboard = (int)gridDim.x + nxt;
        int ngame = 0;
        if (SRC && nboard < nvalid) { ngame = gor[nboard]; if (fast) load_cells(ngame); }
        if (tid == 0 && nxt == nvalid - 1) a.sched[0] = 0;        // exactly nvalid tickets are drawn per launch: the last one leaves the queue zero
        // ---- output rows, straight to memory: [head][0, T) token weights / L, [T, T+3) 1 / L (hi, lo, hi), [256, 320) pooled patch / L ----
        {
            float inv[NH];
            {
                const f32x4 *lr = (const f32x4 *)lred;
                f32x4 s0 = (lr[0] + lr[2]) + (lr[4] + lr[6]), s1 = (lr[1] + lr[3]) + (lr[5] + lr[7]);
                const f32x4 *la = (const f32x4 *)lall_s;
                s0 += la[0]; s1 += la[1];
#pragma unroll
                for (int h = 0; h < NH; h++) inv[h] = EX ? 1.0f / (h < 4 ? s0[h & 3] : s1[h & 3]) : __builtin_amdgcn_rcpf(h < 4 ? s0[h & 3] : s1[h & 3]);   // (bf16 rows: 1 ulp is below their rounding)
            }
            const bool isL = tid >= T && tid < T + 3;
            f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f};
            if (dirty) { b0 = *(const f32x4 *)(bw + dpos * 8); b1 = *(const f32x4 *)(bw + dpos * 8 + 4); }
            float tinv[NH];                                        // the scale of this thread's token entries
#pragma unroll
            for (int h = 0; h < NH; h++) tinv[h] = inv[h];
            if (isL) {                                             // three threads: 1 / L (bf16 rows: as hi, remainder, hi; float32 rows: value, 0, 0)
#pragma unroll
                for (int h = 0; h < NH; h++) {
                    const unsigned hi = bf16_rne(inv[h]);
                    float v = tid == T + 1 ? inv[h] - __uint_as_float(hi << 16) : __uint_as_float(hi << 16);
                    if (EX) v = tid == T ? inv[h] : 0.f;
                    if (h < 4) b0[h & 3] = v; else b1[h & 3] = v;
                    tinv[h] = 1.0f;
                }
            }
            if (EX) {
                float *of = (float *)a.out + (size_t)board * NH * FOLD_ROW;
#pragma unroll
                for (int h = 0; h < NH; h++) of[h * FOLD_ROW + tid] = (h < 4 ? b0[h & 3] : b1[h & 3]) * tinv[h];
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    const int e = tid + 256 * i, h = e >> 6, k = e & 63;
                    if (h < NH) {
                        of[h * FOLD_ROW + 256 + k] = ((pwred[(0 * 8 + h) * 64 + k] + pwred[(1 * 8 + h) * 64 + k]) + (pwred[(2 * 8 + h) * 64 + k] + pwred[(3 * 8 + h) * 64 + k])) * inv[h];
                        of[h * FOLD_ROW + 320 + k] = 0.f;
                    }
                }
            } else {
                unsigned short *ob = (unsigned short *)a.out + (size_t)board * NH * FOLD_ROW;
                const u32x2 p0 = __builtin_bit_cast(u32x2, pack4_bf16(f32x2{b0[0] * tinv[0], b0[1] * tinv[1]}, f32x2{b0[2] * tinv[2], b0[3] * tinv[3]}));
                ob[0 * FOLD_ROW + tid] = (unsigned short)p0[0]; ob[1 * FOLD_ROW + tid] = (unsigned short)(p0[0] >> 16);
                ob[2 * FOLD_ROW + tid] = (unsigned short)p0[1]; ob[3 * FOLD_ROW + tid] = (unsigned short)(p0[1] >> 16);
                if (NH > 4) {
                    const u32x2 p1 = __builtin_bit_cast(u32x2, pack4_bf16(f32x2{b1[0] * tinv[4 % NH], b1[1] * tinv[5 % NH]}, f32x2{b1[2] * tinv[6 % NH], b1[3] * tinv[7 % NH]}));
                    ob[4 * FOLD_ROW + tid] = (unsigned short)p1[0]; ob[5 * FOLD_ROW + tid] = (unsigned short)(p1[0] >> 16);
                    ob[6 * FOLD_ROW + tid] = (unsigned short)p1[1]; ob[7 * FOLD_ROW + tid] = (unsigned short)(p1[1] >> 16);
                }
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    const int e = tid + 256 * i, h = e >> 6, k = e & 63;                      // (a wave writes 64 consecutive entries)
                    if (h < NH) {
                        const float v = ((pwred[(0 * 8 + h) * 64 + k] + pwred[(1 * 8 + h) * 64 + k]) + (pwred[(2 * 8 + h) * 64 + k] + pwred[(3 * 8 + h) * 64 + k])) * inv[h];
                        ob[h * FOLD_ROW + 256 + k] = (unsigned short)bf16_rne(v);
                        ob[h * FOLD_ROW + 320 + k] = 0;
                    }
                }
            }
        }
        board = nboard; game = ngame;
        AZK_FSTAMP(4);                                            // sums, output rows, next board's loads issued
    }
    }
    if (a.wstats != nullptr && tid == 0 && ws_boards) { atomicAdd(a.wstats, (unsigned long long)ws_boards); atomicAdd(a.wstats + 1, (unsigned long long)ws_tiles); }
#undef AZK_FSTAMP
#ifdef AZK_EP_STAMPS
    if (stamp) {
        atomicMax((unsigned long long *)a.dbg + 5, (unsigned long long)(tacc[0] + tacc[1] + tacc[2] + tacc[3] + tacc[4]));   // the busiest workgroup of any launch
        for (int i = 0; i < 5; i++) atomicAdd((unsigned long long *)a.dbg + i, (unsigned long long)tacc[i]);
        atomicAdd((unsigned long long *)a.dbg + 6, (unsigned long long)ws_tiles);
        atomicAdd((unsigned long long *)a.dbg + 7, (unsigned long long)ws_boards);
    }
#endif
}

int g_fold_grid = 0;

template <int NC, int KSZ, int NH, bool SRC, bool EX>
int launch_embed_fold(const EmbedFoldArgs &a, hipStream_t st) {
    const int tp16 = ((a.T + 15) / 16) * 16;
    const int lds = 256 * 16 + tp16 * 8 + tp16 * 4 + 128 + 128 + 32 + tp16 * 32 + 4 * 8 * 64 * 4 + (SRC ? ((a.src.n_games + 7) / 8) * 16 : 0);   // 29 KB at 2 048 games
    if (azk_set_max_lds((const void *)k_embed_fold<NC, KSZ, NH, SRC, EX>, lds + 2 * FOLD_MAX_SLOTS) != hipSuccess) return AZK_ERR_HIP;
    // two resident workgroups per CU by default; each pulls boards until the queue is dry.  azk_nn_embed_fold_grid(n): a caller that steps
    // several game groups on separate streams caps the grid (one workgroup per CU leaves the register file room for another group's
    // tree waves)
    const int cap = g_fold_grid > 0 ? g_fold_grid : 512;
    const int blocks = a.n < cap ? a.n : cap;
    k_embed_fold<NC, KSZ, NH, SRC, EX><<<blocks, 256, lds, st>>>(a);
    return hipGetLastError() == hipSuccess ? AZK_OK : AZK_ERR_HIP;
}
}  // namespace

extern "C" int32_t azk_nn_embed_fold_grid(int32_t max_workgroups) {
    if (max_workgroups < 0) return AZK_ERR_ARG;
    g_fold_grid = max_workgroups;                                   // 0: the default (512)
    return AZK_OK;
}

static int32_t embed_fold_impl(const void *boards_dev, int32_t boards_are_f32, const azk_leaf_source *src, const azk_embed_fold_consts *k,
                               void *rows_out, int32_t n, int32_t channels, int32_t rows, int32_t cols, const int32_t *n_valid_dev,
                               int32_t *sched_dev, void *stream, bool exact = false) {
    if ((!boards_dev && !src) || !k || !rows_out || !sched_dev) return AZK_ERR_ARG;
    if (!k->g_frag || !k->e_frag || !k->u2_tok || !k->score_tok || !k->wconst_tok || !k->l_all || !k->score_ref || !k->inv_scales) return AZK_ERR_ARG;
    const int ksize = k->ksize;
    if (n < 0 || channels < 1 || rows < 1 || cols < 1 || ksize < 1 || (ksize & 1) == 0 || channels * ksize * ksize > 64) return AZK_ERR_ARG;
    if (channels * rows * cols > 62 * 32 || k->embed_dim != 512) return AZK_ERR_ARG;
    if (rows * cols + 1 + 3 > 256) return AZK_ERR_ARG;             // one thread per token, three more for 1 / L
    if (channels * rows > 64 || cols > 28 || ksize > 5) return AZK_ERR_ARG;   // one lane per (plane, row) word: the row + two margin bits a side in 32 bits
    if (k->num_heads != 8 && k->num_heads != 4) return AZK_ERR_ARG;
    if (n == 0) return AZK_OK;
    EmbedFoldArgs a;
    memset(&a, 0, sizeof a);
    a.boards = boards_dev; a.boards_f32 = boards_are_f32; a.gfrag = (const uint4 *)k->g_frag; a.efrag = (const uint4 *)k->e_frag;
    a.u2T = k->u2_tok; a.scoreT = k->score_tok; a.wcT = k->wconst_tok; a.lall = k->l_all; a.sref = k->score_ref; a.inv_scales = k->inv_scales;
    a.out = rows_out; a.count = n_valid_dev; a.sched = sched_dev; a.wstats = (unsigned long long *)k->work_stats;
    a.n = n; a.R = rows; a.Cc = cols; a.T = rows * cols + 1; a.eps = k->ln_eps;
    if (src) a.src = *src;
    {
        static long long *dbg_buf = nullptr;
        const char *ds = getenv("AZK_EMBED_POOL_STAMPS");
        if (ds && atoi(ds)) {
            if (!dbg_buf && (hipMalloc((void **)&dbg_buf, 64) != hipSuccess || hipMemset(dbg_buf, 0, 64) != hipSuccess)) return AZK_ERR_HIP;
            a.dbg = dbg_buf;
            if (atoi(ds) == 2) {          // print-and-reset request
                long long h[8];
                if (hipMemcpy(h, dbg_buf, 64, hipMemcpyDeviceToHost) != hipSuccess) return AZK_ERR_HIP;
                const double nb = (double)(h[7] ? h[7] : 1);
                fprintf(stderr, "[embed_fold stamps] busiest workgroup %lld cycles | boards %lld tiles %lld | launch prologue (total) %lld | cycles per board: "
                        "load %.0f, patch+compact %.0f, wave 0 tiles %.0f, sums+output %.0f\n", h[5], h[7], h[6], h[0], h[1] / nb, h[2] / nb, h[3] / nb, h[4] / nb);
                if (hipMemset(dbg_buf, 0, 64) != hipSuccess) return AZK_ERR_HIP;
            }
        }
    }
    hipStream_t st = (hipStream_t)stream;
    const int nh = k->num_heads;
#define CASE(NC_, KSZ_, NH_) if (channels == NC_ && ksize == KSZ_ && nh == NH_) \
        return exact ? (src ? launch_embed_fold<NC_, KSZ_, NH_, true, true>(a, st) : launch_embed_fold<NC_, KSZ_, NH_, false, true>(a, st)) \
                     : (src ? launch_embed_fold<NC_, KSZ_, NH_, true, false>(a, st) : launch_embed_fold<NC_, KSZ_, NH_, false, false>(a, st))
    CASE(2, 5, 8); CASE(2, 5, 4); CASE(2, 3, 8); CASE(2, 3, 4); CASE(3, 3, 8); CASE(3, 3, 4);
#undef CASE
    return AZK_ERR_ARG;
}

extern "C" int32_t azk_nn_embed_fold(const void *boards_dev, int32_t boards_are_f32, const azk_embed_fold_consts *consts, void *rows_out_bf16_dev,
                                     int32_t n, int32_t channels, int32_t rows, int32_t cols, const int32_t *n_valid_dev, int32_t *sched_dev,
                                     void *stream) {
    if (!boards_dev) return AZK_ERR_ARG;
    return embed_fold_impl(boards_dev, boards_are_f32, nullptr, consts, rows_out_bf16_dev, n, channels, rows, cols, n_valid_dev, sched_dev, stream);
}

static int32_t embed_fold_leaves_impl(const azk_leaf_source *src, const azk_embed_fold_consts *consts, void *rows_out, int32_t *sched_dev,
                                      void *stream, bool exact) {
    if (!src || !src->leaf_flag || !src->leaf_cells || !src->to_move || !src->leaf_depth || !src->leaf_slot || !src->n_leaf) return AZK_ERR_ARG;
    if (src->n_games < 1 || src->rows * src->cols != src->rc || src->flag_bytes < src->n_games) return AZK_ERR_ARG;
    if (src->n_games > AZK_EMBED_FOLD_MAX_SLOTS) return AZK_ERR_ARG;        // the rank -> game table lives in LDS (2 bytes per slot)
    return embed_fold_impl(nullptr, 0, src, consts, rows_out, src->n_games, src->planes, src->rows, src->cols, nullptr, sched_dev, stream, exact);
}

extern "C" int32_t azk_nn_embed_fold_leaves(const azk_leaf_source *src, const azk_embed_fold_consts *consts, void *rows_out_bf16_dev,
                                            int32_t *sched_dev, void *stream) {
    return embed_fold_leaves_impl(src, consts, rows_out_bf16_dev, sched_dev, stream, false);
}

// the float32-accurate form (the fp32 line, include/azk.h): float32 rows for azk_nnx_gemm_h's float32-A first link
extern "C" int32_t azk_nnx_embed_fold(const void *boards_dev, int32_t boards_are_f32, const azk_embed_fold_consts *consts, float *rows_out_f32_dev,
                                      int32_t n, int32_t channels, int32_t rows, int32_t cols, const int32_t *n_valid_dev, int32_t *sched_dev,
                                      void *stream) {
    if (!boards_dev) return AZK_ERR_ARG;
    return embed_fold_impl(boards_dev, boards_are_f32, nullptr, consts, rows_out_f32_dev, n, channels, rows, cols, n_valid_dev, sched_dev, stream, true);
}

extern "C" int32_t azk_nnx_embed_fold_leaves(const azk_leaf_source *src, const azk_embed_fold_consts *consts, float *rows_out_f32_dev,
                                             int32_t *sched_dev, void *stream) {
    return embed_fold_leaves_impl(src, consts, rows_out_f32_dev, sched_dev, stream, true);
}

// =====================================================================================================
// cls-row tail (nn.py:54-60, 78-83 for the one row the heads read): small-M GEMMs with a device-side row count.
//   k_gemm_rows: C[M][N] = A[M][K] (bf16, row-major) x W^T, W packed in MFMA B-fragment order (one 16-byte load per
//   fragment, 1 KB contiguous per wave); wave tile 64 rows x 64 columns (16 accumulators), no LDS: the four waves of a
//   workgroup take neighbouring column groups of the same rows, so their A fragments hit in L1.  The K dimension can be
//   split over `ksplit` waves; partial sums go to separate float32 planes that the next (row-wise) kernel adds up - no
//   atomics, deterministic.  Rows at or beyond *n_valid are never touched.
//   Packed weight: Wp[N/64][K/32][4][64 lanes][8] with element = W[64 g + 4 (lane&15) + c][32 s + 8 (lane>>4) + i], so a
//   lane's four accumulators of a row are four consecutive output columns (16-byte float / 8-byte bf16 stores).
// =====================================================================================================
namespace {

struct GemmArgs {
    const unsigned short *A;   // [M][lda] bf16
    int lda;
    const uint4 *Wp;           // packed weight
    int M, N, K, ksplit;
    const int *count;
    float *P;                  // mode 0: [ksplit][M][N] float32 partial sums
    const float *bias;         // modes 1, 2: [N]
    unsigned short *out;       // mode 1: [M][N] bf16 = gelu(A W^T + bias)
    const float *ln_w, *ln_b;  // mode 2: LayerNorm affine over the K input columns (applied to A on the fly)
    float ln_eps;
    float *logits, *values;    // mode 2: float32 [M][action_dim], [M] = tanh(column action_dim)
    int action_dim;
};

template <int MODE, int RT, int KU>   // RT = 16-row tiles per wave (wave tile 16 RT rows x 64 columns); KU = k-steps whose fragments
                                     // are all requested before the first MFMA of the batch (latency hiding)
__global__ __launch_bounds__(256, 2) void k_gemm_rows(GemmArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, l4 = lane >> 4;
    const int nvalid = a.count ? min(a.M, *a.count) : a.M;
    const int rtiles = (nvalid + 16 * RT - 1) / (16 * RT), ngroups = a.N >> 6, ksteps = a.K >> 5;
    const int nitems = rtiles * ngroups * a.ksplit;
    const int nks = ksteps / a.ksplit;
    for (int item = blockIdx.x * 4 + wave; item < nitems; item += gridDim.x * 4) {
        const int ng = item % ngroups, r = item / ngroups, rt = r % rtiles, s = r / rtiles;
        const int ks0 = s * nks;
        f32x4 acc[RT][4];
#pragma unroll
        for (int i = 0; i < RT; i++)
#pragma unroll
            for (int c = 0; c < 4; c++) acc[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        const unsigned short *ap[RT];
#pragma unroll
        for (int i = 0; i < RT; i++) {
            const int row = min(16 * RT * rt + 16 * i + l15, nvalid - 1);
            ap[i] = a.A + (size_t)row * a.lda + 32 * ks0 + 8 * l4;
        }
        const uint4 *bp = a.Wp + ((size_t)ng * ksteps + ks0) * 4 * 64 + lane;
        // MODE 2: A is LayerNorm(rows) - the wave reads whole rows (ksplit 1), so it takes the row statistics itself in a
        // first pass (fp32 sums over the lane's fragments, then across the four k-groups of a row)
        float mean[RT], rstd[RT];
        if (MODE == 2) {
            float s1[RT], s2[RT];
#pragma unroll
            for (int i = 0; i < RT; i++) { s1[i] = 0.f; s2[i] = 0.f; }
            for (int kb = 0; kb < nks; kb += KU) {
                uint4 raw[KU][RT];
#pragma unroll
                for (int u = 0; u < KU; u++)
#pragma unroll
                    for (int i = 0; i < RT; i++) raw[u][i] = *(const uint4 *)(ap[i] + 32 * (kb + u));
#pragma unroll
                for (int u = 0; u < KU; u++)
#pragma unroll
                    for (int i = 0; i < RT; i++) {
                        const unsigned w4[4] = {raw[u][i].x, raw[u][i].y, raw[u][i].z, raw[u][i].w};
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const float lo = __uint_as_float(w4[q] << 16), hi = __uint_as_float(w4[q] & 0xffff0000u);
                            s1[i] += lo + hi; s2[i] += lo * lo + hi * hi;
                        }
                    }
            }
#pragma unroll
            for (int i = 0; i < RT; i++) {
                s1[i] += __shfl_xor(s1[i], 16); s1[i] += __shfl_xor(s1[i], 32);
                s2[i] += __shfl_xor(s2[i], 16); s2[i] += __shfl_xor(s2[i], 32);
                mean[i] = s1[i] / (float)a.K;
                rstd[i] = rsqrtf(fmaxf(s2[i] / (float)a.K - mean[i] * mean[i], 0.f) + a.ln_eps);
            }
        }
        for (int kb = 0; kb < nks; kb += KU) {
            union { uint4 u; bf16x8 v; } af[KU][RT], bf[KU][4];
            f32x4 lw[KU][2], lb[KU][2];
#pragma unroll
            for (int u = 0; u < KU; u++) {
                const int ks = kb + u;                                // nks is a multiple of KU (checked by the host)
#pragma unroll
                for (int i = 0; i < RT; i++) af[u][i].u = *(const uint4 *)(ap[i] + 32 * ks);
#pragma unroll
                for (int c = 0; c < 4; c++) bf[u][c].u = bp[(ks * 4 + c) * 64];
                if (MODE == 2) {
                    const float *pw = a.ln_w + 32 * (ks0 + ks) + 8 * l4, *pb = a.ln_b + 32 * (ks0 + ks) + 8 * l4;
                    lw[u][0] = *(const f32x4 *)pw; lw[u][1] = *(const f32x4 *)(pw + 4);
                    lb[u][0] = *(const f32x4 *)pb; lb[u][1] = *(const f32x4 *)(pb + 4);
                }
            }
            __builtin_amdgcn_sched_barrier(0);        // every load of the batch is issued before its first MFMA
            if (MODE == 2) {
#pragma unroll
                for (int u = 0; u < KU; u++)
#pragma unroll
                    for (int i = 0; i < RT; i++) {
                        const unsigned w4[4] = {af[u][i].u.x, af[u][i].u.y, af[u][i].u.z, af[u][i].u.w};
                        float v[8];
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            v[2 * q] = (__uint_as_float(w4[q] << 16) - mean[i]) * rstd[i] * lw[u][q >> 1][(2 * q) & 3] + lb[u][q >> 1][(2 * q) & 3];
                            v[2 * q + 1] = (__uint_as_float(w4[q] & 0xffff0000u) - mean[i]) * rstd[i] * lw[u][q >> 1][(2 * q + 1) & 3] + lb[u][q >> 1][(2 * q + 1) & 3];
                        }
                        af[u][i].u = pack8(v);
                    }
            }
#pragma unroll
            for (int u = 0; u < KU; u++) {
#pragma unroll
                for (int i = 0; i < RT; i++)
#pragma unroll
                    for (int c = 0; c < 4; c++) acc[i][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[u][i].v, bf[u][c].v, acc[i][c], 0, 0, 0);
            }
        }
        const int col0 = 64 * ng + 4 * l15;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (MODE >= 1) bv = *(const f32x4 *)(a.bias + col0);
#pragma unroll
        for (int i = 0; i < RT; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int row = 16 * RT * rt + 16 * i + 4 * l4 + j;
                if (row >= nvalid) continue;
                f32x4 v = {acc[i][0][j], acc[i][1][j], acc[i][2][j], acc[i][3][j]};
                if (MODE == 0) {
                    *(f32x4 *)(a.P + ((size_t)s * a.M + row) * a.N + col0) = v;
                } else if (MODE == 2) {                               // merged heads: logits float32, tanh(value) (nn.py:82-83)
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const int col = col0 + c;
                        const float x = v[c] + bv[c];
                        if (col < a.action_dim) a.logits[(size_t)row * a.action_dim + col] = x;
                        else if (col == a.action_dim) a.values[row] = tanhf(x);
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < 4; c++) { const float x = v[c] + bv[c]; v[c] = 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }   // nn.GELU (exact)
                    union { bf16x4 b; uint2 u; } o;
                    o.b = __builtin_convertvector(v, bf16x4);
                    *(uint2 *)(a.out + (size_t)row * a.N + col0) = o.u;
                }
            }
    }
}

// x = sum_s P[s][row][:] (+ bias) (+ resid[row][:]);  y = LayerNorm(x) (bf16);  optionally xout = x + add_bias (bf16)
template <int VPL>
__global__ __launch_bounds__(256) void k_ln_sum(const float *__restrict__ P, int nsplit, int M, const float *__restrict__ bias,
                                                const unsigned short *__restrict__ resid, const float *__restrict__ w,
                                                const float *__restrict__ b, float eps, unsigned short *__restrict__ y,
                                                const float *__restrict__ add_bias, unsigned short *__restrict__ xout, int n,
                                                const int *count) {
    constexpr int D = 64 * VPL;
    const int nvalid = count ? min(n, *count) : n;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= nvalid) return;
    float v[VPL];
#pragma unroll
    for (int q = 0; q < VPL; q++) v[q] = bias ? bias[lane * VPL + q] : 0.f;
    for (int s0 = 0; s0 < nsplit; s0 += 4) {                          // four planes' loads in flight together
        f32x4 t[4][VPL / 4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const float *p = P + ((size_t)min(s0 + u, nsplit - 1) * M + row) * D + lane * VPL;
#pragma unroll
            for (int q = 0; q < VPL / 4; q++) t[u][q] = *(const f32x4 *)(p + 4 * q);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (s0 + u >= nsplit) break;
#pragma unroll
            for (int q = 0; q < VPL / 4; q++) { v[4 * q] += t[u][q][0]; v[4 * q + 1] += t[u][q][1]; v[4 * q + 2] += t[u][q][2]; v[4 * q + 3] += t[u][q][3]; }
        }
    }
    if (resid) {
        const unsigned short *rr = resid + (size_t)row * D + lane * VPL;
#pragma unroll
        for (int q = 0; q < VPL; q++) v[q] += __uint_as_float((unsigned)rr[q] << 16);
    }
    float s1 = 0.f;
#pragma unroll
    for (int q = 0; q < VPL; q++) s1 += v[q];
    const float mean = wave64_sum(s1) * (1.0f / D);
    float ss = 0.f;
#pragma unroll
    for (int q = 0; q < VPL; q++) { const float d = v[q] - mean; ss += d * d; }
    const float rstd = rsqrtf(wave64_sum(ss) * (1.0f / D) + eps);
    float o[VPL], r[VPL];
#pragma unroll
    for (int q = 0; q < VPL; q++) {
        o[q] = (v[q] - mean) * rstd * w[lane * VPL + q] + b[lane * VPL + q];
        r[q] = v[q] + (add_bias ? add_bias[lane * VPL + q] : 0.f);
    }
    if constexpr (VPL == 8) {
        *(uint4 *)(y + (size_t)row * D + lane * VPL) = pack8(o);
        if (xout) *(uint4 *)(xout + (size_t)row * D + lane * VPL) = pack8(r);
    } else {
#pragma unroll
        for (int q = 0; q < VPL; q++) {
            y[(size_t)row * D + lane * VPL + q] = __builtin_bit_cast(unsigned short, (__bf16)o[q]);
            if (xout) xout[(size_t)row * D + lane * VPL + q] = __builtin_bit_cast(unsigned short, (__bf16)r[q]);
        }
    }
}

// logits[row][a] = sum_s P[s][row][a] + bias[a] (a < A);  values[row] = tanh(sum_s P[s][row][A] + bias[A])   (nn.py:82-83)
__global__ void k_heads_finalize_sum(const float *__restrict__ P, int nsplit, int M, int ld, const float *__restrict__ bias, int A, int n,
                                     float *__restrict__ logits, float *__restrict__ values, const int *count) {
    const int nvalid = count ? min(n, *count) : n;
    const long long total = (long long)nvalid * (A + 1);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(i / (A + 1)), col = (int)(i - (long long)row * (A + 1));
        float v = bias[col];
        for (int s = 0; s < nsplit; s++) v += P[((size_t)s * M + row) * ld + col];
        if (col < A) logits[(size_t)row * A + col] = v;
        else values[row] = tanhf(v);
    }
}

// Final LayerNorm + merged heads for embed_dim = 32 KT, LayerNorm's affine folded into the weight (W diag(gamma)) and the bias
// (W beta + b) by the caller: one wave = 16 rows x 64 output columns; the wave's 16 x K slab of x is fetched ONCE (KT loads in
// flight together with the first weight fragments), the row statistics come from those registers, the normalised fragments
// feed the MFMAs - three to four dependent memory round trips per wave instead of eight.
template <int KT>
__global__ __launch_bounds__(256) void k_ln_heads(GemmArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, l4 = lane >> 4;
    const int nvalid = a.count ? min(a.M, *a.count) : a.M;
    const int rtiles = (nvalid + 15) >> 4, ngroups = a.N >> 6;
    const int nitems = rtiles * ngroups;
    constexpr int K = 32 * KT;
    for (int item = blockIdx.x * 4 + wave; item < nitems; item += gridDim.x * 4) {
        const int ng = item % ngroups, rt = item / ngroups;
        const int row = min(16 * rt + l15, nvalid - 1);
        const unsigned short *ap = a.A + (size_t)row * a.lda + 8 * l4;
        const uint4 *bp = a.Wp + (size_t)ng * KT * 4 * 64 + lane;
        uint4 raw[KT];
#pragma unroll
        for (int ks = 0; ks < KT; ks++) raw[ks] = *(const uint4 *)(ap + 32 * ks);
        union BF { uint4 u; bf16x8 v; };
        BF bf[4][4];
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int c = 0; c < 4; c++) bf[u][c].u = bp[(u * 4 + c) * 64];
        __builtin_amdgcn_sched_barrier(0);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int ks = 0; ks < KT; ks++) {
            const unsigned w4[4] = {raw[ks].x, raw[ks].y, raw[ks].z, raw[ks].w};
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float lo = __uint_as_float(w4[q] << 16), hi = __uint_as_float(w4[q] & 0xffff0000u);
                s1 += lo + hi; s2 += lo * lo + hi * hi;
            }
        }
        s1 += __shfl_xor(s1, 16); s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 16); s2 += __shfl_xor(s2, 32);
        const float mean = s1 * (1.0f / K);
        const float rstd = rsqrtf(fmaxf(s2 * (1.0f / K) - mean * mean, 0.f) + a.ln_eps);
        const float shift = -mean * rstd;
        f32x4 acc[4];
#pragma unroll
        for (int c = 0; c < 4; c++) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < KT; kb += 4) {
            BF nb[4][4];
            if (kb + 4 < KT) {
#pragma unroll
                for (int u = 0; u < 4; u++)
#pragma unroll
                    for (int c = 0; c < 4; c++) nb[u][c].u = bp[((kb + 4 + u) * 4 + c) * 64];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint4 r = raw[kb + u];
                const unsigned w4[4] = {r.x, r.y, r.z, r.w};
                float v[8];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    v[2 * q] = __uint_as_float(w4[q] << 16) * rstd + shift;
                    v[2 * q + 1] = __uint_as_float(w4[q] & 0xffff0000u) * rstd + shift;
                }
                BF af;
                af.u = pack8(v);
#pragma unroll
                for (int c = 0; c < 4; c++) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af.v, bf[u][c].v, acc[c], 0, 0, 0);
            }
            if (kb + 4 < KT) {
#pragma unroll
                for (int u = 0; u < 4; u++)
#pragma unroll
                    for (int c = 0; c < 4; c++) bf[u][c] = nb[u][c];
            }
        }
        const int col0 = 64 * ng + 4 * l15;
        const f32x4 bv = *(const f32x4 *)(a.bias + col0);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int orow = 16 * rt + 4 * l4 + j;
            if (orow >= nvalid) continue;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int col = col0 + c;
                const float x = acc[c][j] + bv[c];
                if (col < a.action_dim) a.logits[(size_t)orow * a.action_dim + col] = x;
                else if (col == a.action_dim) a.values[orow] = tanhf(x);
            }
        }
    }
}
}  // namespace

extern "C" int32_t azk_nn_gemm_rows(const void *a_bf16_dev, int32_t lda, const void *w_packed_dev, int32_t m, int32_t n_out,
                                    int32_t k, int32_t ksplit, float *partials_out_dev, const float *bias_dev,
                                    void *gelu_out_bf16_dev, const int32_t *n_valid_dev, void *stream) {
    if (!a_bf16_dev || !w_packed_dev || m < 0 || n_out < 64 || (n_out & 63) || k < 32 || (k & 31) || ksplit < 1) return AZK_ERR_ARG;
    if ((k / 32) % (4 * ksplit) != 0 || lda < k || (lda & 7)) return AZK_ERR_ARG;     // k-steps per split: a multiple of the batch of 4
    if ((partials_out_dev != nullptr) == (gelu_out_bf16_dev != nullptr)) return AZK_ERR_ARG;
    if (gelu_out_bf16_dev && (!bias_dev || ksplit != 1)) return AZK_ERR_ARG;
    if (m == 0) return AZK_OK;
    GemmArgs a;
    a.A = (const unsigned short *)a_bf16_dev; a.lda = lda; a.Wp = (const uint4 *)w_packed_dev; a.M = m; a.N = n_out; a.K = k;
    a.ksplit = ksplit; a.count = n_valid_dev; a.P = partials_out_dev; a.bias = bias_dev; a.out = (unsigned short *)gelu_out_bf16_dev;
    a.ln_w = a.ln_b = nullptr; a.ln_eps = 0.f; a.logits = a.values = nullptr; a.action_dim = 0;
    const long long items = (long long)((m + 31) / 32) * (n_out / 64) * ksplit;      // 32-row x 64-column wave tiles
    const unsigned blocks = (unsigned)((items + 3) / 4 < 4096 ? (items + 3) / 4 : 4096);
    hipStream_t st = (hipStream_t)stream;
    if (partials_out_dev) k_gemm_rows<0, 2, 4><<<blocks, 256, 0, st>>>(a);
    else k_gemm_rows<1, 2, 4><<<blocks, 256, 0, st>>>(a);
    return hipGetLastError() == hipSuccess ? AZK_OK : AZK_ERR_HIP;
}

extern "C" int32_t azk_nn_ln_heads(const void *x_bf16_dev, const float *ln_w_dev, const float *ln_b_dev, float eps,
                                   const void *w_packed_dev, const float *bias_dev, int32_t n, int32_t embed_dim, int32_t n_out_padded,
                                   int32_t action_dim, float *logits_out_dev, float *values_out_dev, const int32_t *n_valid_dev,
                                   void *stream) {
    if (!x_bf16_dev || (ln_w_dev == nullptr) != (ln_b_dev == nullptr) || !w_packed_dev || !bias_dev || !logits_out_dev || !values_out_dev) return AZK_ERR_ARG;
    if (n < 0 || embed_dim < 128 || (embed_dim & 127) || n_out_padded < 64 || (n_out_padded & 63) || action_dim + 1 > n_out_padded) return AZK_ERR_ARG;
    if (n == 0) return AZK_OK;
    GemmArgs a;
    a.A = (const unsigned short *)x_bf16_dev; a.lda = embed_dim; a.Wp = (const uint4 *)w_packed_dev; a.M = n; a.N = n_out_padded;
    a.K = embed_dim; a.ksplit = 1; a.count = n_valid_dev; a.P = nullptr; a.bias = bias_dev; a.out = nullptr;
    a.ln_w = ln_w_dev; a.ln_b = ln_b_dev; a.ln_eps = eps; a.logits = logits_out_dev; a.values = values_out_dev; a.action_dim = action_dim;
    const long long items = (long long)((n + 15) / 16) * (n_out_padded / 64);       // 16-row x 64-column wave tiles
    const unsigned blocks = (unsigned)((items + 3) / 4 < 4096 ? (items + 3) / 4 : 4096);
    if (!ln_w_dev) {                                                                 // affine folded into weight and bias by the caller
        if (embed_dim == 512) k_ln_heads<16><<<blocks, 256, 0, (hipStream_t)stream>>>(a);
        else if (embed_dim == 256) k_ln_heads<8><<<blocks, 256, 0, (hipStream_t)stream>>>(a);
        else return AZK_ERR_ARG;
        return hipGetLastError() == hipSuccess ? AZK_OK : AZK_ERR_HIP;
    }
    k_gemm_rows<2, 1, 4><<<blocks, 256, 0, (hipStream_t)stream>>>(a);
    return hipGetLastError() == hipSuccess ? AZK_OK : AZK_ERR_HIP;
}

extern "C" int32_t azk_nn_layernorm_sum(const float *partials_dev, int32_t nsplit, int32_t m_stride, const float *bias_dev,
                                        const void *resid_bf16_dev, const float *w_dev, const float *b_dev, float eps,
                                        void *y_bf16_dev, const float *add_bias_dev, void *x_out_bf16_dev, int32_t n,
                                        int32_t embed_dim, const int32_t *n_valid_dev, void *stream) {
    if (!partials_dev || nsplit < 1 || !w_dev || !b_dev || !y_bf16_dev || n < 0 || m_stride < n) return AZK_ERR_ARG;
    if (embed_dim != 256 && embed_dim != 512) return AZK_ERR_ARG;
    if (n == 0) return AZK_OK;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((n + 3) / 4), block(256);
    const unsigned short *rs = (const unsigned short *)resid_bf16_dev;
    unsigned short *y = (unsigned short *)y_bf16_dev, *xo = (unsigned short *)x_out_bf16_dev;
    if (embed_dim == 512) k_ln_sum<8><<<grid, block, 0, st>>>(partials_dev, nsplit, m_stride, bias_dev, rs, w_dev, b_dev, eps, y, add_bias_dev, xo, n, n_valid_dev);
    else k_ln_sum<4><<<grid, block, 0, st>>>(partials_dev, nsplit, m_stride, bias_dev, rs, w_dev, b_dev, eps, y, add_bias_dev, xo, n, n_valid_dev);
    return hipGetLastError() == hipSuccess ? AZK_OK : AZK_ERR_HIP;
}

extern "C" int32_t azk_nn_heads_finalize_sum(const float *partials_dev, int32_t nsplit, int32_t m_stride, int32_t ld,
                                             const float *bias_dev, int32_t action_dim, int32_t n, float *logits_out_dev,
                                             float *values_out_dev, const int32_t *n_valid_dev, void *stream) {
    if (!partials_dev || nsplit < 1 || !bias_dev || !logits_out_dev || !values_out_dev || ld < action_dim + 1 || n < 0 || m_stride < n) return AZK_ERR_ARG;
    if (n == 0) return AZK_OK;
    k_heads_finalize_sum<<<1024, 256, 0, (hipStream_t)stream>>>(partials_dev, nsplit, m_stride, ld, bias_dev, action_dim, n, logits_out_dev,
                                                               values_out_dev, n_valid_dev);
    return hipGetLastError() == hipSuccess ? AZK_OK : AZK_ERR_HIP;
}

// =====================================================================================================
// k_tail_gemm: the cls-row tail (nn.py:54-60, 78-83 for the one row the heads read) as a chain of latency-shaped small
// GEMMs.  At ~1000 live rows every GEMM of the tail is a few MFLOP per CU: what costs time is the number of dependent
// memory round trips, so a wave issues EVERY load of its K range (A rows and weight fragments) before its first MFMA: one
// round trip.  NWK > 1 splits K over the waves of a workgroup (partials meet in LDS): a quarter of the loads and MFMAs on
// each wave's chain - what the three small GEMMs of the tail use.
//   wave tile 16 RT rows x 64 columns (RT picked in the kernel from the live row count, see k_tail_gemm below), no LDS staging;
//   A fragments straight from the row-major activations, B fragments from weights packed in fragment order
//   (pack_linear_weight: one 16-byte load per fragment);
//   AMODE 1: A = LayerNorm(rows) with the affine folded into weight and bias by the caller.  The row statistics come from the
//            PRODUCING GEMM: its epilogue leaves, per row and 64-column group, the (sum, sum of squares) of the bf16 values it
//            stored; the consumer adds the groups in a fixed order (deterministic, no atomics) and normalises its fragments on
//            the fly - LayerNorm costs no pass over the rows at all;
//   batched (block-diagonal) form: batch b reads A columns [b a_batch, b a_batch + K), its own weight block, and writes output
//            columns [b N, (b+1) N) - the per-head value projection (8 heads x [64 x 512]) in one launch;
//   epilogues: bf16 (+ bias), bf16 GELU(+ bias), bf16 (+ bias + residual), merged policy / value heads (float32 logits, tanh).
// Rows at or beyond *count are neither read nor written; the grid is sized for the full buffer and idle workgroups exit.
// =====================================================================================================
namespace {

using azk_tail::TailArgs;                         // the argument block, the epilogue selectors and the GELU arithmetic are shared with the
using azk_tail::gelu_erf;                         // LDS-staged form of the wide links (azk_tail.hip)
using azk_tail::TAIL_EPI_BF16;
using azk_tail::TAIL_EPI_GELU;
using azk_tail::TAIL_EPI_RESID;
using azk_tail::TAIL_EPI_HEADS;

// NWK > 1: the K range is split over NWK waves of the workgroup (every load of the whole K in flight at once, one round trip),
// their partial accumulators meet in LDS and wave 0 runs the epilogue.
// A wave keeps every A fragment of its K range in flight at once; the weight fragments are all in flight too when RT <= 2, and
// for the taller tiles the second half of them is fetched into the registers the matrix pipe has just consumed.
template <int RT, int KCH, int AMODE, int EPI, int NWR, int NWC, int NWK>
__device__ __forceinline__ void tail_items(const TailArgs &a, const int nvalid, f32x4 *kred) {
    constexpr int K = 32 * KCH * NWK, KS = KCH * NWK;
    constexpr int BH = RT > 3 ? KCH / 4 : RT > 2 ? KCH / 2 : KCH;  // weight k-steps in flight before the first MFMA
    static_assert(NWK == 1 || (NWR == 1 && NWC == 1), "split-K workgroups hold one wave tile");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, l4 = lane >> 4;
    const int wk = NWK > 1 ? wave : 0, wrc = NWK > 1 ? 0 : wave;
    const int wr = wrc / NWC, wc = wrc - wr * NWC;
    constexpr int WROWS = 16 * RT * NWR;
    const int rtiles = (nvalid + WROWS - 1) / WROWS, ctiles = a.N / (64 * NWC);
    const int nitems = rtiles * ctiles * a.nbatch;
    union BF { uint4 u; bf16x8 v; };
    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        if (NWK > 1 && item != (int)blockIdx.x) __syncthreads();      // wave 0 is done with the previous item's partial sums
        const int ct = item % ctiles, r2 = item / ctiles, rt = r2 % rtiles, b = r2 / rtiles;
        const int row0 = rt * WROWS + wr * 16 * RT, g = ct * NWC + wc;
        const unsigned short *ap[RT];
#pragma unroll
        for (int i = 0; i < RT; i++) ap[i] = a.A + (size_t)min(row0 + 16 * i + l15, a.M - 1) * a.lda + (size_t)b * a.a_batch + 8 * l4 + 32 * KCH * wk;
        const uint4 *bp = a.Wp + (size_t)b * a.w_batch + ((size_t)g * KS + (size_t)KCH * wk) * 4 * 64 + lane;
        f32x4 acc[RT][4];
#pragma unroll
        for (int i = 0; i < RT; i++)
#pragma unroll
            for (int c = 0; c < 4; c++) acc[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        // AMODE 1: the row statistics were left by the producing GEMM as per-column-group partial sums (of the bf16 values this
        // wave now reads), eight (sum, sum of squares) pairs per row: fetched FIRST, as four 16-byte loads per row, so that they
        // are back before the fragments they normalise (a scalar loop over the groups here cost one round trip per group)
        f32x4 st[AMODE == 1 ? RT : 1];                        // a lane fetches the quarter l4 of its row's 64 bytes
        if (AMODE == 1) {
#pragma unroll
            for (int i = 0; i < RT; i++) st[i] = *((const f32x4 *)(a.stats_in + (size_t)min(row0 + 16 * i + l15, a.M - 1) * 16) + l4);
        }
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        const int col0 = b * a.N + 64 * g + 4 * l15;          // a lane's four accumulators of a row are four consecutive output columns
        if (a.bias) bv = *(const f32x4 *)(a.bias + col0);
        BF af[RT][KCH], bf[KCH][4];
#pragma unroll
        for (int s = 0; s < KCH; s++) {
#pragma unroll
            for (int i = 0; i < RT; i++) af[i][s].u = *(const uint4 *)(ap[i] + 32 * s);
            if (s < BH) {
#pragma unroll
                for (int c = 0; c < 4; c++) bf[s][c].u = bp[(s * 4 + c) * 64];
            }
        }
        float rstd[RT], shift[RT];
        if (AMODE == 1) {                                     // the groups are added in a fixed order: deterministic, no atomics
#pragma unroll
            for (int i = 0; i < RT; i++) {
                float s1 = st[i][0] + st[i][2], s2 = st[i][1] + st[i][3];
                s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
                s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
                const float mean = s1 * (1.0f / K);
                rstd[i] = rsqrtf(fmaxf(s2 * (1.0f / K) - mean * mean, 0.f) + a.ln_eps);
                shift[i] = -mean * rstd[i];
            }
        }
        __builtin_amdgcn_sched_barrier(0);                    // every load above is issued before the first MFMA
#pragma unroll
        for (int s = 0; s < KCH; s++) {
            if (AMODE == 1) {
#pragma unroll
                for (int i = 0; i < RT; i++) {
                    const unsigned w4[4] = {af[i][s].u.x, af[i][s].u.y, af[i][s].u.z, af[i][s].u.w};
                    float v[8];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        v[2 * q] = __uint_as_float(w4[q] << 16) * rstd[i] + shift[i];
                        v[2 * q + 1] = __uint_as_float(w4[q] & 0xffff0000u) * rstd[i] + shift[i];
                    }
                    af[i][s].u = pack8(v);
                }
            }
#pragma unroll
            for (int i = 0; i < RT; i++)
#pragma unroll
                for (int c = 0; c < 4; c++) acc[i][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][s].v, bf[s][c].v, acc[i][c], 0, 0, 0);
            if (BH < KCH && s + BH < KCH) {
#pragma unroll
                for (int c = 0; c < 4; c++) bf[s + BH][c].u = bp[((s + BH) * 4 + c) * 64];
                __builtin_amdgcn_sched_barrier(0);            // (the refill stays behind the MFMAs that free its registers)
            }
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
        // epilogue: every load (residual rows) first, then straight-line arithmetic, then the stores under their row predicate with
        // nothing to wait for in between (a load or a branch between two stores makes every store wait for the one before it)
        uint2 rr[EPI == TAIL_EPI_RESID ? RT : 1][4];
        if (EPI == TAIL_EPI_RESID) {
#pragma unroll
            for (int i = 0; i < RT; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) rr[i][j] = *(const uint2 *)(a.resid + (size_t)min(row0 + 16 * i + 4 * l4 + j, a.M - 1) * a.ldr + col0);
        }
        if (EPI == TAIL_EPI_HEADS) {                          // nn.py:82-83
            f32x4 hv[RT][4];
#pragma unroll
            for (int i = 0; i < RT; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    hv[i][j] = f32x4{acc[i][0][j] + bv[0], acc[i][1][j] + bv[1], acc[i][2][j] + bv[2], acc[i][3][j] + bv[3]};
#pragma unroll
                    for (int c = 0; c < 4; c++)
                        if (col0 + c == a.action_dim) hv[i][j][c] = tanhf(hv[i][j][c]);
                }
#pragma unroll
            for (int i = 0; i < RT; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int row = row0 + 16 * i + 4 * l4 + j;
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const int col = col0 + c;
                        if (row < nvalid && col < a.action_dim) a.logits[(size_t)row * a.action_dim + col] = hv[i][j][c];
                        if (row < nvalid && col == a.action_dim) a.values[row] = hv[i][j][c];
                    }
                }
        } else {
            uint2 o[RT][4];
            f32x2 ps[RT][4];
#pragma unroll
            for (int i = 0; i < RT; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    f32x4 v = {acc[i][0][j] + bv[0], acc[i][1][j] + bv[1], acc[i][2][j] + bv[2], acc[i][3][j] + bv[3]};
                    if (EPI == TAIL_EPI_GELU) {
#pragma unroll
                        for (int c = 0; c < 4; c++) v[c] = gelu_erf(v[c]);                                      // nn.GELU (erf form)
                    }
                    if (EPI == TAIL_EPI_RESID) {
                        v[0] += __uint_as_float(rr[i][j].x << 16); v[1] += __uint_as_float(rr[i][j].x & 0xffff0000u);
                        v[2] += __uint_as_float(rr[i][j].y << 16); v[3] += __uint_as_float(rr[i][j].y & 0xffff0000u);
                    }
                    union { bf16x4 b4; uint2 u; } ob;
                    ob.b4 = __builtin_convertvector(v, bf16x4);
                    o[i][j] = ob.u;
                    if (a.stats_out) {
                        // partial LayerNorm statistics of the row, over this wave's 64 columns, from the ROUNDED values
                        const f32x4 vr = __builtin_convertvector(ob.b4, f32x4);
                        ps[i][j] = f32x2{row16_sum((vr[0] + vr[1]) + (vr[2] + vr[3])),
                                         row16_sum((vr[0] * vr[0] + vr[1] * vr[1]) + (vr[2] * vr[2] + vr[3] * vr[3]))};
                    }
                }
            const int ngr = a.nbatch * (a.N >> 6), gr = b * (a.N >> 6) + g;
#pragma unroll
            for (int i = 0; i < RT; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int row = row0 + 16 * i + 4 * l4 + j;
                    if (row < nvalid) *(uint2 *)(a.out + (size_t)row * a.ldo + col0) = o[i][j];
                    if (a.stats_out && l15 == 0 && row < nvalid) *(f32x2 *)(a.stats_out + ((size_t)row * ngr + gr) * 2) = ps[i][j];
                }
        }
    }
}

// The wave tile is 16 RT rows tall, RT chosen per launch from the LIVE row count (RTLO..RTHI): the smallest tile that still puts
// every wave of the launch on the chip at once.  These kernels hold their whole K range in registers (one wave per SIMD), so a
// launch with more waves than SIMDs runs as two rounds of the same latency chain - measured: the five-launch tail took 54 us at
// 1024 live rows and 79 us at 1056 with a fixed 32-row tile.
template <int RTLO, int RTHI, int KCH, int AMODE, int EPI, int NWR, int NWC, int NWK>
__global__ __launch_bounds__(64 * NWR * NWC * NWK, 1) void k_tail_gemm(TailArgs a) {
    __shared__ f32x4 kred[NWK > 1 ? (NWK - 1) * RTHI * 4 * 64 : 1];
    // the items cover the live rows only (measured: letting the dead half of the buffer issue its loads too costs 40-60 % - these
    // GEMMs move ~100 KB per wave through L2 and are bound by that traffic, not by the count's extra round trip)
    // (every kernel argument is wanted in SGPRs HERE: left alone, the compiler fetches the count pointer first, waits for the count,
    //  and only then goes back to the argument segment for the rest - a third dependent scalar round trip before the first load)
    asm volatile("" :: "s"(a.A), "s"(a.Wp), "s"(a.out), "s"(a.bias), "s"(a.resid), "s"(a.stats_in), "s"(a.stats_out), "s"(a.logits), "s"(a.values),
                 "s"(a.lda), "s"(a.ldo), "s"(a.N), "s"(a.nbatch), "s"(a.wave_slots), "s"(a.M), "s"(a.a_batch), "s"(a.w_batch), "s"(a.ldr),
                 "s"(a.ln_eps), "s"(a.action_dim));
    const int nvalid = a.count ? min(a.M, *a.count) : a.M;
    const int strip_waves = (a.N >> 6) * a.nbatch * NWK;          // waves per strip of 16 RT rows
    int rt = RTLO;
    while (rt < RTHI && ((nvalid + 16 * rt * NWR - 1) / (16 * rt * NWR)) * NWR * strip_waves > a.wave_slots) rt++;
    if (RTHI >= RTLO + 2 && rt == RTLO + 2) tail_items<(RTHI >= RTLO + 2 ? RTLO + 2 : RTLO), KCH, AMODE, EPI, NWR, NWC, NWK>(a, nvalid, kred);
    else if (RTHI >= RTLO + 1 && rt == RTLO + 1) tail_items<(RTHI >= RTLO + 1 ? RTLO + 1 : RTLO), KCH, AMODE, EPI, NWR, NWC, NWK>(a, nvalid, kred);
    else tail_items<RTLO, KCH, AMODE, EPI, NWR, NWC, NWK>(a, nvalid, kred);
}

int tail_cu_count() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    }
    return cus;
}

template <int RTLO, int RTHI, int KCH, int AMODE, int EPI, int NWR, int NWC, int NWK = 1>
int launch_tail(TailArgs &a, hipStream_t st) {
    static_assert(RTHI <= RTLO + 2, "three tile heights per kernel");
    // waves of this instantiation the chip holds at once (its register footprint decides: 1 per SIMD for the whole-K variants,
    // 2-3 for the split-K ones)
    static int slots = 0;
    if (!slots) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_tail_gemm<RTLO, RTHI, KCH, AMODE, EPI, NWR, NWC, NWK>, 64 * NWR * NWC * NWK, 0) != hipSuccess || nb < 1) nb = 1;
        slots = nb * NWR * NWC * NWK * tail_cu_count();
    }
    a.wave_slots = slots;
    const long long items = (long long)((a.M + 16 * RTLO * NWR - 1) / (16 * RTLO * NWR)) * (a.N / (64 * NWC)) * a.nbatch;
    const unsigned blocks = (unsigned)(items < 8192 ? items : 8192);
    k_tail_gemm<RTLO, RTHI, KCH, AMODE, EPI, NWR, NWC, NWK><<<blocks, 64 * NWR * NWC * NWK, 0, st>>>(a);
    return hipGetLastError() == hipSuccess ? AZK_OK : AZK_ERR_HIP;
}
}  // namespace

extern "C" int32_t azk_nn_tail_gemm(const azk_tail_gemm *t, void *stream) {
    if (!t || !t->a_bf16 || !t->w_packed || t->m < 0 || t->n_out < 64 || (t->n_out & 63) || t->nbatch < 1) return AZK_ERR_ARG;
    if ((t->k != 512 && t->k != 2048 && t->k != 384) || t->lda < t->k || (t->lda & 7) || (t->a_batch_stride & 7)) return AZK_ERR_ARG;
    if (t->epilogue < 0 || t->epilogue > 3 || (t->layernorm_a && (t->k != 512 || !t->a_stats || t->a_stats_groups != 8))) return AZK_ERR_ARG;
    if (t->epilogue == TAIL_EPI_HEADS ? (!t->logits_out || !t->values_out || t->action_dim + 1 > t->n_out * t->nbatch) : (!t->out_bf16 || t->ldo < t->n_out * t->nbatch || (t->ldo & 3)))
        return AZK_ERR_ARG;
    if (t->epilogue == TAIL_EPI_RESID && (!t->resid_bf16 || (t->ldr & 3))) return AZK_ERR_ARG;
    if (t->m == 0) return AZK_OK;
    TailArgs a = {};
    a.A = (const unsigned short *)t->a_bf16; a.lda = t->lda; a.a_batch = t->a_batch_stride; a.Wp = (const uint4 *)t->w_packed;
    a.w_batch = (long long)(t->n_out / 64) * (t->k / 32) * 4 * 64;
    a.M = t->m; a.N = t->n_out; a.nbatch = t->nbatch; a.count = t->n_valid; a.bias = t->bias; a.out = (unsigned short *)t->out_bf16; a.ldo = t->ldo;
    a.resid = (const unsigned short *)t->resid_bf16; a.ldr = t->ldr; a.ln_eps = t->ln_eps; a.logits = t->logits_out; a.values = t->values_out;
    a.action_dim = t->action_dim; a.stats_in = t->a_stats; a.stats_groups = t->a_stats_groups; a.stats_out = t->stats_out;
    hipStream_t st = (hipStream_t)stream;
    const bool wide = t->n_out % 128 == 0 && t->n_out >= 1024;          // many column groups: 2 x 2 waves share A rows and weight fragments in L1
    if (t->k == 384)                                                    // the embed-fold rows (AZK_EMBED_FOLD_ROW) against [D_t; U_all; M_h]: K split over four waves
        return !t->layernorm_a && t->epilogue == TAIL_EPI_BF16 ? launch_tail<2, 2, 3, 0, TAIL_EPI_BF16, 1, 1, 4>(a, st) : AZK_ERR_ARG;
    if (t->k == 512) {
        if (t->layernorm_a) {
            if (t->epilogue == TAIL_EPI_GELU) return wide ? launch_tail<2, 4, 16, 1, TAIL_EPI_GELU, 2, 2>(a, st) : launch_tail<2, 2, 16, 1, TAIL_EPI_GELU, 1, 1>(a, st);
            if (t->epilogue == TAIL_EPI_HEADS) return launch_tail<1, 1, 4, 1, TAIL_EPI_HEADS, 1, 1, 4>(a, st);
            if (t->epilogue == TAIL_EPI_BF16) return launch_tail<2, 2, 16, 1, TAIL_EPI_BF16, 1, 1>(a, st);
            return AZK_ERR_ARG;
        }
        if (t->epilogue == TAIL_EPI_BF16) return launch_tail<2, 2, 4, 0, TAIL_EPI_BF16, 1, 1, 4>(a, st);       // (K split over four waves: a quarter of the loads and MFMAs on each wave's chain)
        if (t->epilogue == TAIL_EPI_GELU) return wide ? launch_tail<2, 4, 16, 0, TAIL_EPI_GELU, 2, 2>(a, st) : launch_tail<2, 2, 16, 0, TAIL_EPI_GELU, 1, 1>(a, st);
        if (t->epilogue == TAIL_EPI_RESID) return launch_tail<2, 2, 16, 0, TAIL_EPI_RESID, 1, 1>(a, st);
        return launch_tail<2, 2, 16, 0, TAIL_EPI_HEADS, 1, 1>(a, st);
    }
    // k = 2048: four waves of a workgroup take 512 columns of K each
    if (t->epilogue == TAIL_EPI_RESID) return launch_tail<2, 4, 16, 0, TAIL_EPI_RESID, 1, 1, 4>(a, st);
    if (t->epilogue == TAIL_EPI_BF16) return launch_tail<2, 4, 16, 0, TAIL_EPI_BF16, 1, 1, 4>(a, st);
    return AZK_ERR_ARG;
}
