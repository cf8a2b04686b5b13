// azk_nn.hip - hand-written CDNA4 kernels for the policy-value network's token embedding (ai/nn.py:5-36):
//   tokens[n, 0, :]   = cls_token + pos_embedding[0]
//   tokens[n, 1+j, :] = Conv2d(C -> D, k x k, stride 1, 'same')(board)[:, r, c] + pos_embedding[1+j]      j = r*cols + c
// lowered to an im2col GEMM on the matrix cores: per board A[T x KP] (0/1 patches gathered from the LDS-resident
// board, KP = C*k*k rounded up to 16) times B[KP x D] (the conv weight, held in registers for the kernel's
// lifetime) with v_mfma_f32_32x32x16_bf16, fp32 accumulate.  The epilogue adds bias + positional embedding,
// optionally applies the first block's LayerNorm (nn.py:53) in fp32 and stores whole 1 KiB rows (16 B per lane).
// The kernel is bound by its HBM writes (T*D*2 bytes per board per output), not by MFMA.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>

#include "azk.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

struct EmbedArgs {
    const void *boards;        // [n][C][R][Cc] bf16 or f32
    int boards_f32;
    const __hip_bfloat16 *wt;  // [D][KP] conv weight, k index = ch*k*k + ky*k + kx, zero padded
    const float *cpos;         // [T][D]: row 0 = cls + pos[0]; row 1+j = conv bias + pos[1+j]
    const float *ln_w, *ln_b;  // [D] LayerNorm affine (used when xhat != nullptr)
    __hip_bfloat16 *x;         // [n][T][D] tokens (may be null)
    __hip_bfloat16 *xhat;      // [n][T][D] LayerNorm(tokens) (may be null)
    int n, C, R, Cc, ksz, KP, T;
    float eps;
};

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short u) { return __uint_as_float((unsigned)u << 16); }

// XOR swizzle of the 16-byte chunk index inside an A-tile row (conflict-free ds_read_b128 of the fragments);
// only the leading power-of-two group of chunks is permuted so every index stays inside the row.
template <int KS>
__device__ __forceinline__ int swz(int chunk, int row) {
    constexpr int NCH = 2 * KS;
    constexpr int SW = NCH >= 8 ? 7 : (NCH >= 4 ? 3 : (NCH >= 2 ? 1 : 0));
    return chunk <= SW ? (chunk ^ (row & SW)) : chunk;
}

// NT = 32-column tiles per wave; D = 4 waves * NT * 32.  KS = KP / 16 k-steps.
template <int NT, int KS>
__global__ __launch_bounds__(256) void k_embed(EmbedArgs a) {
    constexpr int D = 128 * NT;
    constexpr int KP = 16 * KS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // LDS: stage f32 [32][D] | A tile bf16 [2][32][KP] | board u8 [C*R*Cc]
    float *stage = (float *)smem;
    unsigned short *atile = (unsigned short *)(smem + 32 * D * 4);
    unsigned char *board = smem + 32 * D * 4 + 2 * 32 * KP * 2;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int RC = a.R * a.Cc, T = a.T, kk = a.ksz * a.ksz, pad = a.ksz / 2, Kreal = a.C * kk;
    const int mtiles = (T + 31) >> 5;

    // B fragments: conv weight columns owned by this wave, resident in registers for the whole kernel
    bf16x8 bfrag[NT][KS];
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
#pragma unroll
        for (int s = 0; s < KS; s++) {
            const int col = wave * 32 * NT + nt * 32 + (lane & 31);
            bfrag[nt][s] = *(const bf16x8 *)(a.wt + (size_t)col * KP + 16 * s + 8 * (lane >> 5));
        }

    // im2col builder role: thread -> (row = tid >> 3 of the 32-row tile, 8 consecutive k's starting at 8*(tid & 7) [+64 per pass])
    const int brow = tid >> 3;

    for (int leaf = blockIdx.x; leaf < a.n; leaf += gridDim.x) {
        __syncthreads();                                          // previous leaf's board / stage fully consumed
        for (int i = tid; i < a.C * RC; i += 256) {
            float v = a.boards_f32 ? ((const float *)a.boards)[(size_t)leaf * a.C * RC + i]
                                   : bf16_bits_to_f32(((const unsigned short *)a.boards)[(size_t)leaf * a.C * RC + i]);
            board[i] = v != 0.0f ? 1 : 0;
        }
        __syncthreads();

        for (int mt = 0; mt < mtiles; mt++) {
            unsigned short *at = atile + (mt & 1) * 32 * KP;
            // ---- build the 32 x KP im2col tile (token t = mt*32 + brow; t == 0 is the cls row: all zeros) ----
            {
                const int t = mt * 32 + brow;
                const int j = t - 1;
                const int r = j / a.Cc, c = j - r * a.Cc;
                const bool live = t >= 1 && t < T;
                for (int kc = tid & 7; kc < KP / 8; kc += 8) {
                    unsigned short v8[8];
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        const int k = kc * 8 + q;
                        unsigned short v = 0;
                        if (live && k < Kreal) {
                            const int ch = k / kk, rem = k - ch * kk, ky = rem / a.ksz, kx = rem - ky * a.ksz;
                            const int rr = r + ky - pad, cc = c + kx - pad;
                            if (rr >= 0 && rr < a.R && cc >= 0 && cc < a.Cc && board[ch * RC + rr * a.Cc + cc]) v = 0x3F80;  // bf16 1.0
                        }
                        v8[q] = v;
                    }
                    const int chunk = swz<KS>(kc, brow);
                    uint4 pk;
                    pk.x = v8[0] | ((unsigned)v8[1] << 16); pk.y = v8[2] | ((unsigned)v8[3] << 16);
                    pk.z = v8[4] | ((unsigned)v8[5] << 16); pk.w = v8[6] | ((unsigned)v8[7] << 16);
                    *(uint4 *)(at + brow * KP + chunk * 8) = pk;
                }
            }
            __syncthreads();
            // ---- MFMA: acc[nt] (32 x 32) += A(32 x KP) * B(KP x 32) ----
            f32x16 acc[NT];
#pragma unroll
            for (int nt = 0; nt < NT; nt++)
#pragma unroll
                for (int i = 0; i < 16; i++) acc[nt][i] = 0.f;
#pragma unroll
            for (int s = 0; s < KS; s++) {
                const int row = lane & 31, chunk = swz<KS>(2 * s + (lane >> 5), row);
                const bf16x8 af = *(const bf16x8 *)(at + row * KP + chunk * 8);
#pragma unroll
                for (int nt = 0; nt < NT; nt++) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfrag[nt][s], acc[nt], 0, 0, 0);
            }
            // ---- epilogue 1: + (bias + pos) and stage the fp32 tile in LDS (C/D map: col = lane&31, row = (i&3)+8*(i>>2)+4*(lane>>5)) ----
#pragma unroll
            for (int nt = 0; nt < NT; nt++)
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const int row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                    const int col = wave * 32 * NT + nt * 32 + (lane & 31);
                    const int t = mt * 32 + row;
                    const float v = acc[nt][i] + (t < T ? a.cpos[(size_t)t * D + col] : 0.f);
                    stage[row * D + col] = v;
                }
            __syncthreads();
            // ---- epilogue 2: each wave finishes 8 rows: optional LayerNorm in fp32, bf16 pack, whole-row stores ----
            for (int rr = 0; rr < 8; rr++) {
                const int row = wave * 8 + rr, t = mt * 32 + row;
                if (t >= T) break;
                const size_t orow = ((size_t)leaf * T + t) * D;
                {
                    const int col = lane * 8;                    // D <= 512: one pass of 64 lanes x 8 columns covers the row
                    const bool act = col < D;
                    float v[8];
                    if (act) {
                        const f32x4 v0 = *(const f32x4 *)(stage + row * D + col);
                        const f32x4 v1 = *(const f32x4 *)(stage + row * D + col + 4);
                        v[0] = v0[0]; v[1] = v0[1]; v[2] = v0[2]; v[3] = v0[3]; v[4] = v1[0]; v[5] = v1[1]; v[6] = v1[2]; v[7] = v1[3];
                    } else {
#pragma unroll
                        for (int q = 0; q < 8; q++) v[q] = 0.f;
                    }
                    if (a.x && act) {
                        unsigned short h[8];
#pragma unroll
                        for (int q = 0; q < 8; q++) h[q] = __bfloat16_as_ushort(__float2bfloat16(v[q]));
                        uint4 pk;
                        pk.x = h[0] | ((unsigned)h[1] << 16); pk.y = h[2] | ((unsigned)h[3] << 16);
                        pk.z = h[4] | ((unsigned)h[5] << 16); pk.w = h[6] | ((unsigned)h[7] << 16);
                        *(uint4 *)(a.x + orow + col) = pk;
                    }
                    if (a.xhat) {
                        float s = 0.f;
#pragma unroll
                        for (int q = 0; q < 8; q++) s += v[q];
#pragma unroll
                        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
                        const float mean = s / (float)D;
                        float ss = 0.f;
                        if (act) {
#pragma unroll
                            for (int q = 0; q < 8; q++) { const float dlt = v[q] - mean; ss += dlt * dlt; }
                        }
#pragma unroll
                        for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
                        const float rstd = rsqrtf(ss / (float)D + a.eps);
                        if (act) {
                            unsigned short h[8];
#pragma unroll
                            for (int q = 0; q < 8; q++)
                                h[q] = __bfloat16_as_ushort(__float2bfloat16((v[q] - mean) * rstd * a.ln_w[col + q] + a.ln_b[col + q]));
                            uint4 pk;
                            pk.x = h[0] | ((unsigned)h[1] << 16); pk.y = h[2] | ((unsigned)h[3] << 16);
                            pk.z = h[4] | ((unsigned)h[5] << 16); pk.w = h[6] | ((unsigned)h[7] << 16);
                            *(uint4 *)(a.xhat + orow + col) = pk;
                        }
                    }
                }
            }
            // the next tile's A build writes the other A buffer; `stage` is rewritten only after the next tile's
            // post-build barrier, which every wave reaches after finishing its rows above
        }
    }
}

template <int NT, int KS>
int launch_embed(const EmbedArgs &a, hipStream_t st) {
    constexpr int D = 128 * NT, KP = 16 * KS;
    const int lds = 32 * D * 4 + 2 * 32 * KP * 2 + ((a.C * a.R * a.Cc + 15) & ~15);
    int grid = a.n < 512 ? a.n : 512;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void *)k_embed<NT, KS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return AZK_ERR_HIP;
        attr_set = true;
    }
    k_embed<NT, KS><<<grid, 256, lds, st>>>(a);
    return hipGetLastError() == hipSuccess ? AZK_OK : AZK_ERR_HIP;
}

}  // namespace

extern "C" int32_t azk_nn_patch_embed(const void *boards_dev, int32_t boards_are_f32, const void *wt_bf16_dev,
                                      const float *cpos_dev, const float *ln_w_dev, const float *ln_b_dev,
                                      void *x_out_bf16_dev, void *xhat_out_bf16_dev, int32_t n, int32_t channels,
                                      int32_t rows, int32_t cols, int32_t ksize, int32_t kp, int32_t embed_dim,
                                      float ln_eps, void *stream) {
    if (!boards_dev || !wt_bf16_dev || !cpos_dev || (!x_out_bf16_dev && !xhat_out_bf16_dev)) return AZK_ERR_ARG;
    if (xhat_out_bf16_dev && (!ln_w_dev || !ln_b_dev)) return AZK_ERR_ARG;
    if (n < 0 || channels < 1 || rows < 1 || cols < 1 || ksize < 1 || (ksize & 1) == 0) return AZK_ERR_ARG;
    if (kp < channels * ksize * ksize || kp % 16 != 0) return AZK_ERR_ARG;
    if (n == 0) return AZK_OK;
    EmbedArgs a;
    a.boards = boards_dev; a.boards_f32 = boards_are_f32; a.wt = (const __hip_bfloat16 *)wt_bf16_dev; a.cpos = cpos_dev;
    a.ln_w = ln_w_dev; a.ln_b = ln_b_dev; a.x = (__hip_bfloat16 *)x_out_bf16_dev; a.xhat = (__hip_bfloat16 *)xhat_out_bf16_dev;
    a.n = n; a.C = channels; a.R = rows; a.Cc = cols; a.ksz = ksize; a.KP = kp; a.T = rows * cols + 1; a.eps = ln_eps;
    hipStream_t st = (hipStream_t)stream;
    const int ks = kp / 16;
#define CASE(NT_, KS_) if (embed_dim == 128 * NT_ && ks == KS_) return launch_embed<NT_, KS_>(a, st)
    CASE(4, 4); CASE(4, 5); CASE(4, 2); CASE(4, 1);
    CASE(2, 4); CASE(2, 5); CASE(2, 2); CASE(2, 1);
    CASE(1, 4); CASE(1, 5); CASE(1, 2); CASE(1, 1);
#undef CASE
    return AZK_ERR_ARG;   // unsupported (embed_dim, kp): the caller keeps its generic path
}
