// azk_block.hip - the FULL-TOKEN transformer block of the policy-value network (ai/nn.py:38-61) for networks deeper than one block:
// every block but the last runs over all T = R C + 1 tokens of every board (the last block only feeds the cls row: azk_nn.hip).
//   x -> LayerNorm1 (azk_nn_layernorm_rows) -> QKV = LN1(x) Wi^T + bi (azk_nn_gemm_tok) -> softmax(Q K^T / sqrt(dh)) V per board and
//   head (azk_nn_attention_tok) -> x += O Wo^T + bo (gemm, residual epilogue) -> LayerNorm2 -> GELU(. W0^T + b0) (gemm, GELU epilogue)
//   -> x += . W3^T + b3 (gemm, residual epilogue)
// Two kernels:
//   k_gemm_tok   C[M][N] = A[M][K] W^T (+ bias) through an epilogue, M = boards x tokens (10^4 .. 10^5 rows) or a few hundred cls rows:
//                the LDS-staged structure of azk_tail.hip (activation rows and fragment-packed weights to LDS by LDS-DMA in full 128-byte
//                lines, XOR-swizzled image, ring of K stages retired by counted waits, eight waves per workgroup) with a run-time K loop.
//   k_attn_tok   one workgroup per (board, head): K and V^T of the head staged in LDS once, every wave takes 16-query tiles;
//                S^T = K Q^T on v_mfma_f32_16x16x32_bf16 so that a query's scores sit in ONE lane column (softmax = in-register
//                reduction + two cross-lane steps), and the probabilities are already the B operand of O^T = V^T P^T: no LDS round trip
//                for P (V^T is stored with its keys in the accumulator's row order, so the k-slots of the two operands agree).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "azk.h"
#include "azk_launch.h"
#include "azk_tail_common.h"

namespace {

using namespace azk_tail;

template <int N> __device__ __forceinline__ void wait_vmcnt_c() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
__device__ __forceinline__ void wait_vmcnt(int n) {          // (wave-uniform n: a scalar branch tree)
    switch (n) {
#define AZK_W(N_) case N_: wait_vmcnt_c<N_>(); break;
        AZK_W(0) AZK_W(1) AZK_W(2) AZK_W(3) AZK_W(4) AZK_W(5) AZK_W(6) AZK_W(7) AZK_W(8) AZK_W(9) AZK_W(10) AZK_W(11) AZK_W(12)
        AZK_W(13) AZK_W(14) AZK_W(15) AZK_W(16)
#undef AZK_W
        default: wait_vmcnt_c<0>(); break;
    }
}

// one LDS-DMA piece (see azk_tail.hip glds16: inline assembly so that the compiler does not drain it in front of every ds_read)
__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

struct GemmTokArgs {
    const unsigned short *A; int lda;
    const uint4 *Wp;
    int M, N, KT;                                // KT = K / 64 stages
    const int *count;
    const float *bias;
    void *out; int ldo;
    const unsigned short *resid; int ldr;
};

enum { TOK_EPI_BF16 = 0, TOK_EPI_GELU = 1, TOK_EPI_RESID = 2, TOK_EPI_F32 = 4 };

// 4 (M) x 2 (N) waves, a wave owns 16 RT rows x 64 columns; block tile 64 RT x 128; stages of 64 K columns; NS ring buffers.
template <int RT, int NS, int EPI>
__global__ __launch_bounds__(512, 2) void k_gemm_tok(GemmTokArgs a) {
    constexpr int WM = 4, WN = 2, NW = 8, BM = 16 * RT * WM, BN = 128;
    constexpr int XB = BM * 128, WB = WN * 8192, SB = XB + WB, NXI = XB / 1024, NI = SB / 1024, LPS = NI / NW, LPX = NXI / NW;
    static_assert(NXI % NW == 0 && NI % NW == 0 && (NS - 1) * LPS <= 16, "pieces");
    extern __shared__ uint4 smem[];
    char *const lds = (char *)smem;
    asm volatile("" :: "s"(a.A), "s"(a.Wp), "s"(a.out), "s"(a.bias), "s"(a.resid), "s"(a.lda), "s"(a.ldo), "s"(a.N), "s"(a.M), "s"(a.ldr), "s"(a.KT));
    const int nvalid = a.count ? min(a.M, *a.count) : a.M;
    const int ctiles = a.N / BN;
    const int ct = blockIdx.x % ctiles, rt = blockIdx.x / ctiles;
    const int row0 = rt * BM;
    if (row0 >= nvalid) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int g0 = ct * WN, KT = a.KT, KS = 2 * KT;
    const char *gp[LPS];
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)lds + (unsigned)wave * 1024u);
#pragma unroll
    for (int i = 0; i < LPS; i++) {
        const int j = wave + NW * i;
        if (i < LPX) {
            const int r = 8 * j + (lane >> 3);
            gp[i] = (const char *)(a.A + (size_t)min(row0 + r, a.M - 1) * a.lda + 8 * ((lane & 7) ^ (lane >> 3)));
        } else {
            const int jw = j - NXI, g = jw >> 3, q = jw & 7;
            gp[i] = (const char *)(a.Wp + (size_t)(g0 + g) * KS * 256 + q * 64 + lane);
        }
    }
    auto stage = [&](int t, int buf) {
        const unsigned dst = lds_base + (unsigned)__builtin_amdgcn_readfirstlane(buf) * SB;
#pragma unroll
        for (int i = 0; i < LPS; i++) glds16(gp[i] + (size_t)t * (i < LPX ? 128 : 8192), dst + NW * 1024 * i);
    };
    for (int t = 0; t < NS - 1 && t < KT; t++) stage(t, t);
    int arow[RT], a_off[RT];
#pragma unroll
    for (int i = 0; i < RT; i++) { arow[i] = 16 * (wm * RT + i) + l15; a_off[i] = arow[i] * 128; }
    const int b_off = XB + wn * 8192 + lane * 16;
    f32x4 acc[RT][4];
#pragma unroll
    for (int i = 0; i < RT; i++)
#pragma unroll
        for (int c = 0; c < 4; c++) acc[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    int buf = 0, nbuf = (NS - 1) % NS;
#pragma unroll 1
    for (int t = 0; t < KT; t++) {
        const int younger = min(NS - 2, KT - 1 - t);
        wait_vmcnt(younger * LPS);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t + NS - 1 < KT) stage(t + NS - 1, nbuf);
        const char *const sb = lds + buf * SB;
        uint4 af[2][RT], bf[2][4];
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
#pragma unroll
            for (int i = 0; i < RT; i++) af[kk][i] = *(const uint4 *)(sb + a_off[i] + (((kk * 4 + l4) ^ (arow[i] & 7)) << 4));
#pragma unroll
            for (int c = 0; c < 4; c++) bf[kk][c] = *(const uint4 *)(sb + b_off + (kk * 4 + c) * 1024);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 2; kk++)
#pragma unroll
            for (int i = 0; i < RT; i++)
#pragma unroll
                for (int c = 0; c < 4; c++)
                    acc[i][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[kk][i]), __builtin_bit_cast(bf16x8, bf[kk][c]), acc[i][c], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < RT; i++)
#pragma unroll
            for (int c = 0; c < 4; c++) asm volatile("" : "+v"(acc[i][c]));
        __builtin_amdgcn_sched_barrier(0);
        buf = buf + 1 == NS ? 0 : buf + 1;
        nbuf = nbuf + 1 == NS ? 0 : nbuf + 1;
    }
    // ---- epilogue ----
    const int colg = 64 * (g0 + wn) + 4 * l15;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) bv = *(const f32x4 *)(a.bias + colg);
#pragma unroll
    for (int i = 0; i < RT; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int row = row0 + 16 * (wm * RT + i) + 4 * l4 + j;
            if (row >= nvalid) continue;
            f32x4 v = {acc[i][0][j] + bv[0], acc[i][1][j] + bv[1], acc[i][2][j] + bv[2], acc[i][3][j] + bv[3]};
            if (EPI == TOK_EPI_GELU) {
#pragma unroll
                for (int c = 0; c < 4; c++) v[c] = gelu_erf(v[c]);
            }
            if (EPI == TOK_EPI_RESID) {
                const uint2 rr = *(const uint2 *)(a.resid + (size_t)row * a.ldr + colg);
                v[0] += __uint_as_float(rr.x << 16); v[1] += __uint_as_float(rr.x & 0xffff0000u);
                v[2] += __uint_as_float(rr.y << 16); v[3] += __uint_as_float(rr.y & 0xffff0000u);
            }
            if (EPI == TOK_EPI_F32) {
                *(f32x4 *)((float *)a.out + (size_t)row * a.ldo + colg) = v;
            } else {
                union { bf16x4 b4; uint2 u; } ob;
                ob.b4 = __builtin_convertvector(v, bf16x4);
                *(uint2 *)((unsigned short *)a.out + (size_t)row * a.ldo + colg) = ob.u;
            }
        }
}

template <int RT, int NS, int EPI>
int launch_gemm_tok(const GemmTokArgs &a, hipStream_t st) {
    constexpr int BM = 64 * RT, SB = BM * 128 + 2 * 8192;
    constexpr int lds_bytes = NS * SB;
    auto kern = k_gemm_tok<RT, NS, EPI>;
    if (azk_set_max_lds((const void *)kern, lds_bytes) != hipSuccess) return AZK_ERR_HIP;
    const long long blocks = (long long)((a.M + BM - 1) / BM) * (a.N / 128);
    if (blocks > 0x7fffffffLL) return AZK_ERR_ARG;
    kern<<<(unsigned)blocks, 512, lds_bytes, st>>>(a);
    return hipGetLastError() == hipSuccess ? AZK_OK : AZK_ERR_HIP;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// attention over all tokens of a board (nn.MultiheadAttention in eval mode: softmax(q k^T / sqrt(dh)) v, nn.py:41,54-55)
// ---------------------------------------------------------------------------------------------------------------------------------
struct AttnTokArgs {
    const unsigned short *qkv;      // [n][T][3 D] bf16: q | k | v, head h at columns h dh .. of each third
    unsigned short *out;            // [n][T][D] bf16, head h at columns h dh ..
    const int *count;
    int n, T, D, H;
    float scale;                    // 1 / sqrt(dh)
};

template <int DH>                   // head dimension: 32 or 64
__global__ __launch_bounds__(256, 2) void k_attn_tok(AttnTokArgs a) {
    constexpr int TP = 256;                          // keys padded to 16 tiles of 16
    constexpr int KROW = DH * 2;                     // bytes per K row
    constexpr int KC = KROW / 16;                    // 16-byte chunks per K row: 8 (dh 64) or 4 (dh 32)
    __shared__ __attribute__((aligned(16))) unsigned char smem[TP * KROW + DH * TP * 2];
    unsigned char *const Kimg = smem;                // [256 keys][KROW], chunk c of row r stored at chunk c ^ swz(r)
    unsigned char *const VT = smem + TP * KROW;      // [DH][256 key slots] bf16: slot p of 32-group g = key 32 g + 16 (p >> 2 & 1 ... see below); chunk-swizzled by row
    const int nvalid = a.count ? min(a.n, *a.count) : a.n;
    const int b = blockIdx.x / a.H, h = blockIdx.x - b * a.H;
    if (b >= nvalid) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    const int T = a.T, ld = 3 * a.D;
    const unsigned short *base = a.qkv + (size_t)b * T * ld + h * DH;
    auto kswz = [](int r) { return KC == 8 ? (r & 7) : ((r >> 2) & 3); };
    // ---- stage K (row-major, swizzled chunks) and V^T (keys in accumulator-row order) ----
    for (int i = tid; i < TP * KC; i += 256) {
        const int r = i / KC, c = i - r * KC;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (r < T) v = *(const uint4 *)(base + (size_t)r * ld + a.D + 8 * c);
        *(uint4 *)(Kimg + r * KROW + ((c ^ kswz(r)) << 4)) = v;
    }
    // V^T: key k sits at slot p = 32 (k >> 5) + 8 ((k >> 2) & 3) + 4 ((k >> 4) & 1) + (k & 3): within a 32-key group the slot order is the
    // order in which a lane group holds the rows of two stacked 16-row accumulator tiles (rows 4 l4 + j of tile 0, then of tile 1)
    for (int i = tid; i < TP * (DH / 8); i += 256) {
        const int k = i / (DH / 8), c = i - k * (DH / 8);
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (k < T) v = *(const uint4 *)(base + (size_t)k * ld + 2 * a.D + 8 * c);
        const int p = 32 * (k >> 5) + 8 * ((k >> 2) & 3) + 4 * ((k >> 4) & 1) + (k & 3);
        const unsigned short e[8] = {(unsigned short)v.x, (unsigned short)(v.x >> 16), (unsigned short)v.y, (unsigned short)(v.y >> 16),
                                     (unsigned short)v.z, (unsigned short)(v.z >> 16), (unsigned short)v.w, (unsigned short)(v.w >> 16)};
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int d = 8 * c + q;                                      // row of V^T
            const int chunk = (p >> 3) ^ (d & 15);                        // 32 chunks of 8 slots per row, swizzled by the row
            *(unsigned short *)(VT + d * (TP * 2) + (chunk << 4) + ((p & 7) << 1)) = e[q];
        }
    }
    __syncthreads();
    const int nqt = (T + 15) >> 4;
    for (int qt = wave; qt < nqt; qt += 4) {
        const int q0 = 16 * qt;
        // Q fragments of the tile (B operand of S^T = K Q^T): lane (l4, l15) = Q[q0 + l15][8 l4 + 32 s ..]
        uint4 qf[DH / 32];
#pragma unroll
        for (int s = 0; s < DH / 32; s++) qf[s] = *(const uint4 *)(base + (size_t)min(q0 + l15, T - 1) * ld + 32 * s + 8 * l4);
        f32x4 S[16];
#pragma unroll
        for (int kt = 0; kt < 16; kt++) {
            S[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int kr = 16 * kt + l15;
#pragma unroll
            for (int s = 0; s < DH / 32; s++) {
                const uint4 kf = *(const uint4 *)(Kimg + kr * KROW + (((4 * s + l4) ^ kswz(kr)) << 4));
                S[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kf), __builtin_bit_cast(bf16x8, qf[s]), S[kt], 0, 0, 0);
            }
            if ((kt & 3) == 3) __builtin_amdgcn_sched_barrier(0);         // four key tiles' fragment reads in flight at a time (all sixteen: 128 registers, spills)
        }
        // S[kt][j] = score of key 16 kt + 4 l4 + j for query q0 + l15: softmax over the keys of a query = over (kt, j) and the four lane groups
        float mx = -3.0e38f;
#pragma unroll
        for (int kt = 0; kt < 16; kt++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int key = 16 * kt + 4 * l4 + j;
                S[kt][j] = key < T ? S[kt][j] * a.scale : -3.0e38f;
                mx = fmaxf(mx, S[kt][j]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 16; kt++)
#pragma unroll
            for (int j = 0; j < 4; j++) { S[kt][j] = __expf(S[kt][j] - mx); sum += S[kt][j]; }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float inv = 1.0f / sum;
        // O^T = V^T P^T: per 32-key group the B operand is the lane's eight probabilities of tiles 2 g and 2 g + 1, as they stand
        f32x4 O[DH / 16];
#pragma unroll
        for (int m = 0; m < DH / 16; m++) O[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 8; g++) {
            union { bf16x8 v; bf16x4 hlf[2]; } pb;
            pb.hlf[0] = __builtin_convertvector(S[2 * g], bf16x4);
            pb.hlf[1] = __builtin_convertvector(S[2 * g + 1], bf16x4);
#pragma unroll
            for (int m = 0; m < DH / 16; m++) {
                const int d = 16 * m + l15;
                const uint4 vf = *(const uint4 *)(VT + d * (TP * 2) + (((4 * g + l4) ^ (d & 15)) << 4));
                O[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, vf), pb.v, O[m], 0, 0, 0);
            }
            if (g & 1) __builtin_amdgcn_sched_barrier(0);
        }
        // O[m][j] = output dimension 16 m + 4 l4 + j of query q0 + l15
        if (q0 + l15 < T) {
            unsigned short *orow = a.out + ((size_t)b * T + q0 + l15) * a.D + h * DH;
#pragma unroll
            for (int m = 0; m < DH / 16; m++) {
                union { bf16x4 b4; uint2 u; } ob;
                ob.b4 = __builtin_convertvector(O[m] * inv, bf16x4);
                *(uint2 *)(orow + 16 * m + 4 * l4) = ob.u;
            }
        }
    }
}

}  // namespace

extern "C" int32_t azk_nn_gemm_tok(const azk_gemm_tok *t, void *stream) {
    if (!t || !t->a_bf16 || !t->w_packed || !t->out || t->m < 0) return AZK_ERR_ARG;
    if (t->k < 128 || (t->k & 63) || (t->n_out & 127) || t->n_out < 128 || t->lda < t->k || (t->lda & 7) || ((uintptr_t)t->a_bf16 & 15)) return AZK_ERR_ARG;
    if (t->ldo < t->n_out || (t->ldo & 3)) return AZK_ERR_ARG;
    if (t->epilogue == TOK_EPI_RESID && (!t->resid_bf16 || (t->ldr & 3))) return AZK_ERR_ARG;
    if (t->m == 0) return AZK_OK;
    GemmTokArgs a = {};
    a.A = (const unsigned short *)t->a_bf16; a.lda = t->lda; a.Wp = (const uint4 *)t->w_packed; a.M = t->m; a.N = t->n_out; a.KT = t->k / 64;
    a.count = t->n_valid; a.bias = t->bias; a.out = t->out; a.ldo = t->ldo; a.resid = (const unsigned short *)t->resid_bf16; a.ldr = t->ldr;
    hipStream_t st = (hipStream_t)stream;
    const bool big = t->m >= 8192;                     // token-level GEMMs: 128-row tiles; cls-row GEMMs: 64-row tiles
#define AZK_CASE(E_) if (t->epilogue == E_) return big ? launch_gemm_tok<2, 3, E_>(a, st) : launch_gemm_tok<1, 3, E_>(a, st)
    AZK_CASE(TOK_EPI_BF16); AZK_CASE(TOK_EPI_GELU); AZK_CASE(TOK_EPI_RESID); AZK_CASE(TOK_EPI_F32);
#undef AZK_CASE
    return AZK_ERR_ARG;
}

extern "C" int32_t azk_nn_attention_tok(const void *qkv_bf16_dev, void *out_bf16_dev, int32_t n_boards, int32_t tokens, int32_t embed_dim,
                                        int32_t num_heads, const int32_t *n_valid_dev, void *stream) {
    if (!qkv_bf16_dev || !out_bf16_dev || n_boards < 0 || tokens < 1 || tokens > 256 || num_heads < 1 || embed_dim % num_heads) return AZK_ERR_ARG;
    const int dh = embed_dim / num_heads;
    if ((dh != 32 && dh != 64) || (embed_dim & 7) || ((uintptr_t)qkv_bf16_dev & 15) || ((uintptr_t)out_bf16_dev & 7)) return AZK_ERR_ARG;
    if (n_boards == 0) return AZK_OK;
    AttnTokArgs a;
    a.qkv = (const unsigned short *)qkv_bf16_dev; a.out = (unsigned short *)out_bf16_dev; a.count = n_valid_dev;
    a.n = n_boards; a.T = tokens; a.D = embed_dim; a.H = num_heads; a.scale = 1.0f / sqrtf((float)dh);
    hipStream_t st = (hipStream_t)stream;
    const unsigned blocks = (unsigned)n_boards * (unsigned)num_heads;
    if (dh == 64) k_attn_tok<64><<<blocks, 256, 0, st>>>(a);
    else k_attn_tok<32><<<blocks, 256, 0, st>>>(a);
    return hipGetLastError() == hipSuccess ? AZK_OK : AZK_ERR_HIP;
}
