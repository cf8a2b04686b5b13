// azk_tail.hip - the two WIDE links of the cls-row tail (ai/nn.py:58-60 for the one row the heads read: LayerNorm2 -> Linear(D, 4D)
// -> GELU, and Linear(4D, D) + residual) as LDS-staged MFMA GEMMs for M ~ 256-2048 live rows.
//
// What the register-only form of these links (azk_nn.hip k_tail_gemm) pays for: every wave pulls its own A rows as fragment-shaped
// loads (16 rows x 64 B per instruction, half cache lines) and its own weight fragments, ~100 KB per wave through the CU's vector
// memory path, with 442-466 registers per wave (one wave per SIMD).  Here a workgroup of eight waves shares one tile:
//   * the activation tile AND the weight tile go to LDS by LDS-DMA (global_load_lds_dwordx4: no registers, no ds_write pass), the
//     activations in full 128-byte lines: one wave instruction fills 8 rows x 128 B of an XOR-swizzled [rows][8 x 16 B] image
//     (the swizzle is applied to the per-lane SOURCE address, the LDS image stays lane-linear), so the A fragments come back as
//     conflict-free ds_read_b128;  the weights are already stored in fragment order (pack_linear_weight: 1 KiB per fragment tile),
//     so their image is a plain copy and a B fragment is a lane-linear ds_read_b128;
//   * K runs in 64-column stages through a ring of NS buffers filled NS-1 stages ahead; a stage is retired by a COUNTED
//     s_waitcnt vmcnt(N) (never 0 inside the loop) followed by ONE raw s_barrier per stage - the barrier that publishes stage t also
//     frees the buffer of stage t-1 for the next fill;
//   * ~60 registers per wave: two waves per SIMD, so one wave's LDS reads and epilogue arithmetic run under the other's MFMAs.
// K = 2048 splits K over four wave groups of the workgroup (each an independent accumulation chain over its 512 columns, the
// chains added in LDS in the fixed order 0,1,2,3: bit for bit k_tail_gemm's result for the same inputs).
// LayerNorm of the A rows (link 3) is applied in the EPILOGUE: LN(x) W'^T = rstd (x W'^T - mean colsum(W')) - the matrix pipe
// multiplies the stored bf16 rows as they are (no per-fragment normalise + re-round pass), the row statistics come from the
// producing GEMM's partial sums as before.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "azk.h"
#include "azk_launch.h"
#include "azk_tail_common.h"

namespace {

using namespace azk_tail;

template <int N> __device__ __forceinline__ void wait_vmcnt_c() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
// (the count is a compile-time constant after unrolling; the asm immediate wants an integer constant expression)
__device__ __forceinline__ void wait_vmcnt(int n) {
    switch (n) {
#define AZK_W(N_) case N_: wait_vmcnt_c<N_>(); break;
        AZK_W(0) AZK_W(1) AZK_W(2) AZK_W(3) AZK_W(4) AZK_W(5) AZK_W(6) AZK_W(7) AZK_W(8) AZK_W(9) AZK_W(10) AZK_W(11) AZK_W(12)
        AZK_W(13) AZK_W(14) AZK_W(15) AZK_W(16) AZK_W(17) AZK_W(18) AZK_W(19) AZK_W(20) AZK_W(21) AZK_W(22) AZK_W(23) AZK_W(24)
#undef AZK_W
        default: wait_vmcnt_c<0>(); break;
    }
}

// One LDS-DMA piece: 64 lanes x 16 bytes from per-lane global addresses to 1 KiB of LDS at the wave-uniform byte address lds_dst.
// As inline assembly on purpose: hipcc treats the builtin form as a pending LDS write and drains it with s_waitcnt vmcnt(0) in front of
// the next ds_read - every stage of the ring would be waited for at once.  The statement saves and restores M0 (the destination base);
// the loads are invisible to the compiler's own counters, so every wait for them below is explicit.
__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// One tiling of a link.  WM x WN x WK waves; a wave owns 16 RT rows x 64 columns (RT A fragments, four B fragments per 32-wide
// k-step) and runs NCH accumulation chains one after the other, each over KTC stages of 64 columns: wave group wk covers the
// chains wk NCH .. wk NCH + NCH - 1 of the K range.  Block tile: 16 RT WM rows x 64 WN columns, K = 64 KTC NCH WK.  The chains are
// added in the order 0, 1, 2, ... whatever the tiling (first a wave's own, then the other groups' through LDS).
// PL = operand planes: 1 = bf16 (azk_nn_tail_gemm_lds); 2 = fp16 (hi, lo) planes of the fp32-accurate tail (azk_nnx_gemm_h_lds):
// activations as two [M][lda] fp16 arrays, weights in pack_linear_weight_h's order [N/64][K/32][4][2 planes][64 lanes][8].
template <int WM_, int WN_, int WK_, int NCH_, int KTC_, int NS_, int RT_ = 1, int PL_ = 1>
struct TailTiling {
    static constexpr int WM = WM_, WN = WN_, WK = WK_, NCH = NCH_, KTC = KTC_, NS = NS_, RT = RT_, PL = PL_;
    static constexpr int NW = WM * WN * WK, BM = 16 * RT * WM, BN = 64 * WN, KT = NCH * KTC, KS = 2 * KT * WK;   // KT: stages per wave; KS: 32-wide k-steps of the whole K
    static constexpr int XB = PL * WK * BM * 128, WB = PL * WK * WN * 8192, SB = XB + WB;     // bytes per stage
    static constexpr int LDS_BYTES = NS * SB;
    static constexpr int NXI = XB / 1024, NI = SB / 1024, LPS = NI / NW, LPX = NXI / NW;     // LDS-DMA wave instructions per stage: x part / all / per wave / a wave's x pieces
    static_assert(XB % 1024 == 0 && NI % NW == 0 && BM % 8 == 0 && NXI % NW == 0, "whole 1 KiB wave pieces, evenly over the waves");
    static_assert(NS >= 2 && NS - 1 <= KT && (NS - 1) * LPS <= 24, "ring depth");
    static_assert(WK == 1 || (WK - 1) * WM * WN * NCH * RT * 4096 <= NS * SB, "the split-K partials reuse the stage ring");
};

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

// The main loop shared by both operand forms: stages the block's activation rows (planes A0, A1) and weight groups through the ring
// and returns the wave's chains in accs.  row0: first row of the block; g0: its first 64-column weight group.
template <class T>
__device__ __forceinline__ void lds_mainloop(const char *A0, const char *A1, const size_t lda_bytes, const int M, const uint4 *Wp, const int row0,
                                             const int g0, char *const lds, f32x4 (&accs)[T::NCH][T::RT][4]) {
    constexpr int WM = T::WM, WN = T::WN, WK = T::WK, NCH = T::NCH, KTC = T::KTC, NS = T::NS, RT = T::RT, PL = T::PL;
    constexpr int NW = T::NW, BM = T::BM, KT = T::KT, KS = T::KS, XB = T::XB, SB = T::SB, NXI = T::NXI, LPS = T::LPS, LPX = T::LPX;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    const int wk = wave / (WM * WN), wmn = wave - wk * (WM * WN), wm = wmn / WN, wn = wmn - wm * WN;
    // ---- the wave's LDS-DMA pieces: piece j = wave + NW i of a stage lands at byte 1024 j of the stage buffer ----
    // x pieces (j < NXI): j = ((plane WK + kq) BM / 8 + r8): rows 8 r8 .. + 8 of wave group kq's columns; lane l fills physical 16-byte
    //   chunk l & 7 of row l >> 3 with the row's LOGICAL chunk (l & 7) ^ (l >> 3)  (rows are 128 B; the read applies the same XOR)
    // w pieces: j' = j - NXI = ((kq WN + g) 8 PL + q): the q-th KiB of column group g0 + g's 8 PL KiB for this stage (two k-steps x four
    //   tiles x PL planes, in the packed order)
    const char *gp[LPS];
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)lds + (unsigned)wave * 1024u);
#pragma unroll
    for (int i = 0; i < LPS; i++) {
        const int j = wave + NW * i;
        if (i < LPX) {
            const int pk = j / (BM / 8), pl = pk / WK, kq = pk - pl * WK, r = 8 * (j % (BM / 8)) + (lane >> 3);
            gp[i] = (pl ? A1 : A0) + (size_t)min(row0 + r, M - 1) * lda_bytes + ((size_t)kq * (64 * KT) + 8 * ((lane & 7) ^ (lane >> 3))) * 2;
        } else {
            const int jw = j - NXI, kq = jw / (8 * PL * WN), g = (jw / (8 * PL)) % WN, q = jw % (8 * PL);
            gp[i] = (const char *)(Wp + ((size_t)(g0 + g) * KS + (size_t)kq * 2 * KT) * (256 * PL) + q * 64 + lane);
        }
    }
    auto stage = [&](int t) {                                            // stage t of every wave group -> ring buffer t % NS
        const unsigned dst = lds_base + (t % NS) * SB;
#pragma unroll
        for (int i = 0; i < LPS; i++) glds16(gp[i] + (size_t)t * (i < LPX ? 128 : 8192 * PL), dst + NW * 1024 * i);
    };
#pragma unroll
    for (int t = 0; t < NS - 1; t++) stage(t);
    // fragment addresses inside a stage buffer
    int arow[RT], a_off[RT];
#pragma unroll
    for (int i = 0; i < RT; i++) {
        arow[i] = 16 * (wm * RT + i) + l15;
        a_off[i] = (wk * BM + arow[i]) * 128;                           // + plane * WK * BM * 128 + ((kk * 4 + l4) ^ (arow & 7)) * 16
    }
    const int b_off = XB + (wk * WN + wn) * 8192 * PL + lane * 16;      // + ((kk * 4 + c) * PL + plane) * 1024
#pragma unroll
    for (int ch = 0; ch < NCH; ch++)
#pragma unroll
        for (int i = 0; i < RT; i++)
#pragma unroll
            for (int c = 0; c < 4; c++) accs[ch][i][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < KT; t++) {
        // stages t+1 .. min(t+NS-2, KT-1) may stay in flight; stage t of THIS wave is complete after the wait, of every wave after the barrier
        const int younger = (t + NS - 2 < KT - 1 ? t + NS - 2 : KT - 1) - t;
        wait_vmcnt(younger * LPS);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // this wave's fragment reads of stage t-1 have left the buffer the barrier frees
        __builtin_amdgcn_s_barrier();
        if (t + NS - 1 < KT) stage(t + NS - 1);                          // into the buffer stage t-1 was read from: every wave is past those reads
        const char *const sb = lds + (t % NS) * SB;
        uint4 af[2][RT][PL], bf[2][4][PL];
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
#pragma unroll
            for (int i = 0; i < RT; i++)
#pragma unroll
                for (int p = 0; p < PL; p++) af[kk][i][p] = *(const uint4 *)(sb + p * (WK * BM * 128) + a_off[i] + (((kk * 4 + l4) ^ (arow[i] & 7)) << 4));
#pragma unroll
            for (int c = 0; c < 4; c++)
#pragma unroll
                for (int p = 0; p < PL; p++) bf[kk][c][p] = *(const uint4 *)(sb + b_off + ((kk * 4 + c) * PL + p) * 1024);
        }
        __builtin_amdgcn_sched_barrier(0);                               // every fragment read of the stage in flight before its first MFMA
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
            if constexpr (PL == 1) {
#pragma unroll
                for (int i = 0; i < RT; i++)
#pragma unroll
                    for (int c = 0; c < 4; c++)
                        accs[t / KTC][i][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[kk][i][0]), __builtin_bit_cast(bf16x8, bf[kk][c][0]),
                                                                                      accs[t / KTC][i][c], 0, 0, 0);
            } else {
                // hi hi, hi lo, lo hi - term by term over the accumulators (k_gemm_h's order: an accumulator sees the same sequence)
#pragma unroll
                for (int term = 0; term < 3; term++)
#pragma unroll
                    for (int i = 0; i < RT; i++)
#pragma unroll
                        for (int c = 0; c < 4; c++)
                            accs[t / KTC][i][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, af[kk][i][term == 2 ? 1 : 0]),
                                                                                         __builtin_bit_cast(f16x8, bf[kk][c][term == 1 ? 1 : 0]), accs[t / KTC][i][c], 0, 0, 0);
            }
        }
        // the stage's MFMAs stay in the stage: being register-only instructions they may otherwise be sunk below the following stages'
        // waits and barriers - towards the epilogue, their first use - with every stage's fragments kept alive for them (244 VGPRs, spills
        // in the two-plane form).  An empty statement that "modifies" the accumulators pins their values here.
#pragma unroll
        for (int i = 0; i < RT; i++)
#pragma unroll
            for (int c = 0; c < 4; c++) asm volatile("" : "+v"(accs[t / KTC][i][c]));
        __builtin_amdgcn_sched_barrier(0);
    }
}

// the chains added in the order 0, 1, 2, ...: a wave group's own first, then the other groups' through LDS; false: this wave is done
template <class T>
__device__ __forceinline__ bool lds_reduce_chains(f32x4 (&accs)[T::NCH][T::RT][4], f32x4 (&acc)[T::RT][4], char *const lds) {
    constexpr int WM = T::WM, WN = T::WN, WK = T::WK, NCH = T::NCH, RT = T::RT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wk = wave / (WM * WN), wmn = wave - wk * (WM * WN);
#pragma unroll
    for (int i = 0; i < RT; i++)
#pragma unroll
        for (int c = 0; c < 4; c++) acc[i][c] = accs[0][i][c];
    if (wk == 0) {
#pragma unroll
        for (int ch = 1; ch < NCH; ch++)
#pragma unroll
            for (int i = 0; i < RT; i++)
#pragma unroll
                for (int c = 0; c < 4; c++) acc[i][c] += accs[ch][i][c];
    }
    if (WK > 1) {
        __builtin_amdgcn_s_barrier();                                    // every wave has read its last stage; no LDS-DMA is outstanding (vmcnt(0) in the loop)
        f32x4 *const kred = (f32x4 *)lds;
        if (wk > 0) {
#pragma unroll
            for (int ch = 0; ch < NCH; ch++)
#pragma unroll
                for (int i = 0; i < RT; i++)
#pragma unroll
                    for (int c = 0; c < 4; c++) kred[(((((wk - 1) * WM * WN + wmn) * NCH + ch) * RT + i) * 4 + c) * 64 + lane] = accs[ch][i][c];
        }
        __syncthreads();
        if (wk > 0) return false;
#pragma unroll
        for (int w = 1; w < WK; w++)
#pragma unroll
            for (int ch = 0; ch < NCH; ch++)
#pragma unroll
                for (int i = 0; i < RT; i++)
#pragma unroll
                    for (int c = 0; c < 4; c++) acc[i][c] += kred[(((((w - 1) * WM * WN + wmn) * NCH + ch) * RT + i) * 4 + c) * 64 + lane];
    }
    return true;
}

template <class T, int AMODE, int EPI, bool AZK_XCD_ROWS>
__device__ __forceinline__ void tail_lds_body(const TailArgs &a, const int nvalid, char *const lds) {
    constexpr int WM = T::WM, WN = T::WN, RT = T::RT, BM = T::BM, BN = T::BN, KT = T::KT, WK = T::WK;
    static_assert(T::PL == 1, "bf16 operands");
    const int ctiles = a.N / BN;
    // block -> tile.  Blocks b and b + 8 share an XCD (round-robin dispatch; speed only): with XR the XCD index picks the ROW tile (row tiles
    // x, x + 8, .. and every column tile on XCD x), so that an XCD's L2 pulls an eighth of the activation rows and the whole weight once -
    // the K = 2 048 links, whose activations are the bigger operand; otherwise the column tile varies fastest (an XCD sees two column
    // tiles of the weight and every row)
    const bool XR = AZK_XCD_ROWS && (ctiles & 7) == 0;
    const int ct = XR ? (int)(blockIdx.x >> 3) % ctiles : (int)blockIdx.x % ctiles;
    const int rt = XR ? (int)(blockIdx.x & 7) + 8 * ((int)(blockIdx.x >> 3) / ctiles) : (int)blockIdx.x / ctiles;
    const int row0 = rt * BM;
    if (row0 >= nvalid) return;                                          // (uniform per workgroup; nothing has been issued yet)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    const int wk = wave / (WM * WN), wmn = wave - wk * (WM * WN), wm = wmn / WN, wn = wmn - wm * WN;
    const int g0 = ct * WN;                                              // first 64-column group of the block

    // ---- epilogue operands that do not depend on the product: requested first, consumed last ----
    const int colg = 64 * (g0 + wn) + 4 * l15;                           // a lane's four accumulators of a row are four consecutive output columns
    f32x4 bv = {0.f, 0.f, 0.f, 0.f}, cs = {0.f, 0.f, 0.f, 0.f};
    f32x4 st[RT];
    uint2 rr[EPI == TAIL_EPI_RESID ? RT : 1][4];
#pragma unroll
    for (int i = 0; i < RT; i++) st[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (wk == 0) {
        if (a.bias) bv = *(const f32x4 *)(a.bias + colg);
        if (AMODE == 2) {
            cs = *(const f32x4 *)(a.csum + colg);
            // the statistics of the lane group's four rows: lane l15 fetches quarter l15 & 3 (two column groups) of row l15 >> 2
#pragma unroll
            for (int i = 0; i < RT; i++)
                st[i] = *((const f32x4 *)(a.stats_in + (size_t)min(row0 + 16 * (wm * RT + i) + 4 * l4 + (l15 >> 2), a.M - 1) * 16) + (l15 & 3));
        }
        if (EPI == TAIL_EPI_RESID) {
#pragma unroll
            for (int i = 0; i < RT; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) rr[i][j] = *(const uint2 *)(a.resid + (size_t)min(row0 + 16 * (wm * RT + i) + 4 * l4 + j, a.M - 1) * a.ldr + colg);
        }
    }
    __builtin_amdgcn_sched_barrier(0);

    f32x4 accs[T::NCH][RT][4], accr[RT][4];
    lds_mainloop<T>((const char *)a.A, (const char *)a.A, (size_t)a.lda * 2, a.M, a.Wp, row0, g0, lds, accs);
    if (!lds_reduce_chains<T>(accs, accr, lds)) return;

#pragma unroll
    for (int ri = 0; ri < RT; ri++) {
    f32x4 (&acc)[4] = accr[ri];
    const f32x4 st_i = st[ri];
    const uint2 (&rr_i)[4] = rr[EPI == TAIL_EPI_RESID ? ri : 0];
    const int rbase = row0 + 16 * (wm * RT + ri) + 4 * l4;
    // ---- epilogue (the arithmetic of k_tail_gemm's, RT = 1) ----
    float rstd[4], mrs[4];
    if (AMODE == 2) {
        // eight (sum, sum of squares) pairs per row, added in a fixed order: the lane's two groups, then the quad's four lanes
        const float q1 = quad_sum(st_i[0] + st_i[2]), q2 = quad_sum(st_i[1] + st_i[3]);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float s1 = __shfl(q1, (lane & 48) + 4 * j), s2 = __shfl(q2, (lane & 48) + 4 * j);
            constexpr float invk = 1.0f / (64 * KT * WK);
            const float mean = s1 * invk;
            rstd[j] = rsqrtf(fmaxf(s2 * invk - mean * mean, 0.f) + a.ln_eps);
            mrs[j] = mean * rstd[j];
        }
    }
    uint2 o[4];
    f32x2 ps[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        f32x4 v;
        if (AMODE == 2) v = f32x4{acc[0][j] * rstd[j] - mrs[j] * cs[0] + bv[0], acc[1][j] * rstd[j] - mrs[j] * cs[1] + bv[1],
                                  acc[2][j] * rstd[j] - mrs[j] * cs[2] + bv[2], acc[3][j] * rstd[j] - mrs[j] * cs[3] + bv[3]};
        else v = f32x4{acc[0][j] + bv[0], acc[1][j] + bv[1], acc[2][j] + bv[2], acc[3][j] + bv[3]};
        if (EPI == TAIL_EPI_GELU) {
#pragma unroll
            for (int c = 0; c < 4; c++) v[c] = gelu_erf(v[c]);                                      // nn.GELU (erf form)
        }
        if (EPI == TAIL_EPI_RESID) {
            v[0] += __uint_as_float(rr_i[j].x << 16); v[1] += __uint_as_float(rr_i[j].x & 0xffff0000u);
            v[2] += __uint_as_float(rr_i[j].y << 16); v[3] += __uint_as_float(rr_i[j].y & 0xffff0000u);
        }
        union { bf16x4 b4; uint2 u; } ob;
        ob.b4 = __builtin_convertvector(v, bf16x4);
        o[j] = ob.u;
        if (a.stats_out) {
            const f32x4 vr = __builtin_convertvector(ob.b4, f32x4);
            ps[j] = f32x2{row16_sum((vr[0] + vr[1]) + (vr[2] + vr[3])),
                          row16_sum((vr[0] * vr[0] + vr[1] * vr[1]) + (vr[2] * vr[2] + vr[3] * vr[3]))};
        }
    }
    const int ngr = a.N >> 6, gr = g0 + wn;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int row = rbase + j;
        if (row < nvalid) *(uint2 *)(a.out + (size_t)row * a.ldo + colg) = o[j];
        if (a.stats_out && l15 == 0 && row < nvalid) *(f32x2 *)(a.stats_out + ((size_t)row * ngr + gr) * 2) = ps[j];
    }
    }
}

// Two tilings in one kernel: T1 while the live rows fit one round of workgroups (rows <= SWITCH), T2 (taller tiles, fewer workgroups)
// above - the choice is made in the kernel from the device-side live count, so a captured launch serves every step.  SWITCH = 0: T1 only.
template <class T1, class T2, int SWITCH, int AMODE, int EPI>
__global__ __launch_bounds__(64 * T1::NW, 2) void k_tail_lds(TailArgs a) {
    static_assert(T1::NW == T2::NW, "one workgroup shape");
    extern __shared__ uint4 smem[];
    // (every kernel argument is wanted in SGPRs here, before the count's own round trip: see k_tail_gemm)
    asm volatile("" :: "s"(a.A), "s"(a.Wp), "s"(a.out), "s"(a.bias), "s"(a.resid), "s"(a.stats_in), "s"(a.stats_out), "s"(a.csum),
                 "s"(a.lda), "s"(a.ldo), "s"(a.N), "s"(a.M), "s"(a.ldr), "s"(a.ln_eps));
    const int nvalid = a.count ? min(a.M, *a.count) : a.M;
    if (SWITCH > 0 && nvalid > SWITCH) tail_lds_body<T2, AMODE, EPI, EPI == TAIL_EPI_RESID>(a, nvalid, (char *)smem);
    else tail_lds_body<T1, AMODE, EPI, EPI == TAIL_EPI_RESID>(a, nvalid, (char *)smem);
}

template <class T1, class T2, int SWITCH, int AMODE, int EPI>
int launch_lds(TailArgs &a, hipStream_t st) {
    constexpr size_t lds_bytes = T1::LDS_BYTES > T2::LDS_BYTES ? T1::LDS_BYTES : T2::LDS_BYTES;
    auto kern = k_tail_lds<T1, T2, SWITCH, AMODE, EPI>;
    if (azk_set_max_lds((const void *)kern, (int)lds_bytes) != hipSuccess) return AZK_ERR_HIP;
    constexpr int RND = EPI == TAIL_EPI_RESID ? 8 : 1;                   // (the XCD-row mapping walks the row tiles in groups of eight)
    const unsigned b1 = (unsigned)(((a.M + T1::BM - 1) / T1::BM + RND - 1) / RND * RND * (a.N / T1::BN)), b2 = (unsigned)(((a.M + T2::BM - 1) / T2::BM + RND - 1) / RND * RND * (a.N / T2::BN));
    kern<<<b1 > b2 ? b1 : b2, 64 * T1::NW, lds_bytes, st>>>(a);
    return hipGetLastError() == hipSuccess ? AZK_OK : AZK_ERR_HIP;
}

// ---- the fp32-accurate tail's wide links (azk_nnx_gemm_h_lds): fp16 (hi, lo) operand planes, the arithmetic of azk_nnx.hip k_gemm_h ----
struct HArgs {
    const char *Ahi, *Alo; int lda;
    const uint4 *Wp;
    int M, N;
    const int *count;
    const float *bias, *csum;
    float inv_scale, a_scale;
    _Float16 *ohi, *olo; float *of32; int ldo;
    const float *resid; int ldr;
    float ln_eps;
    const float *stats_in; float *stats_out;
    int *oflow;
};

// nn.GELU (erf form), erf by Abramowitz & Stegun 7.1.26 - k_gemm_h's expression (a true division, not the reciprocal instruction)
__device__ __forceinline__ float gelu_as(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = 1.0f / (1.0f + 0.3275911f * z);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float erf_abs = 1.0f - poly * __expf(-z * z);
    return 0.5f * x * (1.0f + copysignf(erf_abs, x));
}

template <class T, int LNA, int EPI, bool AZK_XCD_ROWS>
__device__ __forceinline__ void gemm_h_lds_body(const HArgs &a, const int nvalid, char *const lds) {
    constexpr int WM = T::WM, WN = T::WN, RT = T::RT, BM = T::BM, BN = T::BN;
    static_assert(T::PL == 2, "(hi, lo) fp16 planes");
    bool oflow_seen = false;                                             // a plane value left fp16's range: reported once, at the end (a.oflow, sticky)
    if ((int)(threadIdx.x >> 6) >= T::NW) return;                       // a tiling with fewer waves than the launch carries: the surplus waves leave before any barrier
    const int ctiles = a.N / BN;
    // block -> tile.  Blocks b and b + 8 share an XCD (round-robin dispatch; speed only): with XR the XCD index picks the ROW tile (row tiles
    // x, x + 8, .. and every column tile on XCD x), so that an XCD's L2 pulls an eighth of the activation rows and the whole weight once -
    // the K = 2 048 links, whose activations are the bigger operand; otherwise the column tile varies fastest (an XCD sees two column
    // tiles of the weight and every row)
    const bool XR = AZK_XCD_ROWS && (ctiles & 7) == 0;
    const int ct = XR ? (int)(blockIdx.x >> 3) % ctiles : (int)blockIdx.x % ctiles;
    const int rt = XR ? (int)(blockIdx.x & 7) + 8 * ((int)(blockIdx.x >> 3) / ctiles) : (int)blockIdx.x / ctiles;
    const int row0 = rt * BM;
    if (row0 >= nvalid) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    const int wk = wave / (WM * WN), wmn = wave - wk * (WM * WN), wm = wmn / WN, wn = wmn - wm * WN;
    const int g0 = ct * WN;
    const int col0 = 64 * (g0 + wn) + 4 * l15;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f}, cs = {0.f, 0.f, 0.f, 0.f};
    f32x4 st[RT], rr[EPI == TAIL_EPI_RESID ? RT : 1][4];
#pragma unroll
    for (int i = 0; i < RT; i++) st[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (wk == 0) {
        if (a.bias) bv = *(const f32x4 *)(a.bias + col0);
        if (LNA) {
            cs = *(const f32x4 *)(a.csum + col0);
#pragma unroll
            for (int i = 0; i < RT; i++)
                st[i] = *((const f32x4 *)(a.stats_in + (size_t)min(row0 + 16 * (wm * RT + i) + 4 * l4 + (l15 >> 2), a.M - 1) * 16) + (l15 & 3));
        }
        if (EPI == TAIL_EPI_RESID) {
#pragma unroll
            for (int i = 0; i < RT; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) rr[i][j] = *(const f32x4 *)(a.resid + (size_t)min(row0 + 16 * (wm * RT + i) + 4 * l4 + j, a.M - 1) * a.ldr + col0);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x4 accs[T::NCH][RT][4], accr[RT][4];
    lds_mainloop<T>(a.Ahi, a.Alo, (size_t)a.lda * 2, a.M, a.Wp, row0, g0, lds, accs);
    if (!lds_reduce_chains<T>(accs, accr, lds)) return;
    const int ngr = a.N >> 6, gr = g0 + wn;
#pragma unroll
    for (int i = 0; i < RT; i++) {
        float rstd[4], mshift[4];
        if (LNA) {
            // the row's eight (sum, sum of squares) pairs in k_gemm_h's association: ((g0 + g1) + (g2 + g3)) + ((g4 + g5) + (g6 + g7))
            const float q1 = quad_sum(st[i][0] + st[i][2]), q2 = quad_sum(st[i][1] + st[i][3]);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float s1 = __shfl(q1, (lane & 48) + 4 * j), s2 = __shfl(q2, (lane & 48) + 4 * j);
                const float mean = s1 * (1.0f / 512.0f);
                rstd[j] = 1.0f / sqrtf(fmaxf(__builtin_fmaf(-mean, mean, s2 * (1.0f / 512.0f)), 0.f) + a.ln_eps);
                mshift[j] = mean;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int row = row0 + 16 * (wm * RT + i) + 4 * l4 + j;
            f32x4 v = {accr[i][0][j], accr[i][1][j], accr[i][2][j], accr[i][3][j]};
            v = v * a.inv_scale;
            if (LNA) v = (v - cs * mshift[j]) * rstd[j];
            v += bv;
            if (EPI == TAIL_EPI_GELU) {
#pragma unroll
                for (int c = 0; c < 4; c++) v[c] = gelu_as(v[c]);
            }
            if (EPI == TAIL_EPI_RESID) v += rr[i][j];
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

template <class T1, class T2, int SWITCH, int LNA, int EPI>
__global__ __launch_bounds__(64 * (T1::NW > T2::NW ? T1::NW : T2::NW), 2) void k_gemm_h_lds(HArgs a) {
    extern __shared__ uint4 smem[];
    asm volatile("" :: "s"(a.Ahi), "s"(a.Alo), "s"(a.Wp), "s"(a.bias), "s"(a.csum), "s"(a.ohi), "s"(a.olo), "s"(a.of32), "s"(a.resid), "s"(a.stats_in),
                 "s"(a.stats_out), "s"(a.lda), "s"(a.ldo), "s"(a.ldr), "s"(a.N), "s"(a.M), "s"(a.inv_scale), "s"(a.a_scale), "s"(a.ln_eps));
    const int nvalid = a.count ? min(a.M, *a.count) : a.M;
    if (SWITCH > 0 && nvalid > SWITCH) gemm_h_lds_body<T2, LNA, EPI, EPI == TAIL_EPI_RESID>(a, nvalid, (char *)smem);
    else gemm_h_lds_body<T1, LNA, EPI, EPI == TAIL_EPI_RESID>(a, nvalid, (char *)smem);
}

template <class T1, class T2, int SWITCH, int LNA, int EPI>
int launch_h_lds(HArgs &a, hipStream_t st) {
    constexpr size_t lds_bytes = T1::LDS_BYTES > T2::LDS_BYTES ? T1::LDS_BYTES : T2::LDS_BYTES;
    auto kern = k_gemm_h_lds<T1, T2, SWITCH, LNA, EPI>;
    if (azk_set_max_lds((const void *)kern, (int)lds_bytes) != hipSuccess) return AZK_ERR_HIP;
    constexpr int RND = EPI == TAIL_EPI_RESID ? 8 : 1;
    const unsigned b1 = (unsigned)(((a.M + T1::BM - 1) / T1::BM + RND - 1) / RND * RND * (a.N / T1::BN)), b2 = (unsigned)(((a.M + T2::BM - 1) / T2::BM + RND - 1) / RND * RND * (a.N / T2::BN));
    kern<<<b1 > b2 ? b1 : b2, 64 * (T1::NW > T2::NW ? T1::NW : T2::NW), lds_bytes, st>>>(a);
    return hipGetLastError() == hipSuccess ? AZK_OK : AZK_ERR_HIP;
}

}  // namespace

// azk_nn_tail_lds_footprint(1): the K = 2048 link keeps two ring buffers instead of three (96 KB of LDS instead of 144): for callers that
// step several game groups on separate streams, so that another group's tree waves (14 KB of LDS each) fit beside a tail workgroup
static int g_tail_small_lds = 0;
extern "C" int32_t azk_nn_tail_lds_footprint(int32_t small) { g_tail_small_lds = small ? 1 : 0; return AZK_OK; }

// azk_nn_tail_gemm_lds: the LDS-staged form of azk_nn_tail_gemm for its two wide links (include/azk.h).
extern "C" int32_t azk_nn_tail_gemm_lds(const azk_tail_gemm *t, void *stream) {
    if (!t || !t->a_bf16 || !t->w_packed || t->m < 0 || t->nbatch != 1 || !t->out_bf16) return AZK_ERR_ARG;
    if ((t->lda & 7) || t->lda < t->k || ((uintptr_t)t->a_bf16 & 15) || (t->ldo & 3) || t->ldo < t->n_out) return AZK_ERR_ARG;
    if (t->m == 0) return AZK_OK;
    TailArgs a = {};
    a.A = (const unsigned short *)t->a_bf16; a.lda = t->lda; a.a_batch = 0; a.Wp = (const uint4 *)t->w_packed; a.w_batch = 0;
    a.M = t->m; a.N = t->n_out; a.nbatch = 1; a.count = t->n_valid; a.bias = t->bias; a.out = (unsigned short *)t->out_bf16; a.ldo = t->ldo;
    a.resid = (const unsigned short *)t->resid_bf16; a.ldr = t->ldr; a.ln_eps = t->ln_eps; a.stats_in = t->a_stats; a.stats_groups = t->a_stats_groups;
    a.stats_out = t->stats_out; a.csum = t->a_col_sums;
    hipStream_t st = (hipStream_t)stream;
    if (t->k == 512 && t->epilogue == TAIL_EPI_GELU && t->layernorm_a && t->n_out % 128 == 0) {
        if (!t->a_stats || t->a_stats_groups != 8 || !t->a_col_sums || t->stats_out) return AZK_ERR_ARG;
        // 64 x 128 tiles, three stages of 24 KB: two workgroups per CU, so one round of workgroups up to 2048 live rows
        using T = TailTiling<4, 2, 1, 1, 8, 3>;
        return launch_lds<T, T, 0, 2, TAIL_EPI_GELU>(a, st);
    }
    if (t->k == 2048 && t->epilogue == TAIL_EPI_RESID && !t->layernorm_a && t->n_out % 64 == 0) {
        if (!t->resid_bf16 || (t->ldr & 3)) return AZK_ERR_ARG;
        // up to 1024 live rows: 32 x 64 tiles, four wave groups x one chain (144 KB of LDS: one workgroup per CU, 8 x rows / 32 of them);
        // above: 64 x 64 tiles, two wave groups x two chains, so that the launch stays one round of workgroups
        if (g_tail_small_lds) return launch_lds<TailTiling<2, 1, 4, 1, 8, 2>, TailTiling<4, 1, 2, 2, 8, 2>, 1024, 0, TAIL_EPI_RESID>(a, st);   // two ring buffers: 96 / 64 KB
        return launch_lds<TailTiling<2, 1, 4, 1, 8, 3>, TailTiling<4, 1, 2, 2, 8, 3>, 1024, 0, TAIL_EPI_RESID>(a, st);
    }
    return AZK_ERR_ARG;
}

// azk_nnx_gemm_h_lds: the LDS-staged form of azk_nnx_gemm_h for the fp32-accurate tail's two wide links (include/azk.h): the same
// accumulation chains and epilogue arithmetic, operands through LDS like azk_nn_tail_gemm_lds.
extern "C" int32_t azk_nnx_gemm_h_lds(const azk_gemm_h *t, void *stream) {
    if (!t || !t->w_packed || !t->a_hi || !t->a_lo || t->a_f32 || t->m < 0 || t->nbatch != 1) return AZK_ERR_ARG;
    if ((t->lda & 7) || t->lda < t->k || ((uintptr_t)t->a_hi & 15) || ((uintptr_t)t->a_lo & 15)) return AZK_ERR_ARG;
    if ((!t->out_f32 && !(t->out_hi && t->out_lo)) || (t->out_hi != nullptr) != (t->out_lo != nullptr) || t->ldo < t->n_out || (t->ldo & 3)) return AZK_ERR_ARG;
    if (!(t->a_scale > 0.f) || !(t->w_scale > 0.f)) return AZK_ERR_ARG;
    if (t->m == 0) return AZK_OK;
    HArgs a = {};
    a.Ahi = (const char *)t->a_hi; a.Alo = (const char *)t->a_lo; a.lda = t->lda; a.Wp = (const uint4 *)t->w_packed;
    a.M = t->m; a.N = t->n_out; a.count = t->n_valid; a.bias = t->bias; a.csum = t->col_sums;
    a.inv_scale = 1.0f / (t->a_scale * t->w_scale); a.a_scale = t->a_scale;
    a.ohi = (_Float16 *)t->out_hi; a.olo = (_Float16 *)t->out_lo; a.of32 = t->out_f32; a.ldo = t->ldo; a.resid = t->resid_f32; a.ldr = t->ldr;
    a.ln_eps = t->ln_eps; a.stats_in = t->a_stats; a.stats_out = t->stats_out; a.oflow = t->overflow_flag;
    hipStream_t st = (hipStream_t)stream;
    if (t->k == 512 && t->epilogue == TAIL_EPI_GELU && t->layernorm_a && t->n_out % 128 == 0) {
        if (!t->a_stats || !t->col_sums || t->stats_out) return AZK_ERR_ARG;
        // up to 1024 live rows: 64 x 128 tiles (48 KB per stage, three stages); above: 128 x 128 tiles, two A fragments per wave
        return launch_h_lds<TailTiling<4, 2, 1, 1, 8, 3, 1, 2>, TailTiling<4, 2, 1, 1, 8, 2, 2, 2>, 1024, 1, TAIL_EPI_GELU>(a, st);
    }
    if (t->k == 2048 && t->epilogue == TAIL_EPI_RESID && !t->layernorm_a && t->n_out % 64 == 0) {
        if (!t->resid_f32 || (t->ldr & 3)) return AZK_ERR_ARG;
        // up to 1024 live rows: 32 x 64 tiles on FOUR waves (two wave groups x two chains; 48 KB per stage, two stages) - twice the workgroups
        // of the 64 x 64 form, half the matrix-pipe time per SIMD; above: 64 x 64 tiles on eight waves, one round of workgroups up to 2048 rows
        using TA = TailTiling<2, 1, 2, 2, 8, 2, 1, 2>;
        using TB = TailTiling<4, 1, 2, 2, 8, 2, 1, 2>;
        return launch_h_lds<TA, TB, 1024, 0, TAIL_EPI_RESID>(a, st);
    }
    return AZK_ERR_ARG;
}
