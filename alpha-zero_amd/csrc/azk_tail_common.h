// azk_tail_common.h - what the two forms of the cls-row tail's GEMM links share (azk_nn.hip k_tail_gemm: whole K in registers;
// azk_tail.hip k_tail_lds: LDS-staged wide links): the argument block, the epilogue selectors and the epilogue arithmetic, so that
// a link computes the same values whichever kernel runs it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace azk_tail {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

struct TailArgs {
    const unsigned short *A; int lda, a_batch;
    const uint4 *Wp; long long w_batch;          // uint4 elements between batches
    int M, N, nbatch;                            // N = output columns per batch (multiple of 64 * NWC)
    const int *count;
    const float *bias;                           // [nbatch * N] or null
    unsigned short *out; int ldo;
    const unsigned short *resid; int ldr;
    float ln_eps;
    const float *stats_in; int stats_groups;     // LayerNorm of A: [M][stats_groups][2] partial (sum, sum of squares) of every A row, written by the producer
    float *stats_out;                            // optional: this GEMM's own partials [M][nbatch * N / 64][2] of the bf16-rounded output rows
    float *logits, *values; int action_dim;
    int wave_slots;                              // k_tail_gemm: waves of the launched instantiation the device holds at once (set by launch_tail)
    const float *csum;                           // k_tail_lds, LayerNorm in the epilogue: [N] column sums of the (bf16) weight
};

enum { TAIL_EPI_BF16 = 0, TAIL_EPI_GELU = 1, TAIL_EPI_RESID = 2, TAIL_EPI_HEADS = 3 };

// nn.GELU (erf form) with erf from Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below the bf16 result's resolution):
// a dozen instructions instead of libm's erff.
__device__ __forceinline__ float gelu_erf(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float erf_abs = 1.0f - poly * __expf(-z * z);
    return 0.5f * x * (1.0f + copysignf(erf_abs, x));
}

// Sum over the 16 lanes of a DPP row (lanes sharing lane>>4), result in every lane: quad_perm [1,0,3,2], quad_perm
// [2,3,0,1], row_half_mirror, row_mirror - four v_add_f32 with a DPP operand instead of four ds_bpermute round trips.
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    return v;
}

// Sum over the 4 lanes of a quad, result in every lane of it.
__device__ __forceinline__ float quad_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    return v;
}

}  // namespace azk_tail
