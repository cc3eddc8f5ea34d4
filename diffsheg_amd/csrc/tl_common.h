// Shared pieces of the token-per-lane kernels (tl_linear.hip, tl2.hip): vector types, LDS stage geometry, bf16 helpers.
#pragma once
#include "dsh_common.h"

namespace dsh {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int TL_TOK = 128;                 // tokens per block (4 waves x 32)
constexpr int TL_STAGE_K = 256;             // k' per LDS stage
constexpr int TL_ROW = TL_STAGE_K * 2 + 16; // padded stage row (528 B): conflict-free ds_read_b128
constexpr int TL_STAGE = 32 * TL_ROW;       // 16,896 B
constexpr int TL_MAXCLIP = 6;               // FiLM prologue: clips whose folded rows a 128-token block stages in LDS
// the whole 32-feature W tile (KD/256 stages) is double buffered in LDS -> one block barrier per tile
constexpr int tl_lds_bytes(int kd) { return 2 * (kd / TL_STAGE_K) * TL_STAGE; }

__device__ __forceinline__ float bf_lo(uint32_t v) { return __builtin_bit_cast(float, v << 16); }
__device__ __forceinline__ float bf_hi(uint32_t v) { return __builtin_bit_cast(float, v & 0xffff0000u); }
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) { return pack_bf16_pair(lo, hi); }

// Row moments of NFRAG packed-bf16 fragments (this lane's half of a token row) with v_dot2c_f32_bf16: two exact products +
// fp32 accumulate per instruction, i.e. one instruction per element for (sum, sum of squares) instead of five for an unpack /
// centre / square pass (2.5 us of every 128-token block's prologue at K = 512, 5 us at K = 1024).  Returns the sums over the
// WHOLE row (both lanes of the token).  var = E[x^2] - mean^2 in fp32: the cancellation error is ~2e-6 (1 + mean^2 / var)
// relative, far below the bf16 operand rounding; zero-padded columns add nothing to either sum.
template <int NFRAG, typename FragT>
__device__ __forceinline__ void row_moments_bf16(const FragT (&frag)[NFRAG], float& sum, float& sumsq) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const bf16x2_t ones = __builtin_bit_cast(bf16x2_t, 0x3f803f80u);
    float sm[2] = {0.f, 0.f}, sq[2] = {0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NFRAG; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t w = frag[s][j];                      // scalar copy first (vector-element bit_cast pitfall)
            const bf16x2_t v = __builtin_bit_cast(bf16x2_t, w);
            sm[j & 1] = __builtin_amdgcn_fdot2_f32_bf16(v, ones, sm[j & 1], false);
            sq[j & 1] = __builtin_amdgcn_fdot2_f32_bf16(v, v, sq[j & 1], false);
        }
    sum = sm[0] + sm[1];
    sumsq = sq[0] + sq[1];
    sum += __shfl_xor(sum, 32, 64);
    sumsq += __shfl_xor(sumsq, 32, 64);
}

// GELU(x) = x Phi(x) for the bf16 epilogue of the 512 -> 1024 FFN GEMM, without transcendentals: the erf-form epilogue
// (rcp + exp, quarter-rate ops) made that kernel VALU-bound (2 waves/SIMD x 16 values/tile).  Phi(x) - 1/2 is odd:
// Phi(x) ~ 1/2 + xc h(xc^2), xc = clamp(x, -4.25, 4.25), h a degree-7 minimax polynomial constrained to h(4.25^2) = 1/(2*4.25)
// so that the tails are exact.  Max |error| vs the exact erf form 9.5e-5 (fit + fp32 Horner, checked on [-10, 10]) — 40x
// below the bf16 resolution of the stored result for |x| >= 1; all ops are plain FMAs (v_pk_fma_f32 pairs).
__device__ __forceinline__ float gelu_fast(float x) {
    const float xc = __builtin_amdgcn_fmed3f(x, -4.25f, 4.25f);
    const float u = xc * xc;
    float h = -8.460346867522617e-10f;
    h = fmaf(h, u, 7.570786664246043e-08f);
    h = fmaf(h, u, -2.938788611572818e-06f);
    h = fmaf(h, u, 6.552687409566715e-05f);
    h = fmaf(h, u, -0.0009404457523487508f);
    h = fmaf(h, u, 0.009257814846932888f);
    h = fmaf(h, u, -0.06545348465442657f);
    h = fmaf(h, u, 0.3984200358390808f);
    return x * fmaf(xc, h, 0.5f);
}

}  // namespace dsh
