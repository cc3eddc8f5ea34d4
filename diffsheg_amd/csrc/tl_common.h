// Shared pieces of the token-per-lane kernels (tl_linear.hip, tl2.hip): vector types, LDS stage geometry, bf16 helpers.
#pragma once
#include <type_traits>
#include <utility>

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

// LayerNorm (+ folded FiLM + SiLU) of a wave's 32 rows held as packed bf16 B fragments, in place (tl_linear.hip prologue)
template <int NFRAG, bool FILM_SILU>
__device__ __forceinline__ void ln_frags(u32x4 (&frag)[NFRAG], const float* ca, const float* cb, float kn, float kfull) {
    float sum, sq;
    row_moments_bf16<NFRAG>(frag, sum, sq);
    const float mean = sum / kn;
    sq = fmaxf(sq - sum * mean, 0.f);            // sum (x - mean)^2; zero-padded columns add nothing to either moment
    (void)kfull;
    const float rstd = 1.0f / sqrtf(sq / kn + 1e-5f);
    const float nmr = -mean * rstd;
    f32x4 pa[2][2], pb[2][2];
#pragma unroll
    for (int q = 0; q < 2; ++q) { pa[0][q] = *reinterpret_cast<const f32x4*>(ca + 4 * q); pb[0][q] = *reinterpret_cast<const f32x4*>(cb + 4 * q); }
#pragma unroll
    for (int s = 0; s < NFRAG; ++s) {
        if (s + 1 < NFRAG) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                pa[(s + 1) & 1][q] = *reinterpret_cast<const f32x4*>(ca + 16 * (s + 1) + 4 * q);
                pb[(s + 1) & 1][q] = *reinterpret_cast<const f32x4*>(cb + 16 * (s + 1) + 4 * q);
            }
        }
        float v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[2 * j] = bf_lo(frag[s][j]); v[2 * j + 1] = bf_hi(frag[s][j]); }
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = fmaf(v[4 * q + e], rstd, nmr);
                const float y = fmaf(t, pa[s & 1][q][e], pb[s & 1][q][e]);
                v[4 * q + e] = FILM_SILU ? y * __builtin_amdgcn_rcpf(1.0f + __expf(-y)) : y;
            }
#pragma unroll
        for (int j = 0; j < 4; ++j) frag[s][j] = pack_bf16(v[2 * j], v[2 * j + 1]);
        __builtin_amdgcn_sched_barrier(0);
    }
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

// ---- residual stream as two bf16 planes (round 4) -----------------------------------------------------------------------------
// h = hi + lo with hi = bf16(h) (round to nearest even: it IS the MFMA operand the next Linear reads, the old "h16 shadow") and
// lo = bf16(h - hi): 16 - 17 mantissa bits, relative error <= 2^-17 per store.  A producer writes 2 + 2 bytes per value instead
// of 4 (fp32) + 2 (shadow); a residual reader loads hi + lo = the 4 bytes it loaded before.  Both planes use the tiled bf16
// layout: fragment s = 2 t + c of accumulator tile t, dword j = the value pair (2 j, 2 j + 1) of this lane.
// v_dot2c_f32_bf16 with a {1, 0} / {0, 1} selector adds one half of a packed pair to an fp32 value in ONE instruction.
typedef __bf16 tl_bf16x2 __attribute__((ext_vector_type(2)));
// (The selector goes through an opaque SGPR: hipcc (ROCm 7.2) encodes the packed constant 0x00003f80 of a v_dot2c_f32_bf16 as the
//  INLINE constant 1.0, which the instruction reads as the fp32 pattern 0x3f800000 — the HIGH half — so that every even element
//  silently took its odd neighbour's value (found by the all-rows op test, round 4).  A literal or register operand is read as is.)
// Non-finite values: the unselected half of the pair is multiplied by 0, so an Inf / NaN in ONE element of a packed pair makes its
// pair neighbour NaN as well (0 * Inf), in the hi and in the lo plane — finite data is unaffected, an overflow spreads to exactly one
// neighbour per step through the residual and is a little harder to localise (pinned by test_hilo_nonfinite_residual_stays_in_its_pair).
__device__ __forceinline__ uint32_t hl_selector(uint32_t bits) { asm("" : "+s"(bits)); return bits; }
__device__ __forceinline__ float hl_add_half(float acc, uint32_t w, int half) {
    const tl_bf16x2 sel = __builtin_bit_cast(tl_bf16x2, hl_selector(half ? 0x3f800000u : 0x00003f80u));
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(tl_bf16x2, w), sel, acc, false);
}
__device__ __forceinline__ float hl_sub_half(float acc, uint32_t w, int half) {
    const tl_bf16x2 sel = __builtin_bit_cast(tl_bf16x2, hl_selector(half ? 0xbf800000u : 0x0000bf80u));
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(tl_bf16x2, w), sel, acc, false);
}
// v[8] (one fragment's values of this lane) += hi + lo
__device__ __forceinline__ void hl_accumulate(float* v, const u32x4& hi, const u32x4& lo) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t hw = hi[j], lw = lo[j];
        v[2 * j] = hl_add_half(hl_add_half(v[2 * j], hw, 0), lw, 0);
        v[2 * j + 1] = hl_add_half(hl_add_half(v[2 * j + 1], hw, 1), lw, 1);
    }
}
// v[8] -> (hi, lo) fragments
__device__ __forceinline__ void hl_split(const float* v, u32x4& hi, u32x4& lo) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t hw = pack_bf16(v[2 * j], v[2 * j + 1]);
        hi[j] = hw;
        lo[j] = pack_bf16(hl_sub_half(v[2 * j], hw, 0), hl_sub_half(v[2 * j + 1], hw, 1));
    }
}

// 16 bytes per lane to base (SGPR pair) + voff + IMM, by a store hipcc does not see: with a store in its scoreboard hipcc waits vmcnt(0)
// in front of the first use of every later load ("loads and stores complete out of order"), which drains the weight DMA
// queue once per tile.  The data registers are read by the instruction itself; the trailing s_nop 1 covers the wait states before
// hipcc's next instruction may overwrite them.  (The hardware's vmcnt DOES count these stores: a counted wait also waits for every
// older store — keep them far in front of the next counted wait.)
template <int IMM, typename V4>
__device__ __forceinline__ void asm_store16(void* base, unsigned voff, const V4& v) {
    asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3\n\ts_nop 1" : : "v"(voff), "v"(v), "s"(base), "n"(IMM) : "memory");
}

// f(integral_constant<int, 0>{}), ..., f(integral_constant<int, N - 1>{}): a loop that is unrolled by construction.  "#pragma unroll" is a
// request: past hipcc's size threshold it is declined silently, and a register array indexed by the loop variable then lives in scratch
// (round 5: a 32-slot main loop with a long epilogue body became a real loop with its 64 activation fragments in scratch).
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{}); }

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// One KB of weights global -> LDS, asynchronously (LDS-DMA): lane L moves 16 B to lds_wave + 16 L.  Piece k (0..7, a compile-time
// constant after unrolling) of a wave's share of a chunk: four consecutive KBs share one M0 value through the instruction's
// immediate offset (it applies to the global AND the LDS address).  MUBUF form — buffer_load_dwordx4 ... lds with the stream's
// buffer descriptor in SGPRs, ONE per-lane offset register (lane's position inside a chunk share, loop invariant) and the chunk
// offset in an SGPR.  Round 2 used global_load_lds with 64-bit per-lane addresses; round-3 microbenchmark
// (scripts/micro/tl_loop_bench.hip, profiles/r03_tl_loop_microbench_*.log): a wave alone on its SIMD pays ~29 cycles of issue per
// global_load_lds piece and ~8 per buffer piece, and hipcc models global_load_lds as a FLAT access that may touch LDS: it then
// waits lgkmcnt(0) before every MFMA group of the phase instead of the exact count.
__device__ __forceinline__ void dma_buf(int k, __amdgpu_buffer_rsrc_t rsrc, int voff, int soff, char* lds_wave) {
    char* d4 = lds_wave + (k >> 2) * 4096;
    const int so = soff + (k >> 2) * 4096;
    switch (k & 3) {
        case 0: __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)d4, 16, voff, so, 0, 0); break;
        case 1: __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)d4, 16, voff, so, 1024, 0); break;
        case 2: __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)d4, 16, voff, so, 2048, 0); break;
        default: __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)d4, 16, voff, so, 3072, 0); break;
    }
}

// First-round start stagger.  A launch's first 256 blocks start together, one per CU, and — all blocks taking the same time — the
// whole chip then moves through load / compute / store phases in lockstep: the memory phases of every CU collide at ~11 B/clk/CU
// (5.6 TB/s chip-wide) while HBM idles during the compute phases (round-4 timeline: the fused FFN block spends 70 k of its 190 k
// cycles moving 768 KB that way).  Delaying group g = b % groups of the first round by g * sleep * 8 k cycles spreads the phases;
// later blocks inherit the offset of the CU they start on.
__device__ __forceinline__ void start_stagger(int groups, int sleep) {
    if (groups > 1 && blockIdx.y == 0 && blockIdx.x < 256) {
        const int n = (int)(blockIdx.x % (unsigned)groups) * sleep;
        for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(127);
    }
}

// Token-block index of this workgroup.  rev = 1 walks the rows in DESCENDING order: consecutive launches of a layer alternate
// direction (denoiser.hip), so that a kernel starts on the rows its producer wrote LAST — the ones still in the 256 MB Infinity
// Cache / the L2s — instead of the ones written first and long evicted (every activation tensor is 170 - 680 MB per launch).
__device__ __forceinline__ int tl_block_index(int rev) { return rev ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x; }

// block-timeline trace (bench only): {t_start, t_main, t_end (100 MHz ticks), blockIdx.x | xcc << 32}
__device__ __forceinline__ void trace_mark(unsigned long long* tr, int slot) {
    if (tr && threadIdx.x == 0) {
        unsigned long long* r = tr + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4;
        r[slot] = wall_clock64();
        if (slot == 0) {
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            r[3] = (unsigned long long)blockIdx.x | ((unsigned long long)(xcc & 0xf) << 32);
        }
    }
}

}  // namespace dsh
