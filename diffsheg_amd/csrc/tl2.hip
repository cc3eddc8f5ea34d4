// Token-per-lane kernels, second generation: the weight operand reaches LDS by LDS-DMA (global_load_lds_dwordx4) from a
// FRAGMENT-ORDERED copy of the weights, and the FFN branch of a decoder layer is one kernel.
//
// Why (round-1 profile of tl_linear.hip): the MFMA-bound instantiations ran at 24-33 % of the matrix peak.
// With the activation operand stationary in registers, the only thing the main loop has to do besides MFMAs is to move W,
// and the first generation moved it global -> VGPR -> ds_write_b128 -> LDS: 16 global loads, 16 wide LDS stores (13 cycles
// of issue each) and two vmcnt waits per wave and 32-feature tile, plus 64 staging VGPRs.  Here:
//
//   * W is stored in HBM exactly as the LDS image the MFMA A-operand reads want ("fragment order"): for the 32-row tile nt and
//     the 16-wide k step s, one contiguous KB = 64 lanes x 16 B, lane L = (n = L & 31, h = L >> 5) holding
//     W'[32 nt + n][16 s + 8 h .. + 7]  (W' = rows pi-permuted inside the tile, tl_weight_src_row).  A tile is a contiguous
//     K * 64 bytes.  global_load_lds writes LDS linearly (M0 base + lane * 16), so the copy is a plain linear stream and the
//     A-fragment read is ds_read_b128 at base + s * 1024 + lane * 16: conflict free by construction, no padding.
//   * per tile and wave: K / 64 DMA instructions, issued in one burst right after the tile barrier together with the
//     residual loads of this tile and the (deferred) stores of the previous one; one counted wait (vmcnt(0)) at the END of
//     the tile, when everything issued at its start has long landed.  Between them the wave issues only MFMAs and
//     ds_read_b128.  No staging registers: K = 512 kernels fit 2 blocks / CU with room to spare.
//   * tl2_ffn_kernel: ffn.linear1 -> GELU -> ffn.linear2 -> StylizationBlock(LN, FiLM, SiLU, Linear) -> + h for a wave's
//     32 tokens without leaving the register file: the 1024-wide hidden is produced 32 features at a time and immediately
//     consumed as one K chunk of linear2, whose 16 accumulators (the wave's 32 x 512 output) stay resident; LayerNorm
//     statistics are taken from the fp32 accumulators.  HBM traffic per token drops from 13 KB (three launches) to 6 KB
//     (h16 in, fp32 h in/out, h16 out); models/transformer.py:169-181, :86-97.
//
// Layouts of the activation tensors (tiled bf16, lane-native fp32) are those of tl_linear.hip.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "dsh_common.h"
#include "dsh_kernels.h"
#include "tl_common.h"

namespace dsh {

namespace {

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// one KB of weights global -> LDS, asynchronously: lane L moves 16 B from src_lane (its own address) to lds_wave + 16 L
template <int OFF>
__device__ __forceinline__ void dma_kb(const char* src_lane, char* lds_wave) {
    __builtin_amdgcn_global_load_lds((gptr_t)(src_lane + OFF), (lptr_t)(lds_wave + OFF), 16, 0, 0);
}
// N KB, consecutive
template <int N>
__device__ __forceinline__ void dma_kbs(const char* src_lane, char* lds_wave) {
#pragma unroll
    for (int i = 0; i < N; ++i)
        __builtin_amdgcn_global_load_lds((gptr_t)(src_lane + i * 1024), (lptr_t)(lds_wave + i * 1024), 16, 0, 0);
}

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

// LayerNorm (+ folded FiLM + SiLU) of a wave's 32 rows held as packed bf16 B fragments, in place (tl_linear.hip prologue)
template <int NFRAG, bool FILM_SILU>
__device__ __forceinline__ void ln_frags(u32x4 (&frag)[NFRAG], const float* ca, const float* cb, float kn, float kfull) {
    float sum = 0.f;
#pragma unroll
    for (int s = 0; s < NFRAG; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) sum += bf_lo(frag[s][j]) + bf_hi(frag[s][j]);
    sum += __shfl_xor(sum, 32, 64);
    const float mean = sum / kn;
#pragma unroll
    for (int s = 0; s < NFRAG; ++s) asm volatile("" : "+v"(frag[s]));
    float sq = 0.f;
#pragma unroll
    for (int s = 0; s < NFRAG; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a = bf_lo(frag[s][j]) - mean, b = bf_hi(frag[s][j]) - mean;
            sq += a * a + b * b;
        }
    sq += __shfl_xor(sq, 32, 64);
#pragma unroll
    for (int s = 0; s < NFRAG; ++s) asm volatile("" : "+v"(frag[s]));
    sq -= (kfull - kn) * mean * mean;            // zero-padded columns each added (0 - mean)^2
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

// NF MFMAs of `acc` against fragments fr[0 .. NF): A fragments at lds + i KB, read 4 ahead (one group = 4 MFMAs = 128 cycles,
// about one ds_read_b128 latency); the issue order is pinned, hipcc otherwise re-serialises each read in front of its MFMA
template <int NF, int GS = 4>
__device__ __forceinline__ void mfma_run(f32x16& acc, const char* lds_lane, const u32x4* fr) {
    static_assert(NF % GS == 0, "whole fragment groups");
    u32x4 aw[2][GS];
#pragma unroll
    for (int i = 0; i < GS; ++i) aw[0][i] = *reinterpret_cast<const u32x4*>(lds_lane + i * 1024);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < NF / GS; ++g) {
        if (g + 1 < NF / GS) {
#pragma unroll
            for (int i = 0; i < GS; ++i) aw[(g + 1) & 1][i] = *reinterpret_cast<const u32x4*>(lds_lane + ((g + 1) * GS + i) * 1024);
        }
#pragma unroll
        for (int i = 0; i < GS; ++i)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aw[g & 1][i]), __builtin_bit_cast(bf16x8, fr[g * GS + i]), acc, 0, 0, 0);
    }
#pragma unroll
    for (int g = 0; g < NF / GS; ++g) {
        if (g + 1 < NF / GS) __builtin_amdgcn_sched_group_barrier(0x100, GS, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, GS, 0);
    }
}

}  // namespace

// =====================================================================================================================
// Fused Linear: out = epilogue(prologue(X) W^T); template parameters and TlArgs as tl_linear_kernel, W in fragment order.
// LDS: [2][KD * 64] W tiles | bias [N] | row_const [N]; prologue parameters overlay the second W tile until the loop starts.
template <int KD, int PRO, bool HAS_R, int OUT, int ACT>
__global__ __launch_bounds__(256, (KD == 512 ? 2 : 1)) void tl2_linear_kernel(TlArgs p) {
    constexpr int NFRAG = KD / 16, TILE = NFRAG * 1024;
    constexpr bool HAS_C = (PRO == 2 && HAS_R && ACT == ACT_NONE);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    trace_mark(p.trace, 0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ml = lane & 31, h = lane >> 5;
    const int tb = blockIdx.x * (TL_TOK / 32) + wave;          // 32-token block owned by this wave (rows are not bounds-checked)
    const int row = tb * 32 + ml;
    const int lane_off = ml * 32 + h * 16;
    const int NT = p.N / 32;
    const int nt0 = blockIdx.y * p.tiles_per_block;
    const int nt1 = (nt0 + p.tiles_per_block) < NT ? (nt0 + p.tiles_per_block) : NT;
    // this wave's quarter of a W tile: DMA source (per lane) and LDS destination (wave uniform)
    const char* wsrc = reinterpret_cast<const char*>(p.W) + wave * (TILE / 4) + lane * 16;
    char* wdst = smem + wave * (TILE / 4);
    auto dma_tile = [&](int nt) {       // tile nt -> slot nt & 1 (clamped: unconditional, the last prefetch re-reads the last tile)
        const int t = nt < nt1 ? nt : nt1 - 1;
        dma_kbs<NFRAG / 4>(wsrc + (size_t)t * TILE, wdst + (nt & 1) * TILE);
    };
    dma_tile(nt0);

    // ---- prologue parameters (PRO 1/3: gamma | beta;  PRO 2: folded FiLM rows A | B of this block's clips) ------------
    constexpr int NPRM = PRO == 2 ? TL_MAXCLIP : (PRO == 0 ? 1 : KD / 512);
    f32x4 prm[NPRM];
    int clip0 = 0;
    if (PRO == 2) {
        const int rb = blockIdx.x * TL_TOK, rrb = rb >= p.half_row0 ? rb - p.half_row0 : rb;
        clip0 = rrb / p.frames;
        const int nclip = (rrb + TL_TOK - 1) / p.frames - clip0 + 1;
#pragma unroll
        for (int c = 0; c < TL_MAXCLIP; ++c) {
            const int cc = c < nclip ? c : nclip - 1;
            prm[c] = *reinterpret_cast<const f32x4*>(p.film + (size_t)((clip0 + cc) % p.bmod) * p.film_ld + p.film_off + tid * 4);
        }
    } else if (PRO != 0) {
#pragma unroll
        for (int c = 0; c < NPRM; ++c) {
            const int f = 1024 * c + 4 * tid;
            prm[c] = *reinterpret_cast<const f32x4*>(f < KD ? p.gamma + f : p.beta + (f - KD));
        }
    }
    // ---- activation rows -> B fragments ---------------------------------------------------------------------------------
    u32x4 frag[NFRAG];
    if (PRO == 3) {
        const char* r0 = reinterpret_cast<const char*>(p.X) + (size_t)tb * (512 / 16) * 1024 + lane_off;
        const char* r1 = reinterpret_cast<const char*>(p.X1) + (size_t)tb * (256 / 16) * 1024 + lane_off;
        const char* r2 = reinterpret_cast<const char*>(p.X2) + (size_t)tb * (128 / 16) * 1024 + lane_off;
        const bool has3 = p.X3 != nullptr;
        const char* r3 = has3 ? reinterpret_cast<const char*>(p.X3) + (size_t)tb * (128 / 16) * 1024 + lane_off : r2;
#pragma unroll
        for (int s = 0; s < NFRAG; ++s) {
            const char* src = s < 32 ? r0 + s * 1024 : (s < 48 ? r1 + (s - 32) * 1024 : (s < 56 ? r2 + (s - 48) * 1024 : r3 + (s - 56) * 1024));
            u32x4 v = *reinterpret_cast<const u32x4*>(src);
            if (s >= 56 && !has3) { v[0] = 0; v[1] = 0; v[2] = 0; v[3] = 0; }
            frag[s] = v;
        }
    } else {
        const char* xr = reinterpret_cast<const char*>(p.X) + (size_t)tb * (p.ldx / 16) * 1024 + lane_off;
#pragma unroll
        for (int s = 0; s < NFRAG; ++s) frag[s] = *reinterpret_cast<const u32x4*>(xr + s * 1024);
    }
    float* sbias = reinterpret_cast<float*>(smem + 2 * TILE);
    float* sconst = sbias + p.N;
    for (int i = tid; i < p.N; i += 256) {
        sbias[i] = p.bias ? p.bias[i] : 0.f;
        sconst[i] = p.row_const ? p.row_const[i] : 0.f;
    }
    float* sprm = reinterpret_cast<float*>(smem + ((nt0 + 1) & 1) * TILE);          // the slot the loop's first DMA will overwrite
    if (PRO >= 1) {
#pragma unroll
        for (int c = 0; c < NPRM; ++c) *reinterpret_cast<f32x4*>(sprm + 1024 * c + 4 * tid) = prm[c];
    }
    __syncthreads();                                  // bias / parameter tables visible to the whole block
    if (PRO >= 1) {
        const float* ca = sprm + 8 * h;
        const float* cb = sprm + KD + 8 * h;
        if (PRO == 2) {
            const int rr = row >= p.half_row0 ? row - p.half_row0 : row;
            int ci = rr / p.frames - clip0;
            ci = ci < TL_MAXCLIP ? ci : TL_MAXCLIP - 1;
            ca = sprm + ci * 1024 + 8 * h;
            cb = ca + 512;
        }
        ln_frags<NFRAG, PRO == 2>(frag, ca, cb, PRO == 3 ? (float)p.kreal : (float)KD, (float)KD);
    }
#pragma unroll
    for (int s = 0; s < NFRAG; ++s) asm volatile("" ::"v"(frag[s]));
    trace_mark(p.trace, 1);

    // ---- main loop: one 32-feature tile per iteration ------------------------------------------------------------------
    char* Ctb = reinterpret_cast<char*>(p.Ct);
    const float const_on = (p.row_const != nullptr && row < p.n_const_rows) ? 1.0f : 0.0f;
    const char* lds_lane = smem + lane * 16;
    f32x16 prev;                                     // finished values of the previous tile, stored one tile later
#pragma unroll
    for (int e = 0; e < 16; ++e) prev[e] = 0.f;
    auto store_tile = [&](int nt, const f32x16& v) {
        const size_t fidx = (((size_t)tb * NT + nt) * 4 * 64 + lane) * 4;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                const int qi = 2 * c + qq;
                if (OUT & 5) {
                    f32x4 o; o.x = v[4 * qi]; o.y = v[4 * qi + 1]; o.z = v[4 * qi + 2]; o.w = v[4 * qi + 3];
                    if (OUT & 4) *reinterpret_cast<f32x4*>(p.Cf + (size_t)row * p.ldcf + nt * 32 + 16 * c + 8 * h + 4 * qq) = o;
                    else *reinterpret_cast<f32x4*>(p.Cf + fidx + qi * 256) = o;
                }
            }
            if (OUT & 2) {
                u32x4 o;
                o.x = pack_bf16(v[8 * c + 0], v[8 * c + 1]); o.y = pack_bf16(v[8 * c + 2], v[8 * c + 3]);
                o.z = pack_bf16(v[8 * c + 4], v[8 * c + 5]); o.w = pack_bf16(v[8 * c + 6], v[8 * c + 7]);
                *reinterpret_cast<u32x4*>(Ctb + ((size_t)tb * (2 * NT) + 2 * nt + c) * 1024 + lane_off) = o;
            }
        }
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wave's share of the first tile has landed
    for (int nt = nt0; nt < nt1; ++nt) {
        // every wave's share of tile nt is in LDS, and nobody reads tile nt - 1 (slot of tile nt + 1) any more
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (nt > nt0) store_tile(nt - 1, prev);
        f32x4 rres[4];
        if (HAS_R) {
            const size_t fidx = (((size_t)tb * NT + nt) * 4 * 64 + lane) * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) rres[q] = *reinterpret_cast<const f32x4*>(p.R + fidx + q * 256);
        }
        dma_tile(nt + 1);
        f32x16 acc;
#pragma unroll
        for (int qi = 0; qi < 4; ++qi) {
            const int col = nt * 32 + 16 * (qi >> 1) + 8 * h + 4 * (qi & 1);
            f32x4 b4 = *reinterpret_cast<const f32x4*>(sbias + col);
            if (HAS_C) {
                const f32x4 c4 = *reinterpret_cast<const f32x4*>(sconst + col);
#pragma unroll
                for (int e = 0; e < 4; ++e) b4[e] = fmaf(const_on, c4[e], b4[e]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[4 * qi + e] = b4[e];
        }
        mfma_run<NFRAG>(acc, lds_lane + (nt & 1) * TILE, frag);
#pragma unroll
        for (int qi = 0; qi < 4; ++qi)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = acc[4 * qi + e];
                if (ACT == ACT_GELU) v = gelu_fast(v);
                else if (ACT == ACT_SILU) v = v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
                if (HAS_R) v += rres[qi][e];
                prev[4 * qi + e] = v;
            }
        // everything this wave issued at the top of the tile (DMA of the next tile, stores of the previous one) is a whole
        // tile of MFMAs old by now
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    store_tile(nt1 - 1, prev);
    trace_mark(p.trace, 2);
}

// =====================================================================================================================
// FFN branch of a decoder layer for 128 tokens per block (one wave per SIMD, 32 tokens each):
//   g = GELU(h16 W1^T + b1); y2 = g W2^T + b2; h <- h + Linear3(SiLU(LN(y2) (1 + scale) + shift)) (+ next layer's CFG-null constant)
// Weight stream `Wffn` (built by tl2_pack_ffn): 32 KB chunks  q = 2 j: W1 tile j (fragment order, K = 512),  q = 2 j + 1: the
// K chunk [32 j, 32 j + 32) of W2 as 32 fragments (output tile ot, k step ks) at (2 ot + ks) KB,  q = 64 + t: W3 tile t.
// LDS: four 32 KB chunk slots (q & 3) | folded FiLM rows of up to FFN_MAXCLIP clips | b1 [1024] | b2, b3, row_const [512].
// Iteration j consumes chunks 2 j (GEMM1 of hidden tile j) and 2 j - 1 (GEMM2 of hidden tile j - 1, whose GELU ran beside
// GEMM1 of tile j), while chunks 2 j + 1 and 2 j + 2 are in flight.
constexpr int FFN_MAXCLIP = 3;                   // clips a 128-token block may span (frames >= 64)
constexpr int FFN_CH = 32 * 1024;
constexpr int FFN_LDS = 4 * FFN_CH + FFN_MAXCLIP * 4096 + (1024 + 3 * 512) * 4;
constexpr int FFN_NQ = 64 + 16;
constexpr int FFN_GS = 2;                        // A fragments read ahead per group (register budget: 128 + 256 + ... of 512)

__global__ __launch_bounds__(256, 1) void tl2_ffn_kernel(Tl2FfnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    trace_mark(p.trace, 0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ml = lane & 31, h = lane >> 5;
    const int tb = blockIdx.x * (TL_TOK / 32) + wave;
    const int row = tb * 32 + ml;
    const int lane_off = ml * 32 + h * 16;
    const char* wsrc = reinterpret_cast<const char*>(p.Wffn) + wave * (FFN_CH / 4) + lane * 16;
    char* wdst = smem + wave * (FFN_CH / 4);
    auto dma_chunk = [&](int q) {
        const int qq = q < FFN_NQ ? q : FFN_NQ - 1;
        dma_kbs<8>(wsrc + (size_t)qq * FFN_CH, wdst + (q & 3) * FFN_CH);
    };
    dma_chunk(0);
    // folded FiLM rows (A | B) of this block's clips
    f32x4 prm[FFN_MAXCLIP];
    int clip0;
    {
        const int rb = blockIdx.x * TL_TOK, rrb = rb >= p.half_row0 ? rb - p.half_row0 : rb;
        clip0 = rrb / p.frames;
        const int nclip = (rrb + TL_TOK - 1) / p.frames - clip0 + 1;
#pragma unroll
        for (int c = 0; c < FFN_MAXCLIP; ++c) {
            const int cc = c < nclip ? c : nclip - 1;
            prm[c] = *reinterpret_cast<const f32x4*>(p.film + (size_t)((clip0 + cc) % p.bmod) * p.film_ld + p.film_off + tid * 4);
        }
    }
    u32x4 hfr[32];
    {
        const char* xr = reinterpret_cast<const char*>(p.X) + (size_t)tb * 32 * 1024 + lane_off;
#pragma unroll
        for (int s = 0; s < 32; ++s) hfr[s] = *reinterpret_cast<const u32x4*>(xr + s * 1024);
    }
    float* sprm = reinterpret_cast<float*>(smem + 4 * FFN_CH);
    float* sb1 = sprm + FFN_MAXCLIP * 1024;
    float* sb2 = sb1 + 1024;
    float* sb3 = sb2 + 512;
    float* sconst = sb3 + 512;
#pragma unroll
    for (int c = 0; c < FFN_MAXCLIP; ++c) *reinterpret_cast<f32x4*>(sprm + 1024 * c + 4 * tid) = prm[c];
    for (int i = tid; i < 1024; i += 256) sb1[i] = p.b1[i];
    for (int i = tid; i < 512; i += 256) {
        sb2[i] = p.b2[i];
        sb3[i] = p.b3[i];
        sconst[i] = p.row_const ? p.row_const[i] : 0.f;
    }
#pragma unroll
    for (int s = 0; s < 32; ++s) asm volatile("" ::"v"(hfr[s]));
    trace_mark(p.trace, 1);

    const char* lds_lane = smem + lane * 16;
    // ---- phase C: y2 accumulators resident, hidden produced / consumed 32 features at a time ---------------------------------
    f32x16 acc2[16];
    __syncthreads();                                  // bias tables visible
#pragma unroll
    for (int ot = 0; ot < 16; ++ot)
#pragma unroll
        for (int qi = 0; qi < 4; ++qi) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(sb2 + ot * 32 + 16 * (qi >> 1) + 8 * h + 4 * (qi & 1));
#pragma unroll
            for (int e = 0; e < 4; ++e) acc2[ot][4 * qi + e] = b4[e];
        }
    u32x4 gfr[2];                                     // GELU(hidden tile j - 1) as two B fragments
#pragma unroll
    for (int c = 0; c < 2; ++c) { gfr[c][0] = 0; gfr[c][1] = 0; gfr[c][2] = 0; gfr[c][3] = 0; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // one iteration: [GEMM1 of hidden tile j] then [GEMM2 of tile j - 1 with the GELU of tile j issued in its MFMA shadow]
    auto iter = [&](int j, auto do1, auto do2) {
        constexpr bool DO1 = decltype(do1)::value, DO2 = decltype(do2)::value;
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        dma_chunk(2 * j + 1);
        dma_chunk(2 * j + 2);
        f32x16 acc1;
        if (DO1) {
#pragma unroll
            for (int qi = 0; qi < 4; ++qi) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(sb1 + j * 32 + 16 * (qi >> 1) + 8 * h + 4 * (qi & 1));
#pragma unroll
                for (int e = 0; e < 4; ++e) acc1[4 * qi + e] = b4[e];
            }
            mfma_run<32, FFN_GS>(acc1, lds_lane + ((2 * j) & 3) * FFN_CH, hfr);
        }
        u32x4 gnew[2];
        if (DO2) {
            const char* w2 = lds_lane + ((2 * j - 1) & 3) * FFN_CH;
            u32x4 aw[2][FFN_GS];                      // fragment (2 ot + ks) of the chunk feeds output tile ot, k step ks
#pragma unroll
            for (int i = 0; i < FFN_GS; ++i) aw[0][i] = *reinterpret_cast<const u32x4*>(w2 + i * 1024);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < 32 / FFN_GS; ++g) {
                if (g + 1 < 32 / FFN_GS) {
#pragma unroll
                    for (int i = 0; i < FFN_GS; ++i) aw[(g + 1) & 1][i] = *reinterpret_cast<const u32x4*>(w2 + ((g + 1) * FFN_GS + i) * 1024);
                }
#pragma unroll
                for (int i = 0; i < FFN_GS; ++i) {
                    constexpr int dummy = 0; (void)dummy;
                    const int fi = g * FFN_GS + i;    // compile-time after unrolling
                    acc2[fi >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aw[g & 1][i]),
                                                                             __builtin_bit_cast(bf16x8, gfr[fi & 1]), acc2[fi >> 1], 0, 0, 0);
                }
            }
        }
        if (DO1) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = gelu_fast(acc1[8 * c + e]);
                gnew[c][0] = pack_bf16(v[0], v[1]); gnew[c][1] = pack_bf16(v[2], v[3]);
                gnew[c][2] = pack_bf16(v[4], v[5]); gnew[c][3] = pack_bf16(v[6], v[7]);
            }
        }
        if (DO2) {
            // issue order: per group of 4 MFMAs the 4 LDS reads of the next group, and the GELU's VALU ops spread under the MFMAs
#pragma unroll
            for (int g = 0; g < 32 / FFN_GS; ++g) {
                if (g + 1 < 32 / FFN_GS) __builtin_amdgcn_sched_group_barrier(0x100, FFN_GS, 0);
#pragma unroll
                for (int i = 0; i < FFN_GS; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (DO1) __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);
                }
            }
        }
        if (DO1) { gfr[0] = gnew[0]; gfr[1] = gnew[1]; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    iter(0, std::true_type{}, std::false_type{});
    for (int j = 1; j < 32; ++j) iter(j, std::true_type{}, std::true_type{});
    iter(32, std::false_type{}, std::true_type{});
    // chunks 64 .. 66 (W3 tiles 0 .. 2) were requested in the last two iterations and have landed

    // ---- LayerNorm statistics from the fp32 accumulators; folded FiLM + SiLU; packed bf16 B fragments ------------------
    u32x4 yfr[32];
    {
        float sum = 0.f;
#pragma unroll
        for (int ot = 0; ot < 16; ++ot)
#pragma unroll
            for (int e = 0; e < 16; ++e) sum += acc2[ot][e];
        sum += __shfl_xor(sum, 32, 64);
        const float mean = sum * (1.0f / 512.f);
        float sq = 0.f;
#pragma unroll
        for (int ot = 0; ot < 16; ++ot)
#pragma unroll
            for (int e = 0; e < 16; ++e) { const float d = acc2[ot][e] - mean; sq = fmaf(d, d, sq); }
        sq += __shfl_xor(sq, 32, 64);
        const float rstd = 1.0f / sqrtf(sq * (1.0f / 512.f) + 1e-5f);
        const float nmr = -mean * rstd;
        const int rr = row >= p.half_row0 ? row - p.half_row0 : row;
        int ci = rr / p.frames - clip0;
        ci = ci < FFN_MAXCLIP ? ci : FFN_MAXCLIP - 1;
        const float* ca = sprm + ci * 1024 + 8 * h;
        const float* cb = ca + 512;
#pragma unroll
        for (int ot = 0; ot < 16; ++ot)
#pragma unroll
            for (int c = 0; c < 2; ++c) {             // fragment s = 2 ot + c holds features 32 ot + 16 c + 8 h + (0..7)
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(ca + 32 * ot + 16 * c), a1 = *reinterpret_cast<const f32x4*>(ca + 32 * ot + 16 * c + 4);
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(cb + 32 * ot + 16 * c), b1 = *reinterpret_cast<const f32x4*>(cb + 32 * ot + 16 * c + 4);
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float t = fmaf(acc2[ot][8 * c + e], rstd, nmr);
                    const float y = fmaf(t, e < 4 ? a0[e & 3] : a1[e & 3], e < 4 ? b0[e & 3] : b1[e & 3]);
                    v[e] = y * __builtin_amdgcn_rcpf(1.0f + __expf(-y));
                }
                u32x4 o;
                o.x = pack_bf16(v[0], v[1]); o.y = pack_bf16(v[2], v[3]); o.z = pack_bf16(v[4], v[5]); o.w = pack_bf16(v[6], v[7]);
                yfr[2 * ot + c] = o;
            }
    }

    // ---- phase D: h <- h + Linear3(yfr); residual tiles prefetched 8 ahead, stores deferred by one tile --------------------
    constexpr int NT = 16, RING = 8;
    f32x4 rres[RING][4];
    const size_t fbase = ((size_t)tb * NT * 4 * 64 + lane) * 4;          // + nt * 1024 floats + qi * 256
#pragma unroll
    for (int u = 0; u < RING; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) rres[u][q] = *reinterpret_cast<const f32x4*>(p.R + fbase + (size_t)u * 1024 + q * 256);
    char* Ctb = reinterpret_cast<char*>(p.Ct);
    const float const_on = (p.row_const != nullptr && row < p.n_const_rows) ? 1.0f : 0.0f;
    f32x16 prev;
#pragma unroll
    for (int e = 0; e < 16; ++e) prev[e] = 0.f;
    auto store_tile = [&](int nt, const f32x16& v) {
        const size_t fidx = fbase + (size_t)nt * 1024;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                const int qi = 2 * c + qq;
                f32x4 o; o.x = v[4 * qi]; o.y = v[4 * qi + 1]; o.z = v[4 * qi + 2]; o.w = v[4 * qi + 3];
                *reinterpret_cast<f32x4*>(p.Cf + fidx + qi * 256) = o;
            }
            u32x4 o;
            o.x = pack_bf16(v[8 * c + 0], v[8 * c + 1]); o.y = pack_bf16(v[8 * c + 2], v[8 * c + 3]);
            o.z = pack_bf16(v[8 * c + 4], v[8 * c + 5]); o.w = pack_bf16(v[8 * c + 6], v[8 * c + 7]);
            *reinterpret_cast<u32x4*>(Ctb + ((size_t)tb * (2 * NT) + 2 * nt + c) * 1024 + lane_off) = o;
        }
    };
    for (int nt0 = 0; nt0 < NT; nt0 += RING) {
#pragma unroll
        for (int u = 0; u < RING; ++u) {
            const int nt = nt0 + u;
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (nt > 0) store_tile(nt - 1, prev);
            dma_chunk(64 + nt + 3);                   // W3 tile nt + 3 into the slot tile nt - 1 was read from (nt .. nt + 2 have landed)
            f32x16 a3;
#pragma unroll
            for (int qi = 0; qi < 4; ++qi) {
                const int col = nt * 32 + 16 * (qi >> 1) + 8 * h + 4 * (qi & 1);
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(sb3 + col);
                const f32x4 c4 = *reinterpret_cast<const f32x4*>(sconst + col);
#pragma unroll
                for (int e = 0; e < 4; ++e) a3[4 * qi + e] = fmaf(const_on, c4[e], b4[e]);
            }
            mfma_run<32>(a3, lds_lane + ((64 + nt) & 3) * FFN_CH, yfr);
#pragma unroll
            for (int qi = 0; qi < 4; ++qi)
#pragma unroll
                for (int e = 0; e < 4; ++e) prev[4 * qi + e] = a3[4 * qi + e] + rres[u][qi][e];
            // refill this ring slot with the residual tile RING ahead (clamped: unconditional loads)
            const int ntn = nt + RING < NT ? nt + RING : NT - 1;
#pragma unroll
            for (int q = 0; q < 4; ++q) rres[u][q] = *reinterpret_cast<const f32x4*>(p.R + fbase + (size_t)ntn * 1024 + q * 256);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    store_tile(NT - 1, prev);
    trace_mark(p.trace, 2);
}

// ---- launchers -------------------------------------------------------------------------------------------------------
int launch_tl2_linear(const TlArgs& a, int pro, hipStream_t s) {
    DSH_REQUIRE(a.M > 0 && a.N > 0 && a.N % 32 == 0, "tl2_linear: N must be a positive multiple of 32");
    DSH_REQUIRE(a.K == 512 || a.K == 1024, "tl2_linear: K must be 512 or 1024");
    DSH_REQUIRE(a.ldx == (pro == 3 ? 512 : a.K), "tl2_linear: the tiled input must be exactly K features wide");
    DSH_REQUIRE(((uintptr_t)a.X % 16) == 0 && ((uintptr_t)a.W % 16) == 0, "tl2_linear: operands must be 16-byte aligned");
    DSH_REQUIRE(!a.Cf || !a.cf_rowmajor || a.ldcf % 4 == 0, "tl2_linear: row-major output leading dim");
    DSH_REQUIRE(!(a.cf_rowmajor && (a.R || a.Ct)), "tl2_linear: the row-major fp32 output has no residual / bf16 shadow");
    DSH_REQUIRE(pro >= 0 && pro <= 3, "tl2_linear: unknown prologue");
    DSH_REQUIRE(pro == 0 || pro == 2 || (a.gamma && a.beta), "tl2_linear: LayerNorm prologue needs gamma/beta");
    DSH_REQUIRE(pro != 2 || (a.film && a.frames > 0 && a.bmod > 0 && a.film_ld % 4 == 0 && a.film_off % 4 == 0),
                "tl2_linear: FiLM prologue needs the folded film table");
    DSH_REQUIRE(pro != 2 || std::min((TL_TOK - 1) / a.frames + 2, a.bmod) <= TL_MAXCLIP,
                "tl2_linear: FiLM prologue: too many clips per 128-token block (clips shorter than 26 frames need batch <= 6)");
    DSH_REQUIRE(pro != 2 || a.K == 512, "tl2_linear: FiLM prologue is instantiated for K = 512");
    DSH_REQUIRE(!a.row_const || (pro == 2 && a.R && a.act == ACT_NONE), "tl2_linear: row_const is only wired into the StylizationBlock instantiation");
    DSH_REQUIRE(pro != 3 || (a.K == 1024 && a.X1 && a.X2 && a.kreal > 896 - 1 && a.kreal <= 1024), "tl2_linear: concat prologue arguments");
    // N is split over grid.y only when the token blocks alone cannot fill the chip (window-chain batches)
    const int mblocks = ceil_div(a.M, TL_TOK), ntiles = a.N / 32;
    int tpb = ntiles;
    if (mblocks < 256) { const int want = ceil_div(512, mblocks); tpb = ceil_div(ntiles, want < ntiles ? want : ntiles); }
    TlArgs b = a;
    b.tiles_per_block = tpb;
    const dim3 grid(mblocks, ceil_div(ntiles, tpb)), block(256);
    const int lds = 2 * a.K * 64 + 2 * a.N * 4;
    DSH_REQUIRE(lds <= 160 * 1024, "tl2_linear: N too large for the LDS bias table");
    typedef void (*kern_t)(TlArgs);
    struct Variant { int k, pro, has_r, out, act; kern_t fn; };
#define TLV(P, R, O, A) {512, P, R, O, A, tl2_linear_kernel<512, P, (R) != 0, O, A>}
#define TLV1K(P, R, O, A) {1024, P, R, O, A, tl2_linear_kernel<1024, P, (R) != 0, O, A>}
    static const Variant variants[] = {
        TLV(1, 0, 2, ACT_NONE),   // sa_block: LayerNorm -> q|k|v                       (bf16 out)
        TLV(2, 1, 3, ACT_NONE),   // StylizationBlock: LN+FiLM+SiLU -> Linear -> +h     (fp32 h + bf16 shadow)
        TLV(0, 0, 2, ACT_GELU),   // ffn.linear1 + GELU                                  (bf16 out)
        TLV(0, 0, 2, ACT_NONE), TLV(0, 1, 3, ACT_NONE), TLV(0, 0, 2, ACT_SILU), TLV(1, 0, 1, ACT_NONE),
        TLV(2, 0, 2, ACT_NONE), TLV(0, 1, 1, ACT_NONE), TLV(0, 0, 1, ACT_NONE),
        TLV(0, 0, 4, ACT_NONE),   // encoder `out` head: plain rows -> fp32 row-major
        TLV1K(0, 0, 2, ACT_NONE),  // ffn.linear2                                        (bf16 out)
        TLV1K(0, 0, 2, ACT_SILU),
        TLV1K(0, 1, 3, ACT_NONE),  // feat_proj.3 + residual                             (fp32 h + bf16 shadow)
        TLV1K(0, 0, 1, ACT_NONE), TLV1K(0, 1, 1, ACT_NONE),
        TLV1K(3, 0, 2, ACT_SILU),  // feat_proj: concat + LayerNorm prologue -> Linear -> SiLU
    };
#undef TLV
#undef TLV1K
    constexpr int NV = sizeof(variants) / sizeof(variants[0]);
    static bool attr = false;
    if (!attr) {
        for (int i = 0; i < NV; ++i)
            DSH_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(variants[i].fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    const int out = (a.Cf ? (a.cf_rowmajor ? 4 : 1) : 0) | (a.Ct ? 2 : 0), has_r = a.R ? 1 : 0;
    kern_t fn = nullptr;
    for (int i = 0; i < NV; ++i)
        if (variants[i].k == a.K && variants[i].pro == pro && variants[i].has_r == has_r && variants[i].out == out && variants[i].act == a.act) fn = variants[i].fn;
    DSH_REQUIRE(fn != nullptr, "tl2_linear: this (prologue, residual, outputs, activation) combination is not instantiated");
    hipLaunchKernelGGL(fn, grid, block, lds, s, b);
    DSH_HIP_CHECK(hipGetLastError());
    return 0;
}

bool tl2_ffn_supported(int M, int frames, int bmod) {
    // whole-chip token counts only (no N split), and at most FFN_MAXCLIP clips per 128-token block
    return M >= 128 * 64 && std::min((TL_TOK - 1) / frames + 2, bmod) <= FFN_MAXCLIP;
}

int launch_tl2_ffn(const Tl2FfnArgs& a, hipStream_t s) {
    DSH_REQUIRE(a.M > 0 && a.X && a.Wffn && a.b1 && a.b2 && a.b3 && a.film && a.R && a.Cf && a.Ct, "tl2_ffn: null operand");
    DSH_REQUIRE(a.frames > 0 && a.bmod > 0 && a.film_ld % 4 == 0 && a.film_off % 4 == 0, "tl2_ffn: folded FiLM table");
    DSH_REQUIRE(std::min((TL_TOK - 1) / a.frames + 2, a.bmod) <= FFN_MAXCLIP, "tl2_ffn: too many clips per 128-token block");
    static bool attr = false;
    if (!attr) {
        DSH_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(tl2_ffn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, FFN_LDS));
        attr = true;
    }
    hipLaunchKernelGGL(tl2_ffn_kernel, dim3(ceil_div(a.M, TL_TOK)), dim3(256), FFN_LDS, s, a);
    DSH_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---- host-side packing ---------------------------------------------------------------------------------------------------
// element index (in bf16 units) of W'[32 nt + n][k] (rows already pi-permuted: n is the STORED row inside the tile) in the
// fragment-ordered copy of a [N, K] weight
size_t tl2_frag_index(int K, int nt, int n, int k) {
    const int s = k >> 4, hh = (k >> 3) & 1, j = k & 7;
    return (((size_t)nt * (K / 16) + s) * 64 + (n + 32 * hh)) * 8 + j;
}

}  // namespace dsh
