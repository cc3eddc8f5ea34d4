// fp32 parity path (BASELINE configs[1]): a Linear together with what the reference computes in FRONT of it, in one launch.
//
//   PRO 1  y = Linear(LayerNorm(concat(x0 .. x3)))      sa_block.norm + q|k|v         models/transformer.py:106-108, :119-125
//                                                       feat_proj.0 + feat_proj.1     models/transformer.py:284-289, :304-312
//          The LayerNorm affine is folded into the weight on the host (W' = gamma (.) W, c[n] = sum_k W'[n][k], d[n] = b[n] + sum_k beta[k] W[n][k]):
//            LN(x) W^T + b = rstd (x W'^T - mean c) + d.
//          The block stages the RAW rows of its 64 tokens (every K tile passes through the staging registers of the same eight lanes per row), so
//          the row moments are accumulated on the way, for free, and applied in the epilogue.  The concat is never materialised: K tile kt is read
//          from the segment it falls in (segment widths are multiples of the 32-float K tile; the last one is zero padded).
//   PRO 2  y = Linear(SiLU(LN(x) (1 + scale) + shift)) + residual    StylizationBlock        models/transformer.py:86-97
//          SiLU is not linear, so the rows are normalised in the staging registers (between the global load and the LDS write): one pass over the
//          block's 64 x K input for the moments (same lanes, same rows as the main loop: no exchange), then xhat scale' + shift' -> SiLU per K tile
//          with the LayerNorm affine folded into the per-clip FiLM table (scale' = gamma (1 + scale), shift' = beta (1 + scale) + shift:
//          film_expand_kernel, fold = 1).
//
// These replace ln_rows / concat_ln_rows / ln_film_silu_rows + gemm_nt_kernel<float>: four row kernels and their 2 x M x K x 4 bytes round trips per
// layer.  Moments are one-pass SHIFTED sums (sum (x - x0), sum (x - x0)^2 with x0 = the row's first element): as robust against a large row mean
// as the two-pass form of the row kernels, to fp32 round-off.
//   PRO 0  y = act(x W^T + b) (+ residual): the same main loop without a front (feat_proj.3, ffn.linear1 / 2 of the fp32 path).
// Tile = gemm_nt_kernel<float, 1, 1, 1, 2>'s (gemm.hip): 64 x 64, four waves, exact-fp32 v_mfma_f32_32x32x2_f32, double-buffered LDS stages; the main
// loop is software-pipelined (below).
#include <stdio.h>
#include <stdlib.h>

#include "dsh_common.h"
#include "dsh_kernels.h"

namespace dsh {

typedef float g32x16 __attribute__((ext_vector_type(16)));
typedef float g32x4 __attribute__((ext_vector_type(4)));

constexpr int GP_ROW = 144;                  // LDS row: 128 B of K tile + 16 B pad (conflict-free ds_read_b128, as gemm.hip)
constexpr int GP_STAGE = 64 * GP_ROW;        // one operand, one stage
constexpr int GP_LDS = 4 * GP_STAGE;         // A + W, double buffered: 36,864 B -> four blocks per CU

__device__ __forceinline__ float gp_silu(float x) {
    // v_exp_f32 / v_rcp_f32 (1 ulp each): the transform is recomputed by every N tile of a row block, it has to stay a handful of instructions
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896340736f));
}

typedef __attribute__((address_space(3))) void* gp_lptr_t;
__device__ __forceinline__ void gp_dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds_wave, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (gp_lptr_t)lds_wave, 16, voff, soff, 0, 0);
}

template <int PRO, bool DMA>
__global__ __launch_bounds__(256) void gemm_f32_pro_kernel(GemmProArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int bm, bn;
    {   // XCD-aware order (block b runs on XCD b % 8): the N tiles of one M tile share an L2
        const int NT = p.nt_n, MT = p.nt_m, bid = blockIdx.x;
        if (MT >= 8) { const int group = bid / (8 * NT), rem = bid % (8 * NT); bm = group * 8 + (rem % 8); bn = rem / 8; }
        else { bm = bid / NT; bn = bid % NT; }
        if (bm >= MT) return;
    }
    const int m0 = bm * 64, n0 = bn * 64;
    const int nk = p.K / 32;
    const int c16 = tid & 7, srow = tid >> 3;               // staging: this thread owns 16-byte column c16 of rows srow and srow + 32

    const char* w_src[2];
    int lds_off[2];
    int arow[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = srow + 32 * i;
        int ra = m0 + row; ra = ra < p.M ? ra : p.M - 1;
        int rw = n0 + row; rw = rw < p.N ? rw : p.N - 1;
        arow[i] = ra;
        w_src[i] = reinterpret_cast<const char*>(p.W) + (size_t)rw * p.ldw * 4 + c16 * 16;
        lds_off[i] = row * GP_ROW + c16 * 16;
    }
    // A sources per segment (PRO 0 / 2: one segment), pre-biased by the segment's first K tile so that tile kt is at base + 128 kt in every segment
    // The concat is walked incrementally: `ap` is where the NEXT tile of this thread's two rows is, advanced by one K tile per fetch and
    // re-based at a segment start (two-way selects one after the other: a four-way select over the segment bases becomes a dynamically indexed
    // scratch table + flat loads).
    const int e0 = p.seg_end[0], e1 = p.seg_end[1], e2 = p.seg_end[2];
    const char *ap[2], *a_s1[2], *a_s2[2], *a_s3[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        ap[i] = reinterpret_cast<const char*>(p.seg[0]) + (size_t)arow[i] * p.seg_ld[0] * 4 + c16 * 16;
        if (PRO == 1) {
            a_s1[i] = reinterpret_cast<const char*>(p.seg[1]) + (size_t)arow[i] * p.seg_ld[1] * 4 + c16 * 16;
            a_s2[i] = reinterpret_cast<const char*>(p.seg[2]) + (size_t)arow[i] * p.seg_ld[2] * 4 + c16 * 16;
            a_s3[i] = reinterpret_cast<const char*>(p.seg[3]) + (size_t)arow[i] * p.seg_ld[3] * 4 + c16 * 16;
        }
    }
    int ak = 0;                                                 // tile the next fetch reads
    auto load_a = [&](int i) -> g32x4 { return *reinterpret_cast<const g32x4*>(ap[i]); };
    auto advance_a = [&]() {                                    // (behind the last tile the pointers stay: the loop re-fetches it once)
        if (ak + 1 < nk) {
            ++ak;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ap[i] += 128;
                if (PRO == 1) {
                    ap[i] = ak == e0 ? a_s1[i] : ap[i];
                    ap[i] = ak == e1 ? a_s2[i] : ap[i];
                    ap[i] = ak == e2 ? a_s3[i] : ap[i];
                }
            }
        }
    };

    float x0[2] = {0.f, 0.f}, s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
    if (PRO != 0)
#pragma unroll
    for (int i = 0; i < 2; ++i) x0[i] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.seg[0]) + (size_t)arow[i] * p.seg_ld[0] * 4);
    auto moments = [&](const g32x4& v, int i) {
        const float a = v.x - x0[i], b = v.y - x0[i], c = v.z - x0[i], d = v.w - x0[i];
        s1[i] += (a + b) + (c + d);
        s2[i] = fmaf(a, a, fmaf(b, b, fmaf(c, c, fmaf(d, d, s2[i]))));
    };
    const float invP = 1.0f / (float)p.k_real;
    float mean[2], rstd[2];
    auto finish_moments = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float a = s1[i], b = s2[i];
#pragma unroll
            for (int o = 1; o < 8; o <<= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
            // the K - k_real zero-padded columns were staged like the others: take their (0 - x0) terms out again
            const float npad = (float)(p.K - p.k_real);
            a = fmaf(npad, x0[i], a);
            b = fmaf(-npad * x0[i], x0[i], b);
            const float dm = a * invP;
            mean[i] = x0[i] + dm;
            const float var = fmaxf(fmaf(-dm, dm, b * invP), 0.f);
            rstd[i] = 1.0f / sqrtf(var + 1e-5f);
        }
    };

    const char* f_src[2] = {nullptr, nullptr};
    if (PRO == 2) {
        if (p.stats) {
            // the producer of x left per-row group moments (mean_g, M2_g = sum (x - mean_g)^2 over a group of stat_gs columns, stat_groups <= 16 groups
            // per row): combined here in a fixed order (Chan et al.), no pass over the rows
            const int G = p.stat_groups;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float2* sp = reinterpret_cast<const float2*>(p.stats) + (size_t)arow[i] * G;
                const float2 g0 = c16 < G ? sp[c16] : make_float2(0.f, 0.f);
                const float2 g1 = c16 + 8 < G ? sp[c16 + 8] : make_float2(0.f, 0.f);
                float sm = g0.x + g1.x, m2 = g0.y + g1.y;
#pragma unroll
                for (int o = 1; o < 8; o <<= 1) { sm += __shfl_xor(sm, o, 64); m2 += __shfl_xor(m2, o, 64); }
                const float mu = sm / (float)G;
                float dv = (c16 < G ? (g0.x - mu) * (g0.x - mu) : 0.f) + (c16 + 8 < G ? (g1.x - mu) * (g1.x - mu) : 0.f);
#pragma unroll
                for (int o = 1; o < 8; o <<= 1) dv += __shfl_xor(dv, o, 64);
                mean[i] = mu;
                rstd[i] = 1.0f / sqrtf((m2 + (float)p.stat_gs * dv) * invP + 1e-5f);
            }
        } else {
            // pass 1: moments of this block's 64 rows (2 x nk independent 16-byte loads per lane, from L2 for every N tile but the first)
#pragma unroll 4
            for (int kt = 0; kt < nk; ++kt) {
                const g32x4 v0 = *reinterpret_cast<const g32x4*>(ap[0] + (size_t)kt * 128), v1 = *reinterpret_cast<const g32x4*>(ap[1] + (size_t)kt * 128);
                moments(v0, 0); moments(v1, 1);
            }
            finish_moments();
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int b = (arow[i] / p.frames) % p.bmod;
            f_src[i] = reinterpret_cast<const char*>(p.film + (size_t)b * p.film_ld + p.film_off) + c16 * 16;
        }
    }
    const size_t shift_off = (size_t)p.k_real * 4;            // [scale'(D) | shift'(D)]
    auto sty = [&](const g32x4& v, const g32x4& sc, const g32x4& sh, int i) -> g32x4 {
        g32x4 r;
        r.x = gp_silu(fmaf((v.x - mean[i]) * rstd[i], sc.x, sh.x));
        r.y = gp_silu(fmaf((v.y - mean[i]) * rstd[i], sc.y, sh.y));
        r.z = gp_silu(fmaf((v.z - mean[i]) * rstd[i], sc.z, sh.z));
        r.w = gp_silu(fmaf((v.w - mean[i]) * rstd[i], sc.w, sh.w));
        return r;
    };

    g32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    if constexpr (DMA) {
    // ---- LDS-DMA form (buffer_load_dwordx4 ... lds): W — and, without a front (PRO 0), the rows too — go global -> LDS without passing through
    // registers: no ds_write instructions, no staging registers, and a whole iteration of cover (the DMA of tile kt + 2 is issued right behind the
    // barrier of iteration kt, into the stage that barrier has just freed, and awaited in front of the barrier of iteration kt + 1).
    // A DMA instruction writes LDS linearly (lane L at base + 16 L), so a stage is 64 unpadded 128-byte rows per operand; the fragment reads (32
    // rows, one 16-byte chunk) stay conflict-free through an XOR swizzle of the chunk position, pos = chunk ^ ((row >> 1) & 7), applied on the
    // GLOBAL side of the DMA (each lane fetches the chunk that belongs at its LDS position).  Fronts (PRO 1 / 2) keep the register path for the
    // rows (their VALU work needs them) and write the same swizzled image.
    constexpr int ST = 8192;                                    // one operand, one stage
    char* sA = smem;
    char* sW = smem + 2 * ST;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.W), 0, (int)((size_t)p.N * p.ldw * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.seg[0]), 0, (int)((size_t)p.M * p.seg_ld[0] * 4), 0x00020000);
    int a_lds[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { const int lr = srow + 32 * i; a_lds[i] = lr * 128 + ((c16 ^ ((lr >> 1) & 7)) * 16); }
    // (PRO 1: one descriptor and one per-lane offset pair per concat segment; the segment of a tile is wave-uniform)
    const __amdgpu_buffer_rsrc_t arsrc1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(PRO == 1 ? p.seg[1] : p.seg[0]), 0, (int)((size_t)p.M * p.seg_ld[PRO == 1 ? 1 : 0] * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t arsrc2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(PRO == 1 ? p.seg[2] : p.seg[0]), 0, (int)((size_t)p.M * p.seg_ld[PRO == 1 ? 2 : 0] * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t arsrc3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(PRO == 1 ? p.seg[3] : p.seg[0]), 0, (int)((size_t)p.M * p.seg_ld[PRO == 1 ? 3 : 0] * 4), 0x00020000);
    int w_voff[2], a_voff[2], a_voff1[2], a_voff2[2], a_voff3[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int lr = 16 * wave + 8 * j + (lane >> 3), g = (lane & 7) ^ ((lr >> 1) & 7);
        int rw = n0 + lr; rw = rw < p.N ? rw : p.N - 1;
        int ra = m0 + lr; ra = ra < p.M ? ra : p.M - 1;
        w_voff[j] = rw * p.ldw * 4 + g * 16;
        a_voff[j] = ra * p.seg_ld[0] * 4 + g * 16;
        a_voff1[j] = ra * p.seg_ld[1] * 4 + g * 16; a_voff2[j] = ra * p.seg_ld[2] * 4 + g * 16; a_voff3[j] = ra * p.seg_ld[3] * 4 + g * 16;
    }
    const int wave_lds = __builtin_amdgcn_readfirstlane(16 * wave * 128);
    auto dma = [&](int kt, int st) {
#pragma unroll
        for (int j = 0; j < 2; ++j) gp_dma16(wrsrc, sW + st * ST + wave_lds + j * 1024, w_voff[j], kt * 128);
        if (PRO == 0 || (PRO == 1 && kt < e0)) {
#pragma unroll
            for (int j = 0; j < 2; ++j) gp_dma16(arsrc, sA + st * ST + wave_lds + j * 1024, a_voff[j], kt * 128);
        } else if (PRO == 1) {
            if (kt < e1) {
#pragma unroll
                for (int j = 0; j < 2; ++j) gp_dma16(arsrc1, sA + st * ST + wave_lds + j * 1024, a_voff1[j], (kt - e0) * 128);
            } else if (kt < e2) {
#pragma unroll
                for (int j = 0; j < 2; ++j) gp_dma16(arsrc2, sA + st * ST + wave_lds + j * 1024, a_voff2[j], (kt - e1) * 128);
            } else {
#pragma unroll
                for (int j = 0; j < 2; ++j) gp_dma16(arsrc3, sA + st * ST + wave_lds + j * 1024, a_voff3[j], (kt - e2) * 128);
            }
        }
    };
    // PRO 1: the row moments are taken from the LDS image of a tile (this thread's chunk c16 of rows srow, srow + 32, read back right behind the
    // barrier that publishes the tile; the VALU work rides in the next MFMA group)
    g32x4 mv[2];
    auto read_mv = [&](int st) {
#pragma unroll
        for (int i = 0; i < 2; ++i) mv[i] = *reinterpret_cast<const g32x4*>(sA + st * ST + a_lds[i]);
    };
    // register path of the rows (PRO 1 / 2): this thread's chunk c16 of rows srow, srow + 32 lands at the swizzled position
    g32x4 ra[2], rsc[2], rsh[2], ta[2];
    auto fetch = [&](int) {                                      // (tiles are fetched in order: the argument documents which one)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            ra[i] = load_a(i);
            if (PRO == 2) { rsc[i] = *reinterpret_cast<const g32x4*>(f_src[i] + (size_t)ak * 128); rsh[i] = *reinterpret_cast<const g32x4*>(f_src[i] + shift_off + (size_t)ak * 128); }
        }
        advance_a();
    };
    auto front = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (PRO == 1) moments(ra[i], i);
            ta[i] = PRO == 2 ? sty(ra[i], rsc[i], rsh[i], i) : ra[i];
        }
    };
    auto stage = [&](int st) {
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<g32x4*>(sA + st * ST + a_lds[i]) = ta[i];
    };
    // fragment read offsets: row (32 w? + lane & 31), chunk 2 c + (lane >> 5) at its swizzled position
    int a_fo[4], w_fo[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int ar = wm * 32 + (lane & 31), wr = wn * 32 + (lane & 31), g = 2 * c + (lane >> 5);
        a_fo[c] = ar * 128 + ((g ^ ((ar >> 1) & 7)) * 16);
        w_fo[c] = wr * 128 + ((g ^ ((wr >> 1) & 7)) * 16);
    }
    dma(0, 0);
    if (PRO == 2) { fetch(0); front(); stage(0); }
    dma(nk > 1 ? 1 : 0, 1);
    if (PRO == 2) fetch(nk > 1 ? 1 : 0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    g32x4 fa[4], fb[4];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        fa[c] = *reinterpret_cast<const g32x4*>(sA + a_fo[c]);
        fb[c] = *reinterpret_cast<const g32x4*>(sW + w_fo[c]);
    }
    if (PRO == 1) read_mv(0);                                   // (tile 0; its VALU work rides in the first MFMA group)
    int cur = 0;
    for (int kt = 0; kt + 1 < nk; ++kt) {
#pragma unroll
        for (int c = 2; c < 4; ++c) {
            fa[c] = *reinterpret_cast<const g32x4*>(sA + cur * ST + a_fo[c]);
            fb[c] = *reinterpret_cast<const g32x4*>(sW + cur * ST + w_fo[c]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[c].x, fa[c].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[c].y, fa[c].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[c].z, fa[c].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[c].w, fa[c].w, acc, 0, 0, 0);
        }
        if (PRO != 0) {
            if (PRO == 2) front();                              // (registers: raw rows of tile kt + 1)
            else { moments(mv[0], 0); moments(mv[1], 1); }      // (registers: this thread's chunks of tile kt, read back from LDS one barrier ago)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, PRO == 2 ? 10 : 3, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (PRO == 2) { stage(cur ^ 1); fetch(kt + 2 < nk ? kt + 2 : nk - 1); }
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[2].x, fa[2].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[2].y, fa[2].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[2].z, fa[2].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[2].w, fa[2].w, acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        // tile kt + 1 complete in LDS.  PRO 2: its DMA is older than exactly the six register loads fetch() has just issued (2 row chunks + 4 FiLM
        // chunks, unconditional, fenced on both sides by the memory-clobber asm / sched_barrier above and below): "at most 6 in flight" = DMA landed.
        // (Fewer younger loads than the count would let the wait pass early — keep fetch() and this count together;
        //  test_lds_dma_and_register_staging_give_identical_results compares the two forms bit for bit.)
        if (PRO != 2) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        dma(kt + 2 < nk ? kt + 2 : nk - 1, cur);                // (behind the last tile: re-fetched once, into a stage nobody reads)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            fa[c] = *reinterpret_cast<const g32x4*>(sA + (cur ^ 1) * ST + a_fo[c]);
            fb[c] = *reinterpret_cast<const g32x4*>(sW + (cur ^ 1) * ST + w_fo[c]);
        }
        if (PRO == 1) read_mv(cur ^ 1);                          // (tile kt + 1 is complete behind this barrier)
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[3].x, fa[3].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[3].y, fa[3].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[3].z, fa[3].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[3].w, fa[3].w, acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        cur ^= 1;
    }
    {   // last tile
        if (PRO == 1) { moments(mv[0], 0); moments(mv[1], 1); }
#pragma unroll
        for (int c = 2; c < 4; ++c) {
            fa[c] = *reinterpret_cast<const g32x4*>(sA + cur * ST + a_fo[c]);
            fb[c] = *reinterpret_cast<const g32x4*>(sW + cur * ST + w_fo[c]);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[c].x, fa[c].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[c].y, fa[c].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[c].z, fa[c].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[c].w, fa[c].w, acc, 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // (the re-fetched last tile must have landed before the stage is reused below)
    } else {
    // Software pipeline (round 6).  The first form of this loop was gemm_nt_kernel's: load tile kt + 1 -> 16 MFMAs -> LDS write -> barrier ->
    // fragment reads, every LDS access issued right in front of its use.  The four waves that share a SIMD run their dependent MFMA chains
    // interleaved, i.e. in lockstep, so all 16 waves of a CU reach the LDS phases (and the staging VALU work) together and the matrix pipe idles
    // behind them (57 - 63 % of the fp32 matrix peak at M = 8704).  Here everything else is issued beside MFMAs of the same wave:
    //   top          fragments of chunks 2, 3 of tile kt are requested
    //   G1 (8 MFMAs) chunks 0, 1 multiply; between them the staging VALU work on the registers that hold the RAW tile kt + 1 (fetched one
    //                iteration ago): row moments (PRO 1) or normalise -> FiLM -> SiLU (PRO 2)
    //   then         tile kt + 1 is written to the other LDS stage, tile kt + 2 is requested into the same registers
    //   G2a (4)      chunk 2 multiplies (cover for the LDS writes)
    //   barrier      tile kt + 1 is visible, stage `cur` is free
    //   G2b (4)      chunk 3 multiplies while the fragments of chunks 0, 1 of tile kt + 1 arrive
    g32x4 ra[2], rw[2], rsc[2], rsh[2], ta[2];
    auto fetch = [&](int) {                                      // (tiles are fetched in order: the argument documents which one)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            ra[i] = load_a(i);
            rw[i] = *reinterpret_cast<const g32x4*>(w_src[i] + (size_t)ak * 128);
            if (PRO == 2) { rsc[i] = *reinterpret_cast<const g32x4*>(f_src[i] + (size_t)ak * 128); rsh[i] = *reinterpret_cast<const g32x4*>(f_src[i] + shift_off + (size_t)ak * 128); }
        }
        advance_a();
    };
    char* sA = smem;
    char* sW = smem + 2 * GP_STAGE;
    auto front = [&](bool real) {                               // VALU side of the staging: ra -> ta
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (PRO == 1 && real) moments(ra[i], i);
            ta[i] = PRO == 2 ? sty(ra[i], rsc[i], rsh[i], i) : ra[i];
        }
    };
    auto stage = [&](int st) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<g32x4*>(sA + st * GP_STAGE + lds_off[i]) = ta[i];
            *reinterpret_cast<g32x4*>(sW + st * GP_STAGE + lds_off[i]) = rw[i];
        }
    };
    fetch(0);
    front(true);
    stage(0);
    fetch(nk > 1 ? 1 : 0);
    __syncthreads();

    const int frag_row = lane & 31, frag_kb = (lane >> 5) * 16;
    const int a_frag0 = (wm * 32 + frag_row) * GP_ROW + frag_kb;
    const int w_frag0 = (wn * 32 + frag_row) * GP_ROW + frag_kb;
    g32x4 fa[4], fb[4];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        fa[c] = *reinterpret_cast<const g32x4*>(sA + a_frag0 + c * 32);
        fb[c] = *reinterpret_cast<const g32x4*>(sW + w_frag0 + c * 32);
    }
    int cur = 0;
    const int abl = PRO == 0 ? p.abl : 0;                      // (ablation switches: front-less launches only)
    for (int kt = 0; kt + 1 < nk; ++kt) {
        const char* cA = sA + cur * GP_STAGE;
        const char* cW = sW + cur * GP_STAGE;
        if (!(abl & 4))
#pragma unroll
        for (int c = 2; c < 4; ++c) {
            fa[c] = *reinterpret_cast<const g32x4*>(cA + a_frag0 + c * 32);
            fb[c] = *reinterpret_cast<const g32x4*>(cW + w_frag0 + c * 32);
        }
        __builtin_amdgcn_sched_barrier(0);                      // (hipcc otherwise sinks MFMAs below the barrier: the LDS accesses lose their cover)
        // G1 + the staging VALU work (hipcc's scheduler interleaves the two inside this region; D[n][m]: each lane ends up with 4 consecutive n of
        // one row m -> 16-byte epilogue I/O)
        if (!(abl & 8))
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[c].x, fa[c].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[c].y, fa[c].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[c].z, fa[c].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[c].w, fa[c].w, acc, 0, 0, 0);
        }
        front(true);
        if (PRO != 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);          // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, PRO == 2 ? 10 : 3, 0);   // a share of the VALU work
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // (every memory access of the loop is unconditional — the last pass re-fetches the last tile once more — so that hipcc's counted waits
        //  know exactly what is in flight at the loop head)
        if (!(abl & 2)) stage(cur ^ 1);
        if (!(abl & 1)) fetch(kt + 2 < nk ? kt + 2 : nk - 1);
        __builtin_amdgcn_sched_barrier(0);
        if (!(abl & 8)) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[2].x, fa[2].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[2].y, fa[2].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[2].z, fa[2].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[2].w, fa[2].w, acc, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!(abl & 16)) __syncthreads();
        if (!(abl & 4))
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            fa[c] = *reinterpret_cast<const g32x4*>(sA + (cur ^ 1) * GP_STAGE + a_frag0 + c * 32);
            fb[c] = *reinterpret_cast<const g32x4*>(sW + (cur ^ 1) * GP_STAGE + w_frag0 + c * 32);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!(abl & 8)) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[3].x, fa[3].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[3].y, fa[3].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[3].z, fa[3].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[3].w, fa[3].w, acc, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        cur ^= 1;
    }
    {   // last tile: nothing left to stage
#pragma unroll
        for (int c = 2; c < 4; ++c) {
            fa[c] = *reinterpret_cast<const g32x4*>(sA + cur * GP_STAGE + a_frag0 + c * 32);
            fb[c] = *reinterpret_cast<const g32x4*>(sW + cur * GP_STAGE + w_frag0 + c * 32);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[c].x, fa[c].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[c].y, fa[c].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[c].z, fa[c].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[c].w, fa[c].w, acc, 0, 0, 0);
        }
    }
    }
    __syncthreads();                                            // (the stage-0 region is reused for the row statistics below)

    float* stat = reinterpret_cast<float*>(smem);               // [64][2] (mean, rstd)
    if (PRO == 1) {
        finish_moments();
        if (c16 == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i) { stat[(srow + 32 * i) * 2] = mean[i]; stat[(srow + 32 * i) * 2 + 1] = rstd[i]; }
        }
        __syncthreads();
    }

    // ---- epilogue: [fold] -> bias -> activation -> (+residual) -> store.  D[n][m] layout: m = lane & 31, n = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    const int rl = wm * 32 + (lane & 31);
    const int row = m0 + rl;
    if (row >= p.M) return;
    float mu = 0.f, rs = 1.f;
    if (PRO == 1) { mu = stat[rl * 2]; rs = stat[rl * 2 + 1]; }
    float outv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) outv[r] = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int col = n0 + wn * 32 + 8 * q + 4 * (lane >> 5);
        if (col >= p.N) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[4 * q + e];
        const g32x4 b4 = *reinterpret_cast<const g32x4*>(p.bias + col);
        if (PRO == 1) {
            const g32x4 c4 = *reinterpret_cast<const g32x4*>(p.fc + col);
            v[0] = fmaf(rs, fmaf(-mu, c4.x, v[0]), b4.x); v[1] = fmaf(rs, fmaf(-mu, c4.y, v[1]), b4.y);
            v[2] = fmaf(rs, fmaf(-mu, c4.z, v[2]), b4.z); v[3] = fmaf(rs, fmaf(-mu, c4.w, v[3]), b4.w);
        } else {
            v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], p.act);
        if (p.R) { const g32x4 r4 = *reinterpret_cast<const g32x4*>(p.R + (size_t)row * p.ldr + col); v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w; }
        if (p.row_const && row < p.n_const_rows) { const g32x4 k4 = *reinterpret_cast<const g32x4*>(p.row_const + col); v[0] += k4.x; v[1] += k4.y; v[2] += k4.z; v[3] += k4.w; }
        g32x4 o4; o4.x = v[0]; o4.y = v[1]; o4.z = v[2]; o4.w = v[3];
        *reinterpret_cast<g32x4*>(p.C + (size_t)row * p.ldc + col) = o4;
#pragma unroll
        for (int e = 0; e < 4; ++e) outv[4 * q + e] = v[e];
    }
    if (p.stats_out) {
        // group moments of the rows just written, for the StylizationBlock launch that consumes them (p.stats there): this wave's 32 columns of row
        // `row` are the 16 values of this lane and the 16 of lane ^ 32 (N % 32 == 0 is required: no partial groups)
        float sm = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) sm += outv[r];
        sm += __shfl_xor(sm, 32, 64);
        const float mg = sm * (1.0f / 32.0f);
        float m2 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) m2 = fmaf(outv[r] - mg, outv[r] - mg, m2);
        m2 += __shfl_xor(m2, 32, 64);
        if (lane < 32) reinterpret_cast<float2*>(p.stats_out)[(size_t)row * (p.N / 32) + (n0 + wn * 32) / 32] = make_float2(mg, m2);
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// StylizationBlock launch with SPECIALISED WAVES (round 6, third step; measured slower than the four-wave form, off by default: see the launcher).  In gemm_f32_pro_kernel<2, .> the normalise -> FiLM -> SiLU transform of
// the rows sits in the instruction streams of the four MFMA waves: a wave issues in order, so its VALU work holds back its own next MFMA, and
// the four waves that share a SIMD do so together (lockstep) — the matrix pipe is busy 40 % of the launch (PMC) against 63 % without a front.
// Here a block is six waves: waves 0 - 3 run the front-less LDS-DMA loop (W by DMA, fragments, MFMAs, epilogue) and never touch the rows in
// global memory; waves 4 - 5 are PRODUCERS: they take the row moments, and per K tile load the raw rows + FiLM rows, transform them and write
// the swizzled LDS image the MFMA waves read.  MFMA and VALU are separate pipes: a producer's VALU instructions issue beside the other waves'
// MFMAs.  Same arithmetic as PRO 2 operation for operation (a producer thread owns four 16-byte chunks of one row instead of one chunk of two
// rows; the moments are accumulated in a different order: fp32 round-off).  One barrier per K tile for all six waves.
__global__ __launch_bounds__(384, 6) void gemm_f32_sty_ws_kernel(GemmProArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bm, bn;
    {
        const int NT = p.nt_n, MT = p.nt_m, bid = blockIdx.x;
        if (MT >= 8) { const int group = bid / (8 * NT), rem = bid % (8 * NT); bm = group * 8 + (rem % 8); bn = rem / 8; }
        else { bm = bid / NT; bn = bid % NT; }
        if (bm >= MT) return;
    }
    const int m0 = bm * 64, n0 = bn * 64;
    const int nk = p.K / 32;
    constexpr int ST = 8192;
    char* sA = smem;
    char* sW = smem + 2 * ST;

    if (wave >= 4) {
        // ================= producer: thread pt owns chunks 4 hh .. 4 hh + 3 of row pt >> 1 =================
        const int pt = tid - 256, lr = pt >> 1, hh = pt & 1;
        int ra = m0 + lr; ra = ra < p.M ? ra : p.M - 1;
        const char* src = reinterpret_cast<const char*>(p.seg[0]) + (size_t)ra * p.seg_ld[0] * 4 + hh * 64;
        const int b = (ra / p.frames) % p.bmod;
        const char* fsrc = reinterpret_cast<const char*>(p.film + (size_t)b * p.film_ld + p.film_off) + hh * 64;
        const size_t shift_off = (size_t)p.K * 4;
        int dst[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[j] = lr * 128 + (((4 * hh + j) ^ ((lr >> 1) & 7)) * 16);
        // moments: shifted one-pass sums over the whole row (this thread's half of every tile), combined with the other half's
        const float x0 = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.seg[0]) + (size_t)ra * p.seg_ld[0] * 4);
        float s1 = 0.f, s2 = 0.f;
        const int nk_m = (p.abl & 32) ? 0 : nk;                 // (ablation: no moments pass)
#pragma unroll 2
        for (int kt = 0; kt < nk_m; ++kt) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const g32x4 v = *reinterpret_cast<const g32x4*>(src + (size_t)kt * 128 + j * 16);
                const float a = v.x - x0, bq = v.y - x0, c = v.z - x0, d = v.w - x0;
                s1 += (a + bq) + (c + d);
                s2 = fmaf(a, a, fmaf(bq, bq, fmaf(c, c, fmaf(d, d, s2))));
            }
        }
        s1 += __shfl_xor(s1, 1, 64);
        s2 += __shfl_xor(s2, 1, 64);
        const float invP = 1.0f / (float)p.K, dm = s1 * invP;
        const float mean = x0 + dm;
        const float rstd = 1.0f / sqrtf(fmaxf(fmaf(-dm, dm, s2 * invP), 0.f) + 1e-5f);
        g32x4 raw[4], sc[4], sh[4];
        auto fetch = [&](int kt) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                raw[j] = *reinterpret_cast<const g32x4*>(src + (size_t)kt * 128 + j * 16);
                sc[j] = *reinterpret_cast<const g32x4*>(fsrc + (size_t)kt * 128 + j * 16);
                sh[j] = *reinterpret_cast<const g32x4*>(fsrc + shift_off + (size_t)kt * 128 + j * 16);
            }
        };
        auto put = [&](int st) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                g32x4 r;
                r.x = gp_silu(fmaf((raw[j].x - mean) * rstd, sc[j].x, sh[j].x));
                r.y = gp_silu(fmaf((raw[j].y - mean) * rstd, sc[j].y, sh[j].y));
                r.z = gp_silu(fmaf((raw[j].z - mean) * rstd, sc[j].z, sh[j].z));
                r.w = gp_silu(fmaf((raw[j].w - mean) * rstd, sc[j].w, sh[j].w));
                *reinterpret_cast<g32x4*>(sA + st * ST + dst[j]) = r;
            }
        };
        fetch(0);
        put(0);
        fetch(nk > 1 ? 1 : 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the tile-1 loads stay in flight across the barrier)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        int cur = 0;
        for (int kt = 0; kt + 1 < nk; ++kt) {
            put(cur ^ 1);                                       // tile kt + 1 (stage cur ^ 1 was last read before the previous barrier)
            fetch(kt + 2 < nk ? kt + 2 : nk - 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            cur ^= 1;
        }
        return;
    }

    // ================= MFMA waves: the front-less LDS-DMA loop, rows from the producers' LDS image =================
    const int wm = wave >> 1, wn = wave & 1;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.W), 0, (int)((size_t)p.N * p.ldw * 4), 0x00020000);
    int w_voff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int lr = 16 * wave + 8 * j + (lane >> 3), g = (lane & 7) ^ ((lr >> 1) & 7);
        int rw = n0 + lr; rw = rw < p.N ? rw : p.N - 1;
        w_voff[j] = rw * p.ldw * 4 + g * 16;
    }
    const int wave_lds = __builtin_amdgcn_readfirstlane(16 * wave * 128);
    int a_fo[4], w_fo[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int ar = wm * 32 + (lane & 31), wr = wn * 32 + (lane & 31), g = 2 * c + (lane >> 5);
        a_fo[c] = ar * 128 + ((g ^ ((ar >> 1) & 7)) * 16);
        w_fo[c] = wr * 128 + ((g ^ ((wr >> 1) & 7)) * 16);
    }
    g32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) gp_dma16(wrsrc, sW + wave_lds + j * 1024, w_voff[j], 0);
#pragma unroll
    for (int j = 0; j < 2; ++j) gp_dma16(wrsrc, sW + ST + wave_lds + j * 1024, w_voff[j], (nk > 1 ? 1 : 0) * 128);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    g32x4 fa[4], fb[4];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        fa[c] = *reinterpret_cast<const g32x4*>(sA + a_fo[c]);
        fb[c] = *reinterpret_cast<const g32x4*>(sW + w_fo[c]);
    }
    int cur = 0;
    for (int kt = 0; kt + 1 < nk; ++kt) {
#pragma unroll
        for (int c = 2; c < 4; ++c) {
            fa[c] = *reinterpret_cast<const g32x4*>(sA + cur * ST + a_fo[c]);
            fb[c] = *reinterpret_cast<const g32x4*>(sW + cur * ST + w_fo[c]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[c].x, fa[c].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[c].y, fa[c].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[c].z, fa[c].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[c].w, fa[c].w, acc, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // W tile kt + 1 has landed; the producers have written the rows
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        {
            const int k2 = kt + 2 < nk ? kt + 2 : nk - 1;
#pragma unroll
            for (int j = 0; j < 2; ++j) gp_dma16(wrsrc, sW + cur * ST + wave_lds + j * 1024, w_voff[j], k2 * 128);
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            fa[c] = *reinterpret_cast<const g32x4*>(sA + (cur ^ 1) * ST + a_fo[c]);
            fb[c] = *reinterpret_cast<const g32x4*>(sW + (cur ^ 1) * ST + w_fo[c]);
        }
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[3].x, fa[3].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[3].y, fa[3].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[3].z, fa[3].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[3].w, fa[3].w, acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        cur ^= 1;
    }
    {
#pragma unroll
        for (int c = 2; c < 4; ++c) {
            fa[c] = *reinterpret_cast<const g32x4*>(sA + cur * ST + a_fo[c]);
            fb[c] = *reinterpret_cast<const g32x4*>(sW + cur * ST + w_fo[c]);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[c].x, fa[c].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[c].y, fa[c].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[c].z, fa[c].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[c].w, fa[c].w, acc, 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // epilogue (D[n][m]: m = lane & 31, n = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)): bias -> activation -> + residual -> [+ row constant] -> store
    const int row = m0 + wm * 32 + (lane & 31);
    if (row >= p.M) return;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int col = n0 + wn * 32 + 8 * q + 4 * (lane >> 5);
        if (col >= p.N) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[4 * q + e];
        const g32x4 b4 = *reinterpret_cast<const g32x4*>(p.bias + col);
        v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], p.act);
        if (p.R) { const g32x4 r4 = *reinterpret_cast<const g32x4*>(p.R + (size_t)row * p.ldr + col); v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w; }
        if (p.row_const && row < p.n_const_rows) { const g32x4 k4 = *reinterpret_cast<const g32x4*>(p.row_const + col); v[0] += k4.x; v[1] += k4.y; v[2] += k4.z; v[3] += k4.w; }
        g32x4 o4; o4.x = v[0]; o4.y = v[1]; o4.z = v[2]; o4.w = v[3];
        *reinterpret_cast<g32x4*>(p.C + (size_t)row * p.ldc + col) = o4;
    }
}

int launch_gemm_f32_pro(const GemmProArgs& a, hipStream_t s) {
    DSH_REQUIRE(a.pro >= 0 && a.pro <= 2, "gemm_f32_pro: unknown prologue");
    DSH_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0 && a.K % 32 == 0 && a.N % 4 == 0, "gemm_f32_pro: K must be whole 32-float tiles, N a multiple of 4");
    DSH_REQUIRE(a.k_real > 0 && a.k_real <= a.K && a.ldw >= a.K && a.ldw % 4 == 0, "gemm_f32_pro: bad K / ldw");
    DSH_REQUIRE(a.bias && a.C && a.ldc % 4 == 0 && (!a.R || a.ldr % 4 == 0), "gemm_f32_pro: bias and an aligned output are required");
    DSH_REQUIRE(((uintptr_t)a.W % 16) == 0 && ((uintptr_t)a.C % 16) == 0 && ((uintptr_t)a.bias % 16) == 0 && ((uintptr_t)a.R % 16) == 0, "gemm_f32_pro: 16-byte alignment");
    int prev = 0;
    for (int i = 0; i < 4; ++i) {
        if (a.pro != 1 && i > 0) break;
        const int end = a.pro != 1 ? a.K / 32 : a.seg_end[i];
        DSH_REQUIRE(end >= prev && end <= a.K / 32, "gemm_f32_pro: segment bounds must be non-decreasing K-tile indices");
        if (end > prev) DSH_REQUIRE(a.seg[i] && a.seg_ld[i] % 4 == 0 && a.seg_ld[i] >= (end - prev) * 32 && ((uintptr_t)a.seg[i] % 16) == 0, "gemm_f32_pro: segment rows must be 16-byte aligned and cover their K tiles");
        prev = end;
    }
    DSH_REQUIRE(a.seg[0] && (a.pro != 1 || a.seg_end[3] == a.K / 32), "gemm_f32_pro: the segments must cover K");
    DSH_REQUIRE(!a.stats_out || (a.N % 64 == 0 && a.N / 32 <= 16), "gemm_f32_pro: group moments need whole 32-column groups, at most 16 per row");
    DSH_REQUIRE(!a.stats || (a.pro == 2 && a.stat_groups >= 1 && a.stat_groups <= 16 && a.stat_groups * a.stat_gs == a.K), "gemm_f32_pro: bad group moments");
    DSH_REQUIRE(!a.row_const || ((uintptr_t)a.row_const % 16) == 0, "gemm_f32_pro: 16-byte alignment");
    if (a.pro == 1) DSH_REQUIRE(a.fc && ((uintptr_t)a.fc % 16) == 0, "gemm_f32_pro: folded LayerNorm needs the weight row sums");
    if (a.pro == 2) DSH_REQUIRE(a.film && a.k_real == a.K && a.frames > 0 && a.bmod > 0 && a.film_ld % 4 == 0 && a.film_off % 4 == 0 && ((uintptr_t)a.film % 16) == 0,
                                "gemm_f32_pro: StylizationBlock front needs the folded FiLM table");
    typedef void (*kern_t)(GemmProArgs);
    static const kern_t kerns[2][3] = {{gemm_f32_pro_kernel<0, false>, gemm_f32_pro_kernel<1, false>, gemm_f32_pro_kernel<2, false>},
                                       {gemm_f32_pro_kernel<0, true>, gemm_f32_pro_kernel<1, true>, gemm_f32_pro_kernel<2, true>}};
    static bool attr = false;
    if (!attr) {
        for (int d = 0; d < 2; ++d)
            for (int q = 0; q < 3; ++q)
                DSH_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kerns[d][q]), hipFuncAttributeMaxDynamicSharedMemorySize, GP_LDS));
        attr = true;
    }
    // operands through LDS-DMA (buffer addressing: 32-bit byte offsets) unless DSH_GP_DMA=0
    static const int dma_on = [] { const char* e = getenv("DSH_GP_DMA"); return e ? atoi(e) : 1; }();
    bool dma = dma_on && (size_t)a.N * a.ldw * 4 < ((size_t)1 << 31);
    for (int i = 0; i < (a.pro == 1 ? 4 : 1); ++i) dma = dma && (!a.seg[i] || (size_t)a.M * a.seg_ld[i] * 4 < ((size_t)1 << 31));
    GemmProArgs b = a;
    {   // bench-only: DSH_GP_ABL drops parts of the register-staged main loop (DSH_GP_DMA=0, front-less launches) — results are garbage
        static const int abl = [] {
            const char* e = getenv("DSH_GP_ABL");
            const int v = e ? atoi(e) : 0;
            if (v) fprintf(stderr, "[diffsheg_hip] WARNING: DSH_GP_ABL=%d is set: fp32 GEMM launches skip parts of their main loop, results are GARBAGE\n", v);
            return v;
        }();
        b.abl = abl;
    }
    b.nt_n = ceil_div(a.N, 64);
    b.nt_m = ceil_div(a.M, 64);
    const int groups = ceil_div(b.nt_m, 8);
    const dim3 grid(b.nt_m >= 8 ? groups * 8 * b.nt_n : b.nt_m * b.nt_n);
    // (built, checked against fp64 and the four-wave form, measured SLOWER — 71.4 vs 63.1 us at M = 8704 (profiles/r06b_y_*): the producers' moments
    //  pass costs 9.5 us with 128 instead of 256 threads, and their VALU work costs the MFMA waves of the same SIMD as much as it did inside
    //  their own instruction streams, 61.9 vs 50.0 us without a front.  Off unless DSH_GP_WS=1.)
    static const int ws_on = [] { const char* e = getenv("DSH_GP_WS"); return e ? atoi(e) : 0; }();
    if (a.pro == 2 && dma && ws_on && !a.stats && !a.stats_out) {
        static bool ws_attr = false;
        if (!ws_attr) { DSH_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f32_sty_ws_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, GP_LDS)); ws_attr = true; }
        hipLaunchKernelGGL(gemm_f32_sty_ws_kernel, grid, dim3(384), 4 * 8192, s, b);
    } else hipLaunchKernelGGL(kerns[dma ? 1 : 0][a.pro], grid, dim3(256), GP_LDS, s, b);
    DSH_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace dsh
