// Token-per-lane Linears, fourth form (round 6): BOTH operands stream through LDS.
//
// tl2.hip keeps a wave's 32 token rows stationary in registers (128 fragment registers at K = 512, 256 at K = 1024) and streams only
// the weights through LDS.  At K = 1024 that costs the kernel its second wave per SIMD, a 256 KB row burst per block in front of the
// first MFMA (23 % of a feat_proj.1 block, every CU in it at once) and one accumulator per wave (64 dependent MFMAs per tile):
// feat_proj.1 ran at 0.83 PFLOP/s where the vendor GEMM shows 1.10 - 1.15 on the same shape (profiles/r04_library_gemm_yardstick.txt).
// This file is the other kernel class for those launches: a classic output-stationary tile with weights AND activations staged
// through an LDS ring by LDS-DMA — possible without a single layout change, because both operands already live in HBM as the
// 1 KB MFMA fragment images the matrix instruction wants:
//   * weights in fragment order (tl2_frag_index): tile nt, k step s -> 1 KB at ((nt * K/16 + s) * 1024), lane-linear;
//   * activations tiled (tl_linear.hip): token block tb, k step s -> 1 KB at ((tb * W/16 + s) * 1024), token n at 32 n + 16 h
//     (the DMA's per-lane SOURCE offset undoes that, so that the LDS image is lane-linear too and every ds_read_b128 is conflict free).
// Geometry: a wave owns 64 tokens x 128 features = 2 x 4 accumulators of 32 x 32 (8 independent MFMA chains); a block is NWT x 2
// waves = (64 NWT) tokens x 256 features, two waves per SIMD.  One stage = 32 of K: 2 fragments per 32-row tile of either operand;
// the ring holds RING stages.  Per stage a wave issues 16 MFMAs, 12 ds_read_b128 (the fragments of the next half stage, one half
// stage = 8 MFMAs ahead, into the other register set), its share of the DMA of stage q + RING and — behind a folded LayerNorm —
// the 16 v_dot2c of the row moments, all in explicit issue slots.  The one barrier of a stage sits BETWEEN its two halves: it
// publishes stage q + 1 and frees stage q's slot, and the MFMAs behind it already have their operands.
// Arithmetic is operation for operation that of tl2_linear_kernel (same k order, accumulators started from the bias or from zero,
// same moment sums, same epilogue expressions): results are bit-identical, so the window-chain kernels (tl_small.hip) and the
// sharded bit-identity invariants hold unchanged.  Zero-padded k steps (K = 896 / 999 inside 1024) are simply not executed.
// Reference: models/transformer.py:284-289 (feat_proj), :304-338 (concat + residual), :119-125 (q | k | v behind one LayerNorm).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "dsh_common.h"
#include "dsh_kernels.h"
#include "tl_common.h"

namespace dsh {

struct Tl4Args {
    const void* X0; const void* X1; const void* X2; const void* X3;   // tiled activation segments of the (virtual concat) row
    int fr0, fr1, fr2, fr3;                                           // fragments (16 features) per token block and segment
    int q1a, q1b, q1c;                                                // first stage of segment 1 / 2 / 3
    const void* W; int KF;                                            // fragment-ordered weight, fragments per 32-row tile (Kpad / 16)
    int nstages;                                                      // stages of K = 32 actually executed
    const float* d; const float* c; float kn;                         // FOLD: d, c vectors and LayerNorm width; else d = bias (or null)
    void* Ct; void* Clo; const void* Rhi; const void* Rlo;            // tiled bf16 out (hi plane) / lo plane / residual planes
    int NT, ntt, nfb, rev;                                            // N / 32, token tiles, 256-feature blocks, descending token order
};

typedef __attribute__((address_space(3))) const u32x4* lfrag4_t;

// 2 KB (the k steps 2 q, 2 q + 1 of one 32-row tile: two consecutive fragments) global -> LDS by two LDS-DMA pieces that share M0 and the
// scalar offset.  A __device__ function, not code inside a lambda of the kernel: the host pass cannot resolve the gfx950 builtins inside a
// lambda body and then drops the kernel's host stub WITHOUT a diagnostic (the library failed to load with an undefined kernel symbol).
__device__ __forceinline__ void tl4_dma_pair(uint64_t base, char* lds, int voff, int soff) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(base), 0, 0x7ffffff0, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr_t)lds, 16, voff, soff, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr_t)lds, 16, voff, soff, 1024, 0);
}
__device__ __forceinline__ void tl4_dma_kb(const void* base, char* lds, int voff, int soff) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7ffffff0, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr_t)lds, 16, voff, soff, 0, 0);
}

template <int NWT, int RING, bool FOLD, int ACT, bool HL, int ABL = 0>      // ABL (bench only): 1 no DMA in the loop, 2 no MFMAs, 3 no moments, 4 no reads
__global__ __attribute__((amdgpu_flat_work_group_size(NWT * 128, NWT * 128), amdgpu_waves_per_eu(2, 2)))
void tl4_linear_kernel(Tl4Args p) {
    constexpr int NW = NWT * 2;                 // waves: NWT along tokens x 2 along features
    constexpr int TT = NWT * 2;                 // 32-token tiles per block
    constexpr int NPAIR = TT + 8;               // fragment pairs (k steps 2 q, 2 q + 1 of one 32-row tile) per stage: tokens, then features
    constexpr int PPW = NPAIR / NW;             // pairs a wave moves per stage
    constexpr int STAGE = NPAIR * 2048;
    static_assert(NPAIR % NW == 0, "whole pairs per wave");
    static_assert(!(FOLD && HL), "the residual form has no folded LayerNorm");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wt = wave >> 1, wf = wave & 1;
    const int h = lane >> 5;
    // block -> (token tile, feature block): the nfb feature blocks of a token tile sit on ONE XCD (block b runs on XCD b % 8) next to
    // each other in dispatch order, so the activation rows are fetched once per XCD L2; the weights (1 - 3 MB) stay resident in all eight
    const int bid = blockIdx.x, xcd = bid & 7, jj = bid >> 3;
    const int fb = jj % p.nfb;
    int ttile = (jj / p.nfb) * 8 + xcd;
    if (ttile >= p.ntt) return;
    if (p.rev) ttile = p.ntt - 1 - ttile;
    const int tb0 = ttile * TT, nt0 = fb * 8;
    const int lane_off = (lane & 31) * 32 + h * 16, lane16 = lane * 16;

    // d (bias or folded LayerNorm constant) and c of this block's 256 features -> LDS behind the ring, by one DMA piece each (256 floats
    // = 64 lanes x 16 B); they are older than every stage piece, so the first counted wait covers them
    float* sd = reinterpret_cast<float*>(smem + RING * STAGE);
    float* sc = sd + 256;
    if (wave == 0) tl4_dma_kb(p.d, reinterpret_cast<char*>(sd), lane16, nt0 * 128);
    if (FOLD && wave == 1) tl4_dma_kb(p.c, reinterpret_cast<char*>(sc), lane16, nt0 * 128);
    // the segment table lives in SGPRs for the whole kernel (selected from the kernarg segment per stage it cost an s_load + lgkmcnt(0) behind
    // every barrier)
    auto sgpr64 = [](const void* q) -> uint64_t {
        const uint64_t v = reinterpret_cast<uint64_t>(q);
        return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    };
    const uint64_t xs0 = sgpr64(p.X0), xs1 = sgpr64(p.X1), xs2 = sgpr64(p.X2), xs3 = sgpr64(p.X3), wbase = sgpr64(p.W);
    const int fr0 = __builtin_amdgcn_readfirstlane(p.fr0), fr1 = __builtin_amdgcn_readfirstlane(p.fr1), fr2 = __builtin_amdgcn_readfirstlane(p.fr2),
              fr3 = __builtin_amdgcn_readfirstlane(p.fr3), q1a = __builtin_amdgcn_readfirstlane(p.q1a), q1b = __builtin_amdgcn_readfirstlane(p.q1b),
              q1c = __builtin_amdgcn_readfirstlane(p.q1c);
    // this wave's PPW fragment pairs of a stage: pair pp < TT = token tile pp, else feature tile pp - TT (wave uniform, loop invariant)
    int vo[PPW], so_mul[PPW], so_add[PPW]; bool is_b[PPW];
#pragma unroll
    for (int ii = 0; ii < PPW; ++ii) {
        const int pp = wave * PPW + ii;
        is_b[ii] = pp < TT;
        vo[ii] = is_b[ii] ? lane_off : lane16;
        so_mul[ii] = is_b[ii] ? tb0 + pp : 0;                       // B: ((tb0 + pp) * sfr + 2 (q - qs0)) KB
        so_add[ii] = is_b[ii] ? 0 : (nt0 + pp - TT) * p.KF;         // A: ((nt0 + ft) * KF + 2 q) KB
    }
    uint64_t cx = xs0; int cfr = fr0, cq0 = 0;                      // segment of the stage being issued
    // issue of one stage in two parts, so that the main loop can spread it over its MFMA slots: issue_begin selects the concat segment
    // (stages are issued in ascending order: the current segment is carried along — a 4-way select per stage became a lookup table in
    // scratch / in the kernarg segment, with an s_load + lgkmcnt(0) behind every barrier), issue_pair moves pair ii of this wave
    char* idst = nullptr; int iq = 0;
    auto issue_begin = [&](int q, int slot) {
        idst = smem + slot * STAGE + wave * (PPW * 2048); iq = q;
        const bool e1 = q == q1a, e2 = q == q1b, e3 = q == q1c;
        cx = e1 ? xs1 : cx; cfr = e1 ? fr1 : cfr;
        cx = e2 ? xs2 : cx; cfr = e2 ? fr2 : cfr;
        cx = e3 ? xs3 : cx; cfr = e3 ? fr3 : cfr;
        cq0 = (e1 || e2 || e3) ? q : cq0;
    };
    auto issue_pair = [&](int ii) {
        const uint64_t base = is_b[ii] ? cx : wbase;
        const int so = (so_mul[ii] * cfr + so_add[ii] + 2 * (is_b[ii] ? iq - cq0 : iq)) * 1024;
        tl4_dma_pair(base, idst + ii * 2048, vo[ii], so);
    };
    auto issue_stage = [&](int q, int slot) {
        issue_begin(q, slot);
#pragma unroll
        for (int ii = 0; ii < PPW; ++ii) issue_pair(ii);
    };

    const int ns = p.nstages;                                       // (the launcher guarantees ns >= RING)
#pragma unroll
    for (int q = 0; q < RING; ++q) issue_stage(q, q);
    // stage 0 (and d / c) have landed when at most the (RING - 1) younger stages' pieces are outstanding
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RING - 1) * PPW * 2) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // ---- accumulators: started from the bias (plain Linear) or from zero (folded LayerNorm: the bias is part of d) ----
    f32x16 acc[2][4];
#pragma unroll
    for (int ft = 0; ft < 4; ++ft)
#pragma unroll
        for (int qi = 0; qi < 4; ++qi) {
            f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
            if (!FOLD) b4 = *reinterpret_cast<const f32x4*>(sd + (wf * 4 + ft) * 32 + 16 * (qi >> 1) + 8 * h + 4 * (qi & 1));
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc[0][ft][4 * qi + e] = b4[e]; acc[1][ft][4 * qi + e] = b4[e]; }
        }

    typedef __attribute__((address_space(3))) const char* lcptr_t;
    const lcptr_t lds_b = (lcptr_t)smem + lane16 + (wt * 2) * 2048;             // this wave's two token tiles
    const lcptr_t lds_a = (lcptr_t)smem + lane16 + (TT + wf * 4) * 2048;        // ... and four feature tiles
    u32x4 fbA[2], faA[4], fbB[2], faB[4];                                        // fragment sets of the two half stages
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const bf16x2_t ones = __builtin_bit_cast(bf16x2_t, 0x3f803f80u);
    float sm[2][2] = {{0.f, 0.f}, {0.f, 0.f}}, sq[2][2] = {{0.f, 0.f}, {0.f, 0.f}};   // row moments [token tile][j & 1] (row_moments_bf16's order)

    fbA[0] = *(lfrag4_t)(lds_b);
    fbA[1] = *(lfrag4_t)(lds_b + 2048);
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) faA[ft] = *(lfrag4_t)(lds_a + ft * 2048);

    // one half stage: 8 MFMAs on (cb, ca); the 6 fragment reads of the NEXT half stage go into (nb, na); moments of the current B fragments
    auto half_stage = [&](u32x4 (&cb)[2], u32x4 (&ca)[4], u32x4 (&nb)[2], u32x4 (&na)[4], lcptr_t nsrc_b, lcptr_t nsrc_a, auto wf_tag, auto&& extra) {
        constexpr int WF = decltype(wf_tag)::value;
        static_for<8>([&](auto m_tag) {
            constexpr int m = decltype(m_tag)::value;
            constexpr int tt = m >> 2, ft = m & 3;
            if constexpr (ABL != 2) acc[tt][ft] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ca[ft]), __builtin_bit_cast(bf16x8, cb[tt]), acc[tt][ft], 0, 0, 0);
            if constexpr (ABL != 4) {
                if constexpr (m == 0) nb[0] = *(lfrag4_t)(nsrc_b);
                else if constexpr (m <= 4) na[m - 1] = *(lfrag4_t)(nsrc_a + (m - 1) * 2048);
                else if constexpr (m == 5) nb[1] = *(lfrag4_t)(nsrc_b + 2048);
            }
            if constexpr (FOLD && ABL != 3) {
                // one dword of the current B fragments per slot (8 slots = 2 token tiles x 4 dwords), in row_moments_bf16's order
                // the two feature waves of a token quarter read the same B fragments: wave WF takes the dwords j with (j & 1) == WF, i.e. exactly
                // the partial sums sm[.][WF] / sq[.][WF] of row_moments_bf16; the halves are exchanged through LDS behind the loop
                constexpr int t2 = m >> 2, j0 = (m & 3);
                if constexpr ((j0 & 1) == WF) {
                    const uint32_t w = cb[t2][j0];
                    const bf16x2_t v = __builtin_bit_cast(bf16x2_t, w);
                    sm[t2][WF] = __builtin_amdgcn_fdot2_f32_bf16(v, ones, sm[t2][WF], false);
                    sq[t2][WF] = __builtin_amdgcn_fdot2_f32_bf16(v, v, sq[t2][WF], false);
                }
            }
            extra(m_tag);
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    auto no_extra = [](auto) {};

    int slot = 0;                                                   // ring slot of stage q
    // ISSUE: stage q + RING exists and goes into stage q's slot behind the barrier.  STRICT: fewer than RING - 2 younger stages are in
    // flight behind stage q + 1 (the last RING - 1 stages): the counted wait would pass too early, drain instead
    auto stage_body = [&](int q, auto issue_tag, auto strict_tag, auto wf_tag) {
        constexpr bool ISSUE = decltype(issue_tag)::value, STRICT = decltype(strict_tag)::value;
        const int nslot = slot + 1 == RING ? 0 : slot + 1;
        // first half: k step 2 q on set A; reads of k step 2 q + 1 (same slot, + 1 KB)
        half_stage(fbA, faA, fbB, faB, lds_b + slot * STAGE + 1024, lds_a + slot * STAGE + 1024, wf_tag, no_extra);
        // stage q + 1 published, slot of stage q free
        if (STRICT || ABL == 1) asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)" ::"n"((RING - 2) * PPW * 2) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // second half: k step 2 q + 1 on set B; reads of k step 2 (q + 1) from the next slot (stale bytes behind the last stage: unused);
        // the DMA of stage q + RING into the slot the barrier just freed rides in its slots: segment select behind MFMA 0, one pair behind
        // each of MFMAs 1, 3, 5 — issued as one burst between the barrier and the MFMAs it kept both waves of a SIMD off the matrix pipe
        half_stage(fbB, faB, fbA, faA, lds_b + nslot * STAGE, lds_a + nslot * STAGE, wf_tag, [&](auto m_tag) {
            constexpr int m = decltype(m_tag)::value;
            if constexpr (ISSUE && ABL != 1) {
                if constexpr (m == 0) issue_begin(q + RING, slot);
                if constexpr ((m & 1) == 1 && (m >> 1) < PPW) issue_pair(m >> 1);
            }
        });
        slot = nslot;
    };
    auto main_loop = [&](auto wf_tag) {
        int q = 0;
        for (; q + RING < ns; ++q) stage_body(q, std::true_type{}, std::false_type{}, wf_tag);
        stage_body(q, std::false_type{}, std::false_type{}, wf_tag);
        for (++q; q < ns; ++q) stage_body(q, std::false_type{}, std::true_type{}, wf_tag);
    };
    if constexpr (!FOLD) main_loop(std::integral_constant<int, 0>{});           // (no moments: one copy of the loop)
    else if (wf == 0) main_loop(std::integral_constant<int, 0>{});
    else main_loop(std::integral_constant<int, 1>{});

    // ---- epilogue ----------------------------------------------------------------------------------------------------------------
    float rstd[2] = {1.f, 1.f}, nmr[2] = {0.f, 0.f};
    if (FOLD) {
        // the partner wave (same tokens, other feature half) holds the other partial sums: exchange through the (now idle) ring
        f32x4* xch = reinterpret_cast<f32x4*>(smem);
        const f32x4 mine = {wf ? sm[0][1] : sm[0][0], wf ? sq[0][1] : sq[0][0], wf ? sm[1][1] : sm[1][0], wf ? sq[1][1] : sq[1][0]};
        xch[wave * 64 + lane] = mine;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const f32x4 theirs = xch[(wave ^ 1) * 64 + lane];
        sm[0][0] = wf ? theirs[0] : mine[0]; sm[0][1] = wf ? mine[0] : theirs[0];
        sq[0][0] = wf ? theirs[1] : mine[1]; sq[0][1] = wf ? mine[1] : theirs[1];
        sm[1][0] = wf ? theirs[2] : mine[2]; sm[1][1] = wf ? mine[2] : theirs[2];
        sq[1][0] = wf ? theirs[3] : mine[3]; sq[1][1] = wf ? mine[3] : theirs[3];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            float sum = sm[tt][0] + sm[tt][1], sumsq = sq[tt][0] + sq[tt][1];
            sum += __shfl_xor(sum, 32, 64);
            sumsq += __shfl_xor(sumsq, 32, 64);
            const float mean = sum / p.kn;
            sumsq = fmaxf(sumsq - sum * mean, 0.f);
            rstd[tt] = 1.0f / sqrtf(sumsq / p.kn + 1e-5f);
            nmr[tt] = -mean * rstd[tt];
        }
    }
    char* Ctb = reinterpret_cast<char*>(p.Ct);
    char* Clb = reinterpret_cast<char*>(p.Clo);
    const char* Rhb = reinterpret_cast<const char*>(p.Rhi);
    const char* Rlb = reinterpret_cast<const char*>(p.Rlo);
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        const int tb = tb0 + wt * 2 + tt;
        u32x4 rhi[4][2], rlo[4][2];
        if (HL) {
#pragma unroll
            for (int ft = 0; ft < 4; ++ft)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const size_t idx = ((size_t)tb * (2 * p.NT) + 2 * (nt0 + wf * 4 + ft) + c) * 1024 + lane_off;
                    rhi[ft][c] = *reinterpret_cast<const u32x4*>(Rhb + idx);
                    rlo[ft][c] = *reinterpret_cast<const u32x4*>(Rlb + idx);
                }
        }
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) {
            const int nt = nt0 + wf * 4 + ft;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                float v[8];
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2) {
                    const int qi = 2 * c + q2;
                    f32x4 d4 = {0.f, 0.f, 0.f, 0.f}, c4 = {0.f, 0.f, 0.f, 0.f};
                    if (FOLD) {
                        const int col = (wf * 4 + ft) * 32 + 16 * (qi >> 1) + 8 * h + 4 * (qi & 1);
                        d4 = *reinterpret_cast<const f32x4*>(sd + col);
                        c4 = *reinterpret_cast<const f32x4*>(sc + col);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x = acc[tt][ft][4 * qi + e];
                        if (FOLD) x = fmaf(x, rstd[tt], fmaf(nmr[tt], c4[e], d4[e]));
                        if (ACT == ACT_GELU) x = gelu_fast(x);
                        else if (ACT == ACT_SILU) x = x * __builtin_amdgcn_rcpf(1.0f + __expf(-x));
                        v[4 * q2 + e] = x;
                    }
                }
                const size_t idx = ((size_t)tb * (2 * p.NT) + 2 * nt + c) * 1024 + lane_off;
                if (HL) {
                    hl_accumulate(v, rhi[ft][c], rlo[ft][c]);
                    u32x4 oh, ol;
                    hl_split(v, oh, ol);
                    *reinterpret_cast<u32x4*>(Ctb + idx) = oh;
                    *reinterpret_cast<u32x4*>(Clb + idx) = ol;
                } else {
                    u32x4 o;
                    o.x = pack_bf16(v[0], v[1]); o.y = pack_bf16(v[2], v[3]);
                    o.z = pack_bf16(v[4], v[5]); o.w = pack_bf16(v[6], v[7]);
                    *reinterpret_cast<u32x4*>(Ctb + idx) = o;
                }
            }
        }
    }
}

namespace {

// DSH_TL4_V=a: 8 waves, 256 tokens x 256 features, ring of 4 stages (128 KB, one block per CU); b (default): 4 waves, 128 x 256, ring of 3
// (72 KB, two blocks per CU whose barriers and epilogues are independent)
int tl4_variant() {               // (read per launch: the op-level tests flip it inside one process)
    const char* e = getenv("DSH_TL4_V");
    return (e && (*e == 'a' || *e == 'A')) ? 0 : 1;
}

int launch_variant(int v, const Tl4Args& b, int mode, int M, hipStream_t s) {
    typedef void (*kern_t)(Tl4Args);
    // [variant][mode]: mode 0 = folded concat-LayerNorm + SiLU (feat_proj.1), 1 = hi / lo residual planes (feat_proj.3), 2 = folded LayerNorm (q|k|v)
    static const kern_t fns[2][3] = {
        {tl4_linear_kernel<4, 4, true, ACT_SILU, false>, tl4_linear_kernel<4, 4, false, ACT_NONE, true>, tl4_linear_kernel<4, 4, true, ACT_NONE, false>},
        {tl4_linear_kernel<2, 3, true, ACT_SILU, false>, tl4_linear_kernel<2, 3, false, ACT_NONE, true>, tl4_linear_kernel<2, 3, true, ACT_NONE, false>}};
    static const kern_t abl[2][4] = {
        {tl4_linear_kernel<4, 4, true, ACT_SILU, false, 1>, tl4_linear_kernel<4, 4, true, ACT_SILU, false, 2>, tl4_linear_kernel<4, 4, true, ACT_SILU, false, 3>, tl4_linear_kernel<4, 4, true, ACT_SILU, false, 4>},
        {tl4_linear_kernel<2, 3, true, ACT_SILU, false, 1>, tl4_linear_kernel<2, 3, true, ACT_SILU, false, 2>, tl4_linear_kernel<2, 3, true, ACT_SILU, false, 3>, tl4_linear_kernel<2, 3, true, ACT_SILU, false, 4>}};
    const char* ae = getenv("DSH_TL4_ABL");                 // bench only: ablations of the feat_proj.1 instantiation (results are garbage)
    const int ab = ae ? atoi(ae) : 0;
    const int nwt = v == 0 ? 4 : 2, ring = v == 0 ? 4 : 3;
    const int lds = ring * (nwt * 2 + 8) * 2048 + 2048;
    static const bool attr = [] {
        bool ok = true;
        for (int vv = 0; vv < 2; ++vv)
            for (int m = 0; m < 3; ++m)
                ok &= hipFuncSetAttribute(reinterpret_cast<const void*>(fns[vv][m]), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
        for (int vv = 0; vv < 2; ++vv)
            for (int m = 0; m < 4; ++m)
                ok &= hipFuncSetAttribute(reinterpret_cast<const void*>(abl[vv][m]), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
        return ok;
    }();
    DSH_REQUIRE(attr, "tl4_linear: hipFuncSetAttribute failed");
    Tl4Args a = b;
    a.ntt = ceil_div(M, nwt * 64);
    const dim3 grid(8 * a.nfb * ceil_div(a.ntt, 8)), block(nwt * 128);
    hipLaunchKernelGGL((mode == 0 && ab >= 1 && ab <= 4) ? abl[v][ab - 1] : fns[v][mode], grid, block, lds, s, a);
    DSH_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace

// which of the token-per-lane launches this kernel class is instantiated for (mode index of launch_variant) or -1
static int tl4_mode(const TlArgs& a, int pro) {
    if (a.N % 256 != 0 || a.Cf || a.cf_rowmajor) return -1;
    if (pro == 3 && a.K == 1024 && !a.R && !a.Rlo && a.Ct && a.act == ACT_SILU && a.X1 && a.X2 && (a.X3 || a.kreal <= 896) && a.kreal > 0 && a.kreal <= 1024) return 0;
    if (pro == 0 && a.K == 1024 && a.R && a.Rlo && a.Ct && a.Clo && a.act == ACT_NONE) return 1;
    if (pro == 1 && a.K == 512 && !a.R && !a.Rlo && a.Ct && a.act == ACT_NONE) return 2;
    return -1;
}

bool tl4_linear_supported(const TlArgs& a, int pro) { return tl4_mode(a, pro) >= 0; }

// same arguments as launch_tl2_linear (a.W = fragment-ordered weight; folded LayerNorm: a.bias = d, a.row_const = c)
int launch_tl4_linear(const TlArgs& a, int pro, hipStream_t s) {
    const int mode = tl4_mode(a, pro);
    DSH_REQUIRE(mode >= 0, "tl4_linear: this launch is not instantiated for the LDS-tiled kernel");
    DSH_REQUIRE(a.M > 0 && ((uintptr_t)a.X % 16) == 0 && ((uintptr_t)a.W % 16) == 0, "tl4_linear: operands must be 16-byte aligned");
    DSH_REQUIRE(a.bias && (mode == 1 || a.row_const), "tl4_linear: needs the bias (folded LayerNorm: d as bias and c as row_const)");
    Tl4Args b;
    const int big = 0x3fffffff;
    b.X0 = a.X; b.X1 = a.X1; b.X2 = a.X2; b.X3 = a.X3 ? a.X3 : a.X2;
    if (mode == 0) {
        b.fr0 = 512 / 16; b.fr1 = a.ld1 / 16; b.fr2 = a.ld2 / 16; b.fr3 = a.ld3 / 16;
        DSH_REQUIRE(a.ldx == 512 && a.ld1 == 256 && a.ld2 == 128 && (!a.X3 || a.ld3 == 128), "tl4_linear: concat segments are 512 | 256 | 128 | 128 wide");
        b.q1a = 16; b.q1b = 24; b.q1c = 28;
        b.nstages = ceil_div(a.kreal, 32);
        b.kn = (float)a.kreal;
    } else {
        b.X1 = b.X2 = b.X3 = a.X;
        b.fr0 = b.fr1 = b.fr2 = b.fr3 = a.K / 16;
        b.q1a = b.q1b = b.q1c = big;
        b.nstages = a.K / 32;
        b.kn = (float)a.K;
    }
    b.W = a.W; b.KF = a.K / 16;
    b.d = a.bias; b.c = a.row_const;
    b.Ct = a.Ct; b.Clo = a.Clo; b.Rhi = a.R; b.Rlo = a.Rlo;
    b.NT = a.N / 32; b.nfb = a.N / 256; b.rev = a.rev; b.ntt = 0;
    DSH_REQUIRE(b.nstages >= 4, "tl4_linear: K too small for the stage ring");
    const int v = tl4_variant();
    g_tl_last_variant = v == 0 ? 4 : 5;
    return launch_variant(v, b, mode, a.M, s);
}

}  // namespace dsh
