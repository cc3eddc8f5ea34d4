// FFN branch of a decoder layer, third generation (round 4): the same math and operand layouts as tl2_ffn_kernel
//   g = GELU(h16 W1^T + b1); y2 = g W2^T + b2; h <- h + Linear3(SiLU(LN(y2) (1 + scale) + shift)) (+ next layer's CFG-null constant)
// (models/transformer.py:169-181, :86-97), re-skeletoned after the round-3 block timeline (prologue 26 k | phase C 102 k |
// LayerNorm / FiLM / SiLU stage 30 k | phase D 32 k cycles of a 190 k block against 82 k of MFMA issue):
//
//   * Linear3 starts K-OUTER on its first four output tiles (pass A).  K step kc only needs the SiLU'd fragment pair of y2 tile
//     kc, so the LayerNorm / FiLM / SiLU conversion of tile kc + 1 rides in the MFMA shadow of K step kc: of the old 30 k-cycle
//     stage only the row statistics and the first tile's conversion stay exposed.  Pass A holds 64 accumulator registers next
//     to the 256 of y2 (which shrink by 16 per K step while the bf16 fragments grow by 8): four tiles are what the 512-register
//     budget allows at its start (eight tiles spilled 300 registers).  The other twelve tiles follow tile-outer (pass B, the old
//     phase D) with every fragment at hand.
//   * the fp32 residual is the INITIAL VALUE of the Linear3 accumulators as before, but it is requested where registers come
//     free instead of in one burst: pass A's four tiles right after the last GEMM2 phase (into the registers the h16 fragments
//     vacate), pass B's two phases ahead of their tile.  These are ordinary loads; hipcc counts them exactly as long
//     as no STORE is in its scoreboard ("loads and stores complete out of order" -> vmcnt(0)), so all outputs
//     leave through inline-asm stores, which hipcc does not see (each ends with the s_nop 1 its hazard recognizer would add).
//   * the epilogue of tile t - 1 (bias, stores) rides in tile t's phase of pass B; pass A's four tiles finish in pass B's first phases.
//   * (measured and rejected: GELU in the sigmoid form x / (1 + 2^(x (ca + cb x^2))) — 7 instructions instead of the 11-FMA
//     polynomial, but two of them transcendental: phase C 1740 instead of 1625 cycles per phase.)
//
// Weight stream: 80 chunks of 32 KB; chunks 0..63 as tl2 (W1 tiles / W2 K chunks interleaved); chunks 64 + pp (pass A, pp = 0..3):
// fragments ((kcl * 4 + otl) * 2 + ks) = W3'[32 otl + n][32 (4 pp + kcl) + 16 ks + 8 hh ..], kcl, otl = 0..3; chunk 64 + t, t = 4..15:
// W3' tile t in fragment order (as tl2).
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "dsh_common.h"
#include "dsh_kernels.h"
#include "tl_common.h"

namespace dsh {

namespace {

constexpr int F3_CH = 32 * 1024;
constexpr int F3_NQ = 80;
constexpr int F3_MAXCLIP = 3;                    // clips a 128-token block may span (frames >= 64)
// LDS: [4][32 KB] ring | folded FiLM rows of up to 3 clips | b1 [1024] | b2 [512] | b3 [512] | b3 + row_const [512] | bias of the fused first stage [512]
constexpr int F3_LDS = 4 * F3_CH + F3_MAXCLIP * 4096 + (1024 + 4 * 512) * 4;   // (+ the bias of the fused attention-branch stage)

typedef f32x4 f32x4_t;

}  // namespace

// probe record (PROBE instantiation, 8 words per block, shader cycles since block start unless noted):
//   [0] end of prologue | [1] end of phase C | [2] end of row statistics + first conversion | [3] end of pass A | [4] end of pass B
//   [5] end of block | [6] 100 MHz ticks of the whole block | [7] cycles spent at the phase tops (counted wait + barrier) of pass B
// HL: the residual stream as two bf16 planes (tl_common.h): the residual is p.Rhi + p.Rlo, the result leaves as p.Ct (hi) + p.Clo
// (lo), p.R / p.Cf are not touched.  Every Linear3 accumulator then starts from zero and meets its residual in the epilogue (the raw
// fragments are requested one phase ahead of it): 4 loads + 4 stores of 1 KB per tile and wave instead of 4 + 6.
namespace {
#ifndef F3_KH0
#define F3_KH0 6
#endif
constexpr int f3_kh0(int pb) { return (pb & 1) ? F3_KH0 : 16; }          // first Linear3 tile whose hi residual fragments are the kept input
constexpr int f3_nraw(int pb, int t) { return t < f3_kh0(pb) ? 4 : 2; }    // loads of one tile's raw residual fragments
constexpr int f3_nres(bool hl, int pb, int q) {   // register-destination loads issued in stream phase q besides its 8 DMA pieces
    // (pb: bit 0 = the hi plane of tiles >= f3_kh0 is not loaded, bit 1 = pass-B tiles requested one phase earlier)
    if (!hl) return (q >= 66 && q <= 69) ? 8 : (q >= 70 && q <= 77) ? 4 : 0;
    if (q < 67 || q > 79) return 0;
    const bool dp = (pb & 2) != 0;
    if (q == 67) return f3_nraw(pb, 0) + (dp ? f3_nraw(pb, 4) : 0);
    const int tt = q - 68, t = 4 + tt;
    return (dp ? (t + 1 < 16 ? f3_nraw(pb, t + 1) : 0) : f3_nraw(pb, t)) + (tt < 3 ? f3_nraw(pb, tt + 1) : 0);
}
constexpr int f3_ny(bool hl, int pb, int q) { return 16 + f3_nres(hl, pb, q - 2) + f3_nres(hl, pb, q - 1); }
}  // namespace

// STY (round 5): the StylizationBlock of the ATTENTION branch (sa_block.proj_out: h <- h + Linear(SiLU(LN(y) (1 + scale) + shift)), models/transformer.py:86-97,130)
// as a first stage of this launch.  Its 16 output tiles run tile-outer on the rolling loop of tl2_linear_kernel<512, 2, ..., ROLL, HL> — the
// same arithmetic operation for operation — and the hi plane of each finished tile IS one fragment pair of the FFN's input: the FFN never loads
// X, the attention branch's output never makes a round trip as a separate launch.  The new residual (hi / lo) is written once and read again
// by the last stage.  Weight stream: 16 chunks (the block's Linear, tile by tile in fragment order) in front of the FFN's 80.
// PB (round 5, DSH_FFN_PB): variants of the last stage.  Bit 0 (default on): in the denoiser's layers the hi plane of the residual IS
// this kernel's input (p.X == p.Rhi), whose fragments are the stationary GEMM1 operand; those of Linear3 tiles F3_KH0 .. 15 stay in
// registers through the last stage, which then loads only their lo plane: 4.4 KB instead of 5 per token through a stage that runs at
// the HBM wall (pass B 32.9 k -> 29.4 k cycles, profiles/r05_o_ffn_pb_timeline.txt).  All 16 tiles' fragments do not fit next to
// y2 + the pass-A accumulators: 43 registers went to scratch and every reload drained the load queue (pass B 31.3 k -> 37.0 k,
// profiles/r05_n_ffn_kh_timeline.txt); from tile 6 on hipcc allocates it without a spill in the loops.  Bit 1 (measured, off): the
// residual fragments of a pass-B tile requested two phases ahead of its epilogue instead of one (three buffers) — pass B 32.9 k ->
// 35.9 k cycles: the stage is bandwidth-bound, more requests in flight only lengthen the queue.
template <bool PROBE, bool HL, int PC, bool STY = false, int PB = 0>
__global__ __launch_bounds__(256, 1) void tl3_ffn_kernel(Tl2FfnArgs p) {
    static_assert(!STY || (HL && PC == 1), "the fused attention-branch stage exists for the hi / lo plane form with the pipelined phase C");
    static_assert(PB == 0 || (HL && PC == 1 && !STY), "the last-stage variants exist for the hi / lo plane form with the pipelined phase C");
    constexpr bool DP = (PB & 2) != 0;
    constexpr int KH0 = f3_kh0(PB);                       // tiles >= KH0 take the hi plane of their residual from hfr (16: none)
    constexpr int QOFF = STY ? 16 : 0;                     // chunks in front of the FFN's own 80
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned long long pc0 = PROBE ? __builtin_readcyclecounter() : 0, pw0 = PROBE ? wall_clock64() : 0;
    unsigned long long pst[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long ptopB = 0;                            // PROBE: cycles spent at the 12 phase tops of pass B (counted wait + barrier)
    trace_mark(p.trace, 0);
    start_stagger(p.stag_groups, p.stag_sleep);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ml = lane & 31, h = lane >> 5;
    const int bx = tl_block_index(p.rev);
    const int tb = bx * (TL_TOK / 32) + wave;
    const int row = tb * 32 + ml;
    const int lane_off = ml * 32 + h * 16;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.Wffn), 0, (F3_NQ + QOFF) * F3_CH, 0x00020000);
    const int wvoff = wave * (F3_CH / 4) + lane * 16;              // this lane's position inside every chunk
    char* wdst = smem + wave * (F3_CH / 4);
    // q counts the FFN's own chunks (the ring slot is q & 3 either way: QOFF is a multiple of 4)
    auto dma_soff = [&](int q) -> int { return ((q < F3_NQ ? q : F3_NQ - 1) + QOFF) * F3_CH; };
    auto dma_dst = [&](int q) -> char* { return wdst + (q & 3) * F3_CH; };
    auto dma_chunk = [&](int q) {                         // the first chunks of the launch: absolute index q
#pragma unroll
        for (int k = 0; k < 8; ++k) dma_buf(k, wrsrc, wvoff, q * F3_CH, dma_dst(q));
    };
    dma_chunk(0);
    dma_chunk(1);
    // folded FiLM rows (A | B) of this block's clips
    f32x4 prm[F3_MAXCLIP];
    int clip0;
    {
        const int rb = bx * TL_TOK, rrb = rb >= p.half_row0 ? rb - p.half_row0 : rb;
        clip0 = rrb / p.frames;
        const int nclip = (rrb + TL_TOK - 1) / p.frames - clip0 + 1;
#pragma unroll
        for (int c = 0; c < F3_MAXCLIP; ++c) {
            const int cc = c < nclip ? c : nclip - 1;
            prm[c] = *reinterpret_cast<const f32x4*>(p.film + (size_t)((clip0 + cc) % p.bmod) * p.film_ld + p.film_off + tid * 4);
        }
    }
    f32x4 prm1[F3_MAXCLIP];                                 // STY: the folded FiLM rows of the attention branch's block
    if (STY) {
        const int rb = bx * TL_TOK, rrb = rb >= p.half_row0 ? rb - p.half_row0 : rb;
        const int nclip = (rrb + TL_TOK - 1) / p.frames - clip0 + 1;
#pragma unroll
        for (int c = 0; c < F3_MAXCLIP; ++c) {
            const int cc = c < nclip ? c : nclip - 1;
            prm1[c] = *reinterpret_cast<const f32x4*>(p.film + (size_t)((clip0 + cc) % p.bmod) * p.film_ld + p.film_off1 + tid * 4);
        }
    }
    float tb1[4], tb2[2], tb3[2], tbc[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) tb1[i] = p.b1[tid + 256 * i];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        tb2[i] = p.b2[tid + 256 * i];
        tb3[i] = p.b3[tid + 256 * i];
        tbc[i] = p.row_const ? p.row_const[tid + 256 * i] : 0.f;
    }
    float tbs[2] = {0.f, 0.f};
    if (STY) {
#pragma unroll
        for (int i = 0; i < 2; ++i) tbs[i] = p.bs1[tid + 256 * i];
    }
    u32x4 hfr[32];                                          // the FFN's input rows (STY: produced by the first stage), yin: the attention output rows
    u32x4 yin[STY ? 32 : 1];
    {
        const char* xr = reinterpret_cast<const char*>(STY ? p.Y : p.X) + (size_t)tb * 32 * 1024 + lane_off;
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(xr + s * 1024);
            if constexpr (STY) yin[s] = v; else hfr[s] = v;
        }
    }
    float* sprm = reinterpret_cast<float*>(smem + 4 * F3_CH);
    float* sb1 = sprm + F3_MAXCLIP * 1024;
    float* sb2 = sb1 + 1024;
    float* sb3 = sb2 + 512;
    float* sb3c = sb3 + 512;
    float* sbs1 = sb3c + 512;
#pragma unroll
    for (int c = 0; c < F3_MAXCLIP; ++c) *reinterpret_cast<f32x4*>(sprm + 1024 * c + 4 * tid) = STY ? prm1[c] : prm[c];
    if (STY) {
#pragma unroll
        for (int i = 0; i < 2; ++i) sbs1[tid + 256 * i] = tbs[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) sb1[tid + 256 * i] = tb1[i];
#pragma unroll
    for (int i = 0; i < 2; ++i) { sb2[tid + 256 * i] = tb2[i]; sb3[tid + 256 * i] = tb3[i]; sb3c[tid + 256 * i] = tb3[i] + tbc[i]; }
    __syncthreads();                                        // bias tables visible (the row loads are still in flight)
    f32x16 acc2[16];
    auto init_acc2 = [&]() {
#pragma unroll
        for (int ot = 0; ot < 16; ++ot)
#pragma unroll
            for (int qi = 0; qi < 4; ++qi) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(sb2 + ot * 32 + 16 * (qi >> 1) + 8 * h + 4 * (qi & 1));
#pragma unroll
                for (int e = 0; e < 4; ++e) acc2[ot][4 * qi + e] = b4[e];
            }
    };
    if (!STY) init_acc2();                                  // (STY: after the first stage — until then the accumulator file holds its tiles and the parked hi fragments)
#pragma unroll
    for (int s = 0; s < 32; ++s) { if constexpr (STY) asm volatile("" ::"v"(yin[s])); else asm volatile("" ::"v"(hfr[s])); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // rows and the first two chunks have landed
    const char* lds_lane = smem + lane * 16;
    if constexpr (STY) {
        // LayerNorm -> folded FiLM -> SiLU of the attention output rows, in place (the register prologue of the StylizationBlock
        // instantiation of tl2_linear_kernel, tl_common.h ln_frags)
        const int rr = row >= p.half_row0 ? row - p.half_row0 : row;
        int ci = rr / p.frames - clip0;
        ci = ci < F3_MAXCLIP ? ci : F3_MAXCLIP - 1;
        const float* ca1 = sprm + ci * 1024 + 8 * h;
        ln_frags<32, true>(yin, ca1, ca1 + 512, 512.f, 512.f);
        __syncthreads();                                    // every wave is done with the attention branch's FiLM rows (and chunks 0, 1 are visible)
#pragma unroll
        for (int c = 0; c < F3_MAXCLIP; ++c) *reinterpret_cast<f32x4*>(sprm + 1024 * c + 4 * tid) = prm[c];   // the FFN branch's rows: read from pass A on
    } else {
        __syncthreads();                                    // ... for every wave
    }
    dma_chunk(2);
    trace_mark(p.trace, 1);

    if constexpr (STY) {
        // ---- stage S: 16 output tiles of the attention branch's StylizationBlock Linear, tile-outer, one tile per phase (chunk t in ring
        //      slot t & 3), on the rolling slots of tl2_linear_kernel<512, 2, true, 3, ACT_NONE, false, true, true>: MFMA, fragment read 4
        //      slots ahead, one DMA piece every fourth slot (chunk t + 3: from t = 13 on these are the FFN's first chunks), the epilogue of
        //      tile t - 1 — accumulator started from the bias, + hi + lo of the residual (requested in slot 0 of tile t - 1's own phase),
        //      split into planes, stored — in slots 2 .. 24 and 28, the bias of tile t + 1 in slots 28 .. 31, counted wait + barrier
        //      behind slot 27.  The hi fragments are the FFN's input rows: parked in the accumulator file until phase C.
        typedef __attribute__((address_space(3))) const char* lcp_t;
        typedef __attribute__((address_space(3))) const u32x4* lfr_t;
        auto s_base = [&](int a) -> lcp_t { lcp_t b = (lcp_t)lds_lane + (a & 3) * F3_CH; asm volatile("" : "+v"(b)); return b; };
        u32x4 sw[2][4];
        f32x16 sA, sB;
        struct SRes { u32x4 hi[2], lo[2]; };
        SRes rA, rB;
        // (buffer loads: one 32-bit lane offset + a per-fragment scalar offset.  With 64-bit flat addresses hipcc computed the 64 fragment
        //  addresses of the stage up front and spilled them — 217 registers)
        const size_t plane_bytes = (size_t)((p.M + TL_TOK - 1) / TL_TOK) * TL_TOK * 1024;
        const __amdgpu_buffer_rsrc_t rh_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.Rhi), 0, (int)plane_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rl_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.Rlo), 0, (int)plane_bytes, 0x00020000);
        const size_t sbase = (size_t)tb * 32 * 1024 + lane_off;             // byte offset of fragment 0 of this wave's rows in a bf16 plane
        auto s_load_res = [&](SRes& r, int t) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                r.hi[c] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rh_rsrc, (int)sbase, (2 * t + c) * 1024, 0));
                r.lo[c] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rl_rsrc, (int)sbase, (2 * t + c) * 1024, 0));
            }
        };
        auto s_bias = [&](f32x16& a, int t, int qi) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(sbs1 + t * 32 + 16 * (qi >> 1) + 8 * h + 4 * (qi & 1));
#pragma unroll
            for (int e = 0; e < 4; ++e) a[4 * qi + e] = b4[e];
        };
        // loads younger than chunk t + 1's DMA behind slot 27 of phase t: the residual fragments and DMA pieces of phase t - 1 (4 + 8) and
        // of this phase so far (4 + 7)
        auto s_mid = [&]() {
            asm volatile("s_waitcnt vmcnt(23)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        };
        struct SEpi { float v[8]; };
        const unsigned svo = (unsigned)sbase;
        // step k = 6 qi + step of the epilogue of tile te (accumulator E, residual rs): steps 1..4 take one value each, step 5 of an odd
        // quad finishes the fragment c = qi >> 1
        auto s_epi = [&](auto te_tag, const f32x16& E, const SRes& rs, SEpi& st, auto k_tag) {
            constexpr int te = decltype(te_tag)::value, k = decltype(k_tag)::value, qi = k / 6, step = k % 6;
            if constexpr (step >= 1 && step <= 4) {
                float x = E[4 * qi + step - 1];
                asm volatile("" : "+v"(x));
                st.v[4 * (qi & 1) + step - 1] = x;
            } else if constexpr (step == 5 && (qi & 1) == 1) {
                constexpr int c = qi >> 1;
                hl_accumulate(st.v, rs.hi[c], rs.lo[c]);
                u32x4 oh, ol;
                hl_split(st.v, oh, ol);
                const unsigned vo = svo + (unsigned)(2 * te + c) * 1024u;
                asm_store16<0>(p.Ct, vo, oh);
                asm_store16<0>(p.Clo, vo, ol);
                hfr[2 * te + c] = oh;
                asm volatile("" : "+a"(hfr[2 * te + c]));
            }
        };
        auto s_tile = [&](auto t_tag, f32x16& W, f32x16& E, SRes& rw, const SRes& re) {
            constexpr int t = decltype(t_tag)::value;
            SEpi st;
            s_load_res(rw, t);
            const lcp_t cur = s_base(t), nxt = s_base(t + 1);
            const int so_next = (t + 3) * F3_CH;
            char* dst_next = wdst + ((t + 3) & 3) * F3_CH;
            static_for<32>([&](auto m_tag) {
                constexpr int m = decltype(m_tag)::value;
                W = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, sw[(m >> 2) & 1][m & 3]), __builtin_bit_cast(bf16x8, yin[m]), W, 0, 0, 0);
                if constexpr (m < 24) sw[((m + 4) >> 2) & 1][(m + 4) & 3] = *(lfr_t)(cur + (m + 4) * 1024);
                if constexpr (m >= 20 && m < 24) sw[((m + 8) >> 2) & 1][(m + 8) & 3] = *(lfr_t)(cur + (m + 8) * 1024);
                if constexpr (m >= 28 && t < 15) sw[0][m - 28] = *(lfr_t)(nxt + (m - 28) * 1024);
                if constexpr ((m & 3) == 1) dma_buf(m >> 2, wrsrc, wvoff, so_next, dst_next);
                if constexpr (t > 0 && m >= 2 && m <= 24) s_epi(std::integral_constant<int, (t > 0 ? t - 1 : 0)>{}, E, re, st, std::integral_constant<int, m - 2>{});
                if constexpr (t > 0 && m == 28) s_epi(std::integral_constant<int, (t > 0 ? t - 1 : 0)>{}, E, re, st, std::integral_constant<int, 23>{});
                if constexpr (m >= 28 && t < 15) s_bias(E, t + 1, m - 28);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (m == 27) s_mid();
            });
        };
#pragma unroll
        for (int qi = 0; qi < 4; ++qi) s_bias(sA, 0, qi);
#pragma unroll
        for (int i = 0; i < 4; ++i) sw[0][i] = *reinterpret_cast<const u32x4*>(lds_lane + i * 1024);
        __builtin_amdgcn_sched_barrier(0);
        static_for<8>([&](auto tt) {
            constexpr int t2 = decltype(tt)::value * 2;
            s_tile(std::integral_constant<int, t2>{}, sA, sB, rA, rB);
            s_tile(std::integral_constant<int, t2 + 1>{}, sB, sA, rB, rA);
        });
        {   // the last tile's epilogue has no phase of this stage left to ride in: exposed once per block
            SEpi st;
            static_for<24>([&](auto k_tag) { s_epi(std::integral_constant<int, 15>{}, sB, rB, st, k_tag); });
        }
#pragma unroll
        for (int s2 = 0; s2 < 32; ++s2) asm volatile("" : "+v"(hfr[s2]));      // back into the VGPRs: operands of the GEMM1 chain
        init_acc2();
    }
    if (PROBE) pst[0] = __builtin_readcyclecounter() - pc0;

    // the top of a phase: this wave's share of chunk q has landed (NY loads younger than it may be in flight), then everybody's
#define F3_PHASE_TOP(NY)                                               \
    do {                                                               \
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NY) : "memory");      \
        __builtin_amdgcn_s_barrier();                                  \
        asm volatile("" ::: "memory");                                 \
    } while (0)

    // ---- phase C (as tl2_ffn_kernel: GEMM1(j) with the GELU of tile j - 1 in its MFMA shadow, GEMM2(j - 1) into the 16 resident
    //      accumulators) ---------------------------------------------------------------------------------------------------------
    f32x16 hprev;                                           // newest hidden tile (pre-activation, bias included)
    u32x4 gfr[2];                                           // GELU(hidden tile) as two B fragments (k steps 0 / 1 of a GEMM2 chunk)
    u32x4 gnx0;                                             // first fragment of the NEXT tile's GELU, built during GEMM2
#pragma unroll
    for (int e = 0; e < 16; ++e) hprev[e] = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c) { gfr[c][0] = 0; gfr[c][1] = 0; gfr[c][2] = 0; gfr[c][3] = 0; }
    gnx0[0] = 0; gnx0[1] = 0; gnx0[2] = 0; gnx0[3] = 0;
    auto pack8 = [&](const float* v) -> u32x4 {
        u32x4 o;
        o[0] = pack_bf16(v[0], v[1]); o[1] = pack_bf16(v[2], v[3]); o[2] = pack_bf16(v[4], v[5]); o[3] = pack_bf16(v[6], v[7]);
        return o;
    };
    auto gelu_here = [&](float x) -> float { float y = gelu_fast(x); asm volatile("" : "+v"(y)); return y; };
    // issue pattern of a phase (one scheduling region): every MFMA is followed by one A-fragment read, NV VALU instructions and,
    // every fourth time, one vector-memory instruction
    auto phase_pattern = [&](auto nv_tag) {
        constexpr int NV = decltype(nv_tag)::value;
#pragma unroll
        for (int m = 0; m < 32; ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (m < 28) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if ((m & 3) == 1) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    typedef std::integral_constant<int, 2> I2;
    typedef std::integral_constant<int, 3> I3;
    // residual tile t (lane-native fp32, four 1 KB pieces per wave): the initial value of Linear3 accumulator t for pass B's
    // tiles (t >= 4); pass A's four accumulators start from zero and meet their residual (ra) in the epilogue
    f32x16 a3[16], ra[4];
    const size_t fbase = ((size_t)tb * 16 * 4 * 64 + lane) * 4;           // + t * 1024 floats + qi * 256
    auto load_res_into = [&](f32x16& dst, int t) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 r4 = *reinterpret_cast<const f32x4*>(p.R + fbase + (size_t)t * 1024 + q * 256);
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[4 * q + e] = r4[e];
        }
    };
    auto load_res_tile = [&](int t) { if (t < 4) load_res_into(ra[t], t); else load_res_into(a3[t], t); };
    // HL: the raw hi / lo fragments of a tile's residual (two of each), requested one phase before the tile's epilogue
    struct RawRes { u32x4 hi[2], lo[2]; };
    RawRes rawA[2], rawB[3];
    constexpr int NRB = DP ? 3 : 2;                                      // buffers of the pass-B tiles in use
    const size_t pbase = (size_t)tb * 32 * 1024 + lane_off;             // byte offset of fragment 0 in a bf16 plane
    // (STY: buffer loads — one 32-bit lane offset and a scalar offset per fragment.  With flat 64-bit addresses hipcc computes the 48
    //  fragment addresses of pass B at the top of the kernel; next to the first stage's registers they were spilled there)
    const int plane_bytes_i = (int)((size_t)((p.M + TL_TOK - 1) / TL_TOK) * TL_TOK * 1024);
    const __amdgpu_buffer_rsrc_t rawh_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.Rhi), 0, STY ? plane_bytes_i : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rawl_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.Rlo), 0, STY ? plane_bytes_i : 0, 0x00020000);
    auto load_raw = [&](RawRes& r, int t) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if constexpr (STY) {
                r.hi[c] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rawh_rsrc, (int)pbase, (2 * t + c) * 1024, 0));
                r.lo[c] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rawl_rsrc, (int)pbase, (2 * t + c) * 1024, 0));
            } else {
                if (t < KH0) r.hi[c] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(p.Rhi) + pbase + (size_t)(2 * t + c) * 1024);
                r.lo[c] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(p.Rlo) + pbase + (size_t)(2 * t + c) * 1024);
            }
        }
    };
#pragma unroll
    for (int t = 0; t < (HL ? 16 : 4); ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) a3[t][e] = 0.f;
    // GEMM1 phase q on hidden tile j.  GMODE 0: no GELU rides along; 1: the whole GELU of the previous hidden tile; 2: its second
    // half (values 8 .. 15 -> gfr[1]; the first half was built during the preceding GEMM2 phase -> gnx0)
    auto gemm1 = [&](int q, int j, auto gmode_tag) {
        constexpr int GMODE = decltype(gmode_tag)::value;
        F3_PHASE_TOP(16);
        const int so_next = dma_soff(q + 3);
        char* dst_next = dma_dst(q + 3);
        f32x16 acc1;
#pragma unroll
        for (int qi = 0; qi < 4; ++qi) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(sb1 + j * 32 + 16 * (qi >> 1) + 8 * h + 4 * (qi & 1));
#pragma unroll
            for (int e = 0; e < 4; ++e) acc1[4 * qi + e] = b4[e];
        }
        const char* cur = lds_lane + (q & 3) * F3_CH;
        u32x4 aw[2][4];
        float gv[16];
#pragma unroll
        for (int i = 0; i < 4; ++i) aw[0][i] = *reinterpret_cast<const u32x4*>(cur + i * 1024);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (g + 1 < 8) {
#pragma unroll
                for (int i = 0; i < 4; ++i) aw[(g + 1) & 1][i] = *reinterpret_cast<const u32x4*>(cur + ((g + 1) * 4 + i) * 1024);
            }
            dma_buf(g, wrsrc, wvoff, so_next, dst_next);
            if (GMODE == 1) { gv[2 * g] = gelu_here(hprev[2 * g]); gv[2 * g + 1] = gelu_here(hprev[2 * g + 1]); }
            if (GMODE == 2) gv[8 + g] = gelu_here(hprev[8 + g]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aw[g & 1][i]), __builtin_bit_cast(bf16x8, hfr[g * 4 + i]), acc1, 0, 0, 0);
        }
        phase_pattern(I3{});
        if (GMODE == 1) { gfr[0] = pack8(gv); gfr[1] = pack8(gv + 8); }
        if (GMODE == 2) { gfr[0] = gnx0; gfr[1] = pack8(gv + 8); }
        hprev = acc1;
        asm volatile("" : "+v"(hprev));                     // the accumulator read happens HERE (MFMA wait states in straight-line code)
    };
    // GEMM2 phase q: K chunk (32 hidden features, gfr) into the 16 resident accumulators.  WITH_HALF: the first half of the GELU of
    // the newest hidden tile rides along -> gnx0.  RES 1: this phase also requests the residual of pass A's tiles 0..3
    auto gemm2 = [&](int q, auto half_tag, auto ny_tag, auto res_tag) {
        constexpr bool WITH_HALF = decltype(half_tag)::value;
        constexpr int NY = decltype(ny_tag)::value, RES = decltype(res_tag)::value;
        F3_PHASE_TOP(NY);
        const int so_next = dma_soff(q + 3);
        char* dst_next = dma_dst(q + 3);
        const char* cur = lds_lane + (q & 3) * F3_CH;
        u32x4 aw[2][4];
        float gn[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) aw[0][i] = *reinterpret_cast<const u32x4*>(cur + i * 1024);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 8; ++g) {                       // group g: output tiles 2 g, 2 g + 1 (x 2 k steps): fragments 4 g .. 4 g + 3
            if (g + 1 < 8) {
#pragma unroll
                for (int i = 0; i < 4; ++i) aw[(g + 1) & 1][i] = *reinterpret_cast<const u32x4*>(cur + ((g + 1) * 4 + i) * 1024);
            }
            if (RES != 0 && g < 4) load_res_tile(g);
            dma_buf(g, wrsrc, wvoff, so_next, dst_next);
            if (WITH_HALF) gn[g] = gelu_here(hprev[g]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc2[2 * g + (i >> 1)] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aw[g & 1][i]), __builtin_bit_cast(bf16x8, gfr[i & 1]),
                                                                                  acc2[2 * g + (i >> 1)], 0, 0, 0);
            if (RES != 0) __builtin_amdgcn_sched_barrier(0);
        }
        if (RES == 0) phase_pattern(I3{});
        if (WITH_HALF) gnx0 = pack8(gn);
    };
    typedef std::integral_constant<int, 16> N16;
    if constexpr (PC == 0) {
    gemm1(0, 0, I0{});
    gemm1(1, 1, I1{});                                      // GELU(0) -> gfr
    gemm2(2, std::true_type{}, N16{}, I0{});                // consumes hidden tile 0; first half of GELU(1)
    for (int j = 2; j < 31; ++j) {
        gemm1(2 * j - 1, j, I2{});                          // second half of GELU(j - 1) -> gfr
        gemm2(2 * j, std::true_type{}, N16{}, I0{});        // consumes hidden tile j - 1; first half of GELU(j)
    }
    gemm1(61, 31, I2{});                                    // the last GEMM1: the h16 fragments are dead from here on
    gemm2(62, std::true_type{}, N16{}, I0{});
    {   // the second half of the last hidden tile's GELU has no GEMM1 left to hide under: exposed once per block
        float gv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) gv[e] = gelu_fast(hprev[8 + e]);
        gfr[0] = gnx0; gfr[1] = pack8(gv);
    }
    gemm2(63, std::false_type{}, N16{}, I0{});
    } else {
        // ---- phase C, pipelined (round 5).  The arithmetic and its order are those of the loop above; what changes is where a phase
        //      ends and who places the instructions.  Round-4 disassembly of the loop above: at the top of every GEMM1 phase
        //      barrier -> 16 v_accvgpr_read (hipcc selects the AGPR form for every MFMA of a 512-register kernel, all 256 AGPRs hold
        //      y2, so one y2 tile is moved out and back around every hidden tile) -> bias + first four fragment reads -> LDS latency ->
        //      first MFMA; behind its last MFMA s_nop + 16 v_accvgpr_read (the copy to hprev): ~600 of the 3078 cycles of a phase
        //      pair with the matrix pipe idle.  Here:
        //   * a phase is 32 explicit ISSUE SLOTS — one MFMA, the fragment read 4 slots ahead, every fourth slot one DMA piece, one
        //     three-instruction stage of the GELU polynomial — fenced by sched_barrier(0): source order is issue order;
        //   * the A fragments roll ACROSS the phase boundary: slots 28..31 of phase q read the first four fragments of chunk q + 1,
        //     so the first MFMA of the next phase finds its operand in registers;
        //   * the counted wait + barrier that publish chunk q + 1 therefore sit behind slot 27 of phase q (seven of this phase's DMA
        //     pieces issued: vmcnt(8 + 7)); the lgkmcnt(0) in front of the barrier retires this wave's reads of chunk q, whose ring
        //     slot the next phase's DMA refills — the last four of them are issued by slot 23 (two reads per slot in 20..23);
        //   * the hidden tile alternates between two accumulators that live in VGPRs (the GEMM1 chain is the VGPR form of the MFMA,
        //     spelled in inline asm: A from registers hipcc loaded — its own lgkmcnt bookkeeping covers the operands —, C / D tied;
        //     consecutive MFMAs of one chain need no wait states, and the first VALU read of a finished tile is a whole phase away):
        //     the GELU reads the finished tile in place while the next one accumulates, no accumulator moves; the bias of tile j + 1
        //     is read into its accumulator during the last slots of the GEMM2 phase in front of it;
        //   * the GELU'd fragments alternate between two register pairs as well (no copies).
        f32x16 accA, accB;                                  // hidden tiles (pre-activation, bias included): even j -> accA, odd j -> accB
        u32x4 GA[2], GB[2];                                 // GELU(hidden tile) as two B fragments: even j -> GA, odd j -> GB
        u32x4 aw[2][4];                                     // rolling A fragments: at the top of a phase aw[0] = fragments 0..3 of its chunk
#pragma unroll
        for (int c = 0; c < 2; ++c) { GA[c] = gfr[c]; GB[c] = gfr[c]; }
        auto bias_quad = [&](f32x16& a, int j, int qi) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(sb1 + j * 32 + 16 * (qi >> 1) + 8 * h + 4 * (qi & 1));
#pragma unroll
            for (int e = 0; e < 4; ++e) a[4 * qi + e] = b4[e];
        };
        // one stage (0..3) of GELU(x) = x (1/2 + xc h(xc^2)) (gelu_fast, tl_common.h, operation for operation); y is valid after stage 3
        struct GeluPipe { float xc, u, hh; };
        auto gelu_stage = [&](int st, float x, GeluPipe& s, float& y) {
            if (st == 0) {
                s.xc = __builtin_amdgcn_fmed3f(x, -4.25f, 4.25f);
                s.u = s.xc * s.xc;
                s.hh = fmaf(-8.460346867522617e-10f, s.u, 7.570786664246043e-08f);
            } else if (st == 1) {
                s.hh = fmaf(s.hh, s.u, -2.938788611572818e-06f);
                s.hh = fmaf(s.hh, s.u, 6.552687409566715e-05f);
                s.hh = fmaf(s.hh, s.u, -0.0009404457523487508f);
            } else if (st == 2) {
                s.hh = fmaf(s.hh, s.u, 0.009257814846932888f);
                s.hh = fmaf(s.hh, s.u, -0.06545348465442657f);
                s.hh = fmaf(s.hh, s.u, 0.3984200358390808f);
            } else {
                y = x * fmaf(s.xc, s.hh, 0.5f);
                asm volatile("" : "+v"(y));
            }
            // (opaque per stage: the SLP vectoriser otherwise pairs the polynomials of two values into v_pk_fma_f32 chains that land
            //  in ONE slot with an s_nop between every two dependent packed operations — round-5 disassembly)
            if (st < 3) asm volatile("" : "+v"(s.hh));
        };
        auto mid_barrier = [&]() {
            asm volatile("s_waitcnt vmcnt(15)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        };
        // the fragment reads of slot m: fragment m + 4 of this chunk (two per slot in 20..23, none in 24..27), then the next chunk's
        // first four behind the barrier.  Fragment f lives in aw[(f >> 2) & 1][f & 3]; its register is free once MFMA f - 8 has issued
        typedef __attribute__((address_space(3))) const char* lcptr_t;
        typedef __attribute__((address_space(3))) const u32x4* lfrag_t;
        // (the chunk base goes through an opaque register: one address VGPR + immediate offsets per phase.  With the ring slot a
        //  compile-time constant inside the unrolled loop body hipcc otherwise materialises an address register per fragment)
        auto chunk_base = [&](int q) -> lcptr_t { lcptr_t b = (lcptr_t)lds_lane + (q & 3) * F3_CH; asm volatile("" : "+v"(b)); return b; };
        auto slot_reads = [&](int m, lcptr_t cur, lcptr_t nxt, bool last) {
            auto rd = [&](int f) { aw[(f >> 2) & 1][f & 3] = *(lfrag_t)(cur + f * 1024); };
            if (m < 24) rd(m + 4);
            if (m >= 20 && m < 24) rd(m + 8);
            if (m >= 28 && !last) aw[0][m - 28] = *(lfrag_t)(nxt + (m - 28) * 1024);
        };
        // GEMM1 phase q on a hidden tile into W; R = the previous hidden tile, G its GELU'd fragment pair.  GMODE 0: no GELU rides
        // along; 1: the whole GELU of R (two stages per slot); 2: its second half (values 8 .. 15 -> G[1]; the first half was built
        // during the preceding GEMM2 phase)
        auto gemm1x = [&](int q, f32x16& W, const f32x16& R, u32x4 (&G)[2], auto gmode_tag) {
            constexpr int GMODE = decltype(gmode_tag)::value;
            const int so_next = dma_soff(q + 3);
            char* dst_next = dma_dst(q + 3);
            const lcptr_t cur = chunk_base(q), nxt = chunk_base(q + 1);
            GeluPipe gp = {0.f, 0.f, 0.f};
            float yv[2] = {0.f, 0.f};
#pragma unroll
            for (int m = 0; m < 32; ++m) {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(W) : "v"(aw[(m >> 2) & 1][m & 3]), "v"(hfr[m]));
                slot_reads(m, cur, nxt, false);
                if ((m & 3) == 1) dma_buf(m >> 2, wrsrc, wvoff, so_next, dst_next);
                if (GMODE == 2) {
                    const int v = 8 + (m >> 2);                               // value index inside the tile
                    gelu_stage(m & 3, R[v], gp, yv[v & 1]);
                    if ((m & 7) == 7) G[1][(v - 8) >> 1] = pack_bf16(yv[0], yv[1]);
                }
                if (GMODE == 1) {
                    // R finished with the LAST slot of the phase before (an asm MFMA: hipcc's hazard recognizer does not see it): its
                    // first VALU read waits two slots (two MFMA issue times > the 8-pass write-back); 64 stage steps over slots 2..31
                    const int t0 = m < 2 ? 0 : (m < 28 ? 2 * (m - 2) : 52 + 3 * (m - 28));
                    const int t1 = m < 2 ? 0 : (m < 28 ? t0 + 2 : t0 + 3);
#pragma unroll
                    for (int t = t0; t < t1; ++t) {
                        const int v = t >> 2;
                        gelu_stage(t & 3, R[v], gp, yv[v & 1]);
                        if ((t & 7) == 7) G[v >> 3][(v & 7) >> 1] = pack_bf16(yv[0], yv[1]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (m == 27) mid_barrier();
            }
        };
        // GEMM2 phase q: K chunk (32 hidden features, the fragment pair G) into the 16 resident accumulators.  WITH_HALF: the first
        // half of the GELU of the newest hidden tile R rides along -> GN[0].  INIT: the bias of hidden tile jinit is read into its
        // accumulator NI (slots 28..31).  LAST: phase 63 — the next phase (pass A) opens with its own wait + barrier + fragment reads
        auto gemm2x = [&](int q, const u32x4 (&G)[2], const f32x16& R, u32x4 (&GN)[2], f32x16& NI, int jinit, auto half_tag, auto init_tag, auto last_tag) {
            constexpr bool WITH_HALF = decltype(half_tag)::value, INIT = decltype(init_tag)::value, LAST = decltype(last_tag)::value;
            const int so_next = dma_soff(q + 3);
            char* dst_next = dma_dst(q + 3);
            const lcptr_t cur = chunk_base(q), nxt = chunk_base(q + 1);
            GeluPipe gp = {0.f, 0.f, 0.f};
            float yv[2] = {0.f, 0.f};
#pragma unroll
            for (int m = 0; m < 32; ++m) {                  // slot m: output tile m >> 1, k step m & 1
                acc2[m >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aw[(m >> 2) & 1][m & 3]), __builtin_bit_cast(bf16x8, G[m & 1]),
                                                                       acc2[m >> 1], 0, 0, 0);
                slot_reads(m, cur, nxt, LAST);
                if ((m & 3) == 1) dma_buf(m >> 2, wrsrc, wvoff, so_next, dst_next);
                if (WITH_HALF) {
                    // R finished with the last slot of the GEMM1 phase before (asm MFMAs, invisible to the hazard recognizer): the first
                    // VALU read of it waits two slots; the 32 stage steps run in slots 2..31 (two each in the last two)
                    const int t0 = m < 2 ? 0 : (m < 30 ? m - 2 : 28 + 2 * (m - 30));
                    const int t1 = m < 2 ? 0 : (m < 30 ? t0 + 1 : t0 + 2);
#pragma unroll
                    for (int t = t0; t < t1; ++t) {
                        const int v = t >> 2;
                        gelu_stage(t & 3, R[v], gp, yv[v & 1]);
                        if ((t & 7) == 7) GN[0][v >> 1] = pack_bf16(yv[0], yv[1]);
                    }
                }
                if (INIT && m >= 28) bias_quad(NI, jinit, m - 28);
                __builtin_amdgcn_sched_barrier(0);
                if (m == 27 && !LAST) mid_barrier();
            }
        };
#pragma unroll
        for (int qi = 0; qi < 4; ++qi) { bias_quad(accA, 0, qi); bias_quad(accB, 1, qi); }
#pragma unroll
        for (int i = 0; i < 4; ++i) aw[0][i] = *reinterpret_cast<const u32x4*>(lds_lane + i * 1024);
        __builtin_amdgcn_sched_barrier(0);
        const std::true_type yes{};
        const std::false_type no{};
        gemm1x(0, accA, accB, GB, I0{});                    // hidden tile 0
        gemm1x(1, accB, accA, GA, I1{});                    // tile 1; GELU(0) -> GA
        gemm2x(2, GA, accB, GB, accA, 2, yes, yes, no);     // consumes hidden tile 0; first half of GELU(1) -> GB[0]; bias of tile 2
        for (int jj = 2; jj < 30; jj += 2) {
            gemm1x(2 * jj - 1, accA, accB, GB, I2{});       // tile jj (even); second half of GELU(jj - 1) -> GB[1]
            gemm2x(2 * jj, GB, accA, GA, accB, jj + 1, yes, yes, no);
            gemm1x(2 * jj + 1, accB, accA, GA, I2{});       // tile jj + 1 (odd); second half of GELU(jj) -> GA[1]
            gemm2x(2 * jj + 2, GA, accB, GB, accA, jj + 2, yes, yes, no);
        }
        gemm1x(59, accA, accB, GB, I2{});                   // tile 30
        gemm2x(60, GB, accA, GA, accB, 31, yes, yes, no);
        gemm1x(61, accB, accA, GA, I2{});                   // tile 31, the last GEMM1: the h16 fragments are dead from here on
        gemm2x(62, GA, accB, GB, accA, 0, yes, no, no);     // consumes tile 30; first half of GELU(31) -> GB[0]
        {   // the second half of the last hidden tile's GELU has no GEMM1 left to hide under: exposed once per block
            float gv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) gv[e] = gelu_fast(accB[8 + e]);
            GB[1] = pack8(gv);
        }
        gemm2x(63, GB, accB, GA, accA, 0, no, no, yes);
    }
    if (PROBE) pst[1] = __builtin_readcyclecounter() - pc0;

    // ---- row statistics of y2 (fp32 accumulators), then the conversion y2 tile -> SiLU(LN * (1 + scale) + shift) as two bf16 B
    //      fragments, one tile at a time ------------------------------------------------------------------------------------------
    float rstd, nmr;
    {
        // (the "+a" pin: y2 LIVES in the accumulator file.  Without it hipcc keeps the VGPR copies it reads for the statistics alive
        //  for the conversions below and spills ~100 registers; a pin after the reads, or an opaque copy, spill 31 - 107: measured
        //  by compile only, round 4)
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 sm = {0.f, 0.f}, sq = {0.f, 0.f};       // packed pairs: no MFMA runs beside this stage, v_pk_add / v_pk_fma halve its VALU count
#pragma unroll
        for (int ot = 0; ot < 16; ++ot) {
            asm volatile("" : "+a"(acc2[ot]));
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                const f32x2 v = {acc2[ot][e], acc2[ot][e + 1]};
                sm += v;
                sq = __builtin_elementwise_fma(v, v, sq);
            }
        }
        float sum = sm[0] + sm[1], ssq = sq[0] + sq[1];
        sum += __shfl_xor(sum, 32, 64);
        ssq += __shfl_xor(ssq, 32, 64);
        const float mean = sum * (1.0f / 512.f);
        const float var = fmaxf(ssq * (1.0f / 512.f) - mean * mean, 0.f);
        rstd = 1.0f / sqrtf(var + 1e-5f);
        nmr = -mean * rstd;
    }
    const float* ca;
    {
        const int rr = row >= p.half_row0 ? row - p.half_row0 : row;
        int ci = rr / p.frames - clip0;
        ci = ci < F3_MAXCLIP ? ci : F3_MAXCLIP - 1;
        ca = sprm + ci * 1024 + 8 * h;
    }
    u32x4 yfr[32];
    // one HALF of the conversion of y2 tile ot: fragment s = 2 ot + c holds features 32 ot + 16 c + 8 h + (0..7).  The folded
    // FiLM rows of the half (A: fa, B: fb) are read one step ahead (film_rows) so that no LDS latency sits inside the chain.
    struct FilmRows { f32x4 a0, a1, b0, b1; };
    auto film_rows = [&](int ot, int c) -> FilmRows {
        FilmRows f;
        f.a0 = *reinterpret_cast<const f32x4*>(ca + 32 * ot + 16 * c); f.a1 = *reinterpret_cast<const f32x4*>(ca + 32 * ot + 16 * c + 4);
        f.b0 = *reinterpret_cast<const f32x4*>(ca + 512 + 32 * ot + 16 * c); f.b1 = *reinterpret_cast<const f32x4*>(ca + 512 + 32 * ot + 16 * c + 4);
        return f;
    };
    auto convert_half = [&](int ot, int c, const FilmRows& f) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float t = fmaf(acc2[ot][8 * c + e], rstd, nmr);
            const float y = fmaf(t, e < 4 ? f.a0[e & 3] : f.a1[e & 3], e < 4 ? f.b0[e & 3] : f.b1[e & 3]);
            v[e] = y * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(y * -1.4426950408889634f));
        }
        yfr[2 * ot + c] = pack8(v);
    };
    FilmRows fnext = film_rows(0, 0);
    { const FilmRows f0 = fnext; fnext = film_rows(0, 1); convert_half(0, 0, f0); }
    { const FilmRows f1 = fnext; fnext = film_rows(1, 0); convert_half(0, 1, f1); }
    if (PROBE) pst[2] = __builtin_readcyclecounter() - pc0;

    // vector-memory LOADS issued in phase q (stream position) besides its 8 DMA pieces: residual tiles, each requested two phases
    // before it is needed — phases 66 / 67: tiles 0, 4 / 1, 5; the phase of tile t = 4 + tt (68 + tt): tile t + 2, and for tt < 2 also
    // pass-A tile 2 + tt (whose epilogue rides in phase 68 + 2 + tt).  The counted wait at the top of phase q must leave at most the
    // loads of phases q - 2 and q - 1 in flight (tracked loads stay inside their phase: every phase top is a compiler memory
    // barrier; a smaller count only waits longer).
    // (HL: phase 67 requests pass-A tile 0; the phase of tile t requests tile t itself and, for tt < 3, pass-A tile tt + 1: f3_nres)
#define F3_NY(q) f3_ny(HL, PB, q)
    // ---- pass A: Linear3 output tiles 0..3, K-outer: phase pp = K steps 4 pp .. 4 pp + 3; the conversion of y2 tile kc + 1 and the
    //      residual request of pass-B tile kc + 2 ride in K step kc ------------------------------------------------------------------
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {
        const int q = 64 + pp;
        switch (pp) {
            case 0: F3_PHASE_TOP(F3_NY(64)); break;
            case 1: F3_PHASE_TOP(F3_NY(65)); break;
            case 2: F3_PHASE_TOP(F3_NY(66)); break;
            default: F3_PHASE_TOP(F3_NY(67)); break;
        }
        const int so_next = dma_soff(q + 3);
        char* dst_next = dma_dst(q + 3);
        const char* cur = lds_lane + (q & 3) * F3_CH;
        u32x4 aw[2][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) aw[0][i] = *reinterpret_cast<const u32x4*>(cur + i * 1024);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 8; ++g) {                       // group g: K step kc = 4 pp + (g >> 1), output tiles 2 (g & 1), + 1 (x 2 k steps)
            if (g + 1 < 8) {
#pragma unroll
                for (int i = 0; i < 4; ++i) aw[(g + 1) & 1][i] = *reinterpret_cast<const u32x4*>(cur + ((g + 1) * 4 + i) * 1024);
            }
            const int kc = 4 * pp + (g >> 1);
            if (!HL && g == 0 && pp >= 2) { load_res_tile(pp - 2); load_res_tile(pp + 2); }
            if (HL && g == 0 && pp == 3) { load_raw(rawA[0], 0); if (DP) load_raw(rawB[4 % NRB], 4); }
            dma_buf(g, wrsrc, wvoff, so_next, dst_next);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int t = 2 * (g & 1) + (i >> 1);
                a3[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aw[g & 1][i]), __builtin_bit_cast(bf16x8, yfr[2 * kc + (i & 1)]), a3[t], 0, 0, 0);
            }
            // half (g & 1) of the conversion of tile kc + 1 (needed by the MFMAs of K step kc + 1) behind this group's MFMAs; its FiLM
            // rows were read during the previous group, the next half's are requested first
            if (kc + 1 < 16) {
                const FilmRows fc = fnext;
                const int nh = 2 * (kc + 1) + (g & 1) + 1;                       // next half, as tile * 2 + c
                if (nh < 32) fnext = film_rows(nh >> 1, nh & 1);
                convert_half(kc + 1, g & 1, fc);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (PROBE) pst[3] = __builtin_readcyclecounter() - pc0;

    // ---- pass B: output tiles 4..15, tile-outer (one chunk = one tile); the epilogue (bias (+ CFG-null constant), fp32 + bf16
    //      stores) of pass-A tile pp rides in phase pp, that of pass-B tile t - 1 in the phase of tile t ------------------------------
    const float* sbias = (p.row_const != nullptr && row < p.n_const_rows) ? sb3c : sb3;
    const unsigned cf_voff = (unsigned)(fbase * sizeof(float));                       // byte offset of (tile 0, quad 0) in Cf
    const unsigned ct_voff = (unsigned)((size_t)tb * 32 * 1024 + lane_off);          // byte offset of fragment 0 in Ct
    // quad qi (fp32 piece) of tile t: add bias, store; keeps the 4 values for the bf16 tile
    auto finish_quad = [&](int t, const f32x16& a, const f32x16* r, int qi, float* v4) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(sbias + t * 32 + 16 * (qi >> 1) + 8 * h + 4 * (qi & 1));
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[e] = a[4 * qi + e] + (r ? (*r)[4 * qi + e] + b4[e] : b4[e]); v4[e] = o[e]; }
        const unsigned vo = cf_voff + (unsigned)t * 4096u;
        switch (qi) {
            case 0: asm_store16<0>(p.Cf, vo, o); break;
            case 1: asm_store16<1024>(p.Cf, vo, o); break;
            case 2: asm_store16<2048>(p.Cf, vo, o); break;
            default: asm_store16<3072>(p.Cf, vo, o); break;
        }
    };
    auto store_bf16 = [&](int t, int c, const float* v8) {
        const u32x4 o = pack8(v8);
        const unsigned vo = ct_voff + (unsigned)t * 2048u;
        if (c == 0) asm_store16<0>(p.Ct, vo, o); else asm_store16<1024>(p.Ct, vo, o);
    };
    // epilogue piece g (0..3) of tile t: quad g; the bf16 fragment c = g >> 1 once both its quads are done
    auto finish_piece = [&](int t, const f32x16& a, const f32x16* r, int g, float* v8) {
        finish_quad(t, a, r, g, v8 + 4 * (g & 1));
        if (g & 1) store_bf16(t, g >> 1, v8);
    };
    // HL: quad g: + bias; once a fragment's two quads are there: + hi + lo of the residual, split, two plane stores
    auto finish_piece_hl = [&](int t, const f32x16& a, const RawRes& r, int g, float* v8) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(sbias + t * 32 + 16 * (g >> 1) + 8 * h + 4 * (g & 1));
#pragma unroll
        for (int e = 0; e < 4; ++e) v8[4 * (g & 1) + e] = a[4 * g + e] + b4[e];
        if (g & 1) {
            const int c = g >> 1;
            if (t >= KH0) hl_accumulate(v8, hfr[2 * t + c], r.lo[c]); else hl_accumulate(v8, r.hi[c], r.lo[c]);
            u32x4 oh, ol;
            hl_split(v8, oh, ol);
            const unsigned vo = ct_voff + (unsigned)t * 2048u;
            if (c == 0) { asm_store16<0>(p.Ct, vo, oh); asm_store16<0>(p.Clo, vo, ol); }
            else { asm_store16<1024>(p.Ct, vo, oh); asm_store16<1024>(p.Clo, vo, ol); }
        }
    };
#pragma unroll
    for (int tt = 0; tt < 12; ++tt) {
        const int q = 68 + tt, t = 4 + tt;
        const unsigned long long ptb0 = PROBE ? __builtin_readcyclecounter() : 0;
        switch (tt) {
            case 0: F3_PHASE_TOP(F3_NY(68)); break;
            case 1: F3_PHASE_TOP(F3_NY(69)); break;
            case 2: F3_PHASE_TOP(F3_NY(70)); break;
            case 3: F3_PHASE_TOP(F3_NY(71)); break;
            case 4: F3_PHASE_TOP(F3_NY(72)); break;
            case 10: F3_PHASE_TOP(F3_NY(78)); break;
            case 11: F3_PHASE_TOP(F3_NY(79)); break;
            default: F3_PHASE_TOP(F3_NY(73)); break;       // = F3_NY(73 .. 77): 24 for both residual forms, 20 with the kept hi plane
        }
        if (PROBE) ptopB += __builtin_readcyclecounter() - ptb0;
        static_assert(f3_ny(HL, PB, 74) == f3_ny(HL, PB, 77) && f3_ny(HL, PB, 73) == f3_ny(HL, PB, 74), "steady-state count");
        const int so_next = dma_soff(q + 3);
        char* dst_next = dma_dst(q + 3);
        const char* cur = lds_lane + (q & 3) * F3_CH;
        u32x4 aw[2][4];
        float va[8], vb[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) aw[0][i] = *reinterpret_cast<const u32x4*>(cur + i * 1024);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (g + 1 < 8) {
#pragma unroll
                for (int i = 0; i < 4; ++i) aw[(g + 1) & 1][i] = *reinterpret_cast<const u32x4*>(cur + ((g + 1) * 4 + i) * 1024);
            }
            if (HL) {
                if (g == 0) {                                                                                // next phase's epilogues (DP: the one after)
                    if (!DP) load_raw(rawB[t % NRB], t); else if (t + 1 < 16) load_raw(rawB[(t + 1) % NRB], t + 1);
                    if (tt < 3) load_raw(rawA[(tt + 1) & 1], tt + 1);
                }
                if (PC == 2) {
                    // PC 2 (round 5): the epilogue of pass-B tile t - 1 — and with it its four stores — in the FIRST half of the phase.
                    // vmcnt counts stores as well, so every counted wait (the phase top's, and the ones hipcc places in front of the
                    // residual fragments' first use — it does not see the asm stores, so its counts are exact for LOADS and over-wait by
                    // the stores still in flight) also waits for every older store: issued in groups 5 and 7 they were a few hundred
                    // cycles old at the next phase top and the top stalled for their acknowledgement; issued in groups 1 and 3 they are
                    // most of a phase old at the next counted wait.
                    if (tt > 0 && g < 4) finish_piece_hl(t - 1, a3[t > 0 ? t - 1 : 0], rawB[(t - 1) % NRB], g, vb);
                    if (tt < 4 && g >= 4) finish_piece_hl(tt, a3[tt], rawA[tt & 1], g - 4, va);
                } else {
                    if (tt < 4 && g < 4) finish_piece_hl(tt, a3[tt], rawA[tt & 1], g, va);
                    if (tt > 0 && g >= 4) finish_piece_hl(t - 1, a3[t > 0 ? t - 1 : 0], rawB[(t - 1) % NRB], g - 4, vb);
                }
            } else {
                if (g == 0 && tt < 10) load_res_tile(t + 2);                               // (just in time: hipcc sees no store, so its wait for these loads stays counted)
                if (g == 0 && tt < 2) load_res_tile(2 + tt);
                if (tt < 4 && g < 4) finish_piece(tt, a3[tt], &ra[tt], g, va);             // pass-A tile tt (+ its residual)
                if (tt > 0 && g >= 4) finish_piece(t - 1, a3[t > 0 ? t - 1 : 0], nullptr, g - 4, vb);   // pass-B tile t - 1
            }
            dma_buf(g, wrsrc, wvoff, so_next, dst_next);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                a3[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aw[g & 1][i]), __builtin_bit_cast(bf16x8, yfr[g * 4 + i]), a3[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (PROBE) pst[4] = __builtin_readcyclecounter() - pc0;
    {
        float v8[8];
#pragma unroll
        for (int g = 0; g < 4; ++g) { if (HL) finish_piece_hl(15, a3[15], rawB[15 % NRB], g, v8); else finish_piece(15, a3[15], nullptr, g, v8); }
    }
    trace_mark(p.trace, 2);
    if (PROBE && p.clk && threadIdx.x == 0) {
        unsigned long long* r = p.clk + (size_t)blockIdx.x * 8;
        r[0] = pst[0]; r[1] = pst[1]; r[2] = pst[2]; r[3] = pst[3]; r[4] = pst[4];
        r[5] = __builtin_readcyclecounter() - pc0; r[6] = wall_clock64() - pw0; r[7] = ptopB;
    }
#undef F3_PHASE_TOP
#undef F3_NY
}

// what launch_tl3_ffn accepts beyond tl2_ffn_supported: at most F3_MAXCLIP clips per 128-token block, and 32-bit byte offsets inside
// the largest tensor a lane addresses — the fp32 stream (4 B per value), or one bf16 plane (2 B) when the residual travels as planes
bool tl3_ffn_supported(int M, int frames, int bmod, bool planes) {
    if (!tl2_ffn_supported(M, frames, bmod) || frames <= 0 || bmod <= 0) return false;
    if (std::min((TL_TOK - 1) / frames + 2, bmod) > F3_MAXCLIP) return false;
    return (size_t)round_up(M, TL_TOK) * 512 * (planes ? 2 : sizeof(float)) < ((size_t)1 << 32);
}

int launch_tl3_ffn(const Tl2FfnArgs& a, hipStream_t s) {
    DSH_REQUIRE(a.M > 0 && (a.X || a.Y) && a.Wffn && a.b1 && a.b2 && a.b3 && a.film && a.Ct && ((a.R && a.Cf) || (a.Rhi && a.Rlo && a.Clo)), "tl3_ffn: null operand");
    DSH_REQUIRE(a.frames > 0 && a.bmod > 0 && a.film_ld % 4 == 0 && a.film_off % 4 == 0, "tl3_ffn: folded FiLM table");
    DSH_REQUIRE(std::min((TL_TOK - 1) / a.frames + 2, a.bmod) <= F3_MAXCLIP, "tl3_ffn: too many clips per 128-token block");
    DSH_REQUIRE((size_t)round_up(a.M, TL_TOK) * 512 * (a.Rhi ? 2 : sizeof(float)) < ((size_t)1 << 32), "tl3_ffn: output offsets are 32-bit");
    // phase-C structure: 1 (default) = pipelined across the phase boundary (round 5), 0 = the round-4 loop (DSH_FFN_PC=0)
    // DSH_FFN_PC: 0 = the round-4 kernel; 1 (default) = phase C pipelined across the phase boundary; 2 = 1 + the epilogues of pass B in the
    // first half of their phase — measured and rejected (profiles/r05_e_ffn_block_timeline.txt: pass B 35.3 k -> 38.6 k cycles; the phase
    // tops it was meant to relieve take 1.7 k of those cycles, the time goes into waiting for the residual fragments, which the early
    // epilogue needs half a phase sooner).  (Read per launch: the op-level tests flip it inside one process.)
    const char* pc_e = getenv("DSH_FFN_PC");
    const int pc = pc_e ? std::min(2, std::max(0, atoi(pc_e))) : 1;
    // DSH_FFN_PB (with DSH_FFN_PC=1, hi / lo planes; default 1): bit 0 = the hi plane of the residual kept in registers — only where the
    // kernel's input IS that plane, as in the denoiser's layers; bit 1 = the pass-B residual requested one phase earlier
    const char* pb_e = getenv("DSH_FFN_PB");
    int pb = (pc == 1 && !a.Y && a.Rhi) ? (pb_e ? (atoi(pb_e) & 3) : 1) : 0;
    if (reinterpret_cast<const void*>(a.X) != reinterpret_cast<const void*>(a.Rhi)) pb &= ~1;
    static const bool attr = [] {
        auto set = [](const void* f) { return hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, F3_LDS) == hipSuccess; };
        bool ok = true;
#define F3_SET(PCV) ok &= set(reinterpret_cast<const void*>(tl3_ffn_kernel<false, false, PCV>)) && set(reinterpret_cast<const void*>(tl3_ffn_kernel<true, false, PCV>)) && \
                          set(reinterpret_cast<const void*>(tl3_ffn_kernel<false, true, PCV>)) && set(reinterpret_cast<const void*>(tl3_ffn_kernel<true, true, PCV>))
        F3_SET(0); F3_SET(1); F3_SET(2);
#undef F3_SET
        return ok;
    }();
    DSH_REQUIRE(attr, "tl3_ffn: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
    Tl2FfnArgs b = a;
    tl_stagger_config(0, &b.stag_groups, &b.stag_sleep);
    const dim3 grid(ceil_div(a.M, TL_TOK)), block(256);
#define F3_LAUNCH(PB, HLV) do { if (pc == 2) hipLaunchKernelGGL((tl3_ffn_kernel<PB, HLV, 2>), grid, block, F3_LDS, s, b); \
                                else if (pc == 1) hipLaunchKernelGGL((tl3_ffn_kernel<PB, HLV, 1>), grid, block, F3_LDS, s, b); \
                                else hipLaunchKernelGGL((tl3_ffn_kernel<PB, HLV, 0>), grid, block, F3_LDS, s, b); } while (0)
    if (a.Y) {
        DSH_REQUIRE(a.Rhi && a.Rlo && a.bs1 && a.Ct == a.Rhi && a.Clo == a.Rlo && a.film_off1 % 4 == 0, "tl3_ffn: the fused attention-branch stage updates the hi / lo planes in place");
        static const bool sattr = [] {
            return hipFuncSetAttribute(reinterpret_cast<const void*>(tl3_ffn_kernel<false, true, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, F3_LDS) == hipSuccess &&
                   hipFuncSetAttribute(reinterpret_cast<const void*>(tl3_ffn_kernel<true, true, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, F3_LDS) == hipSuccess;
        }();
        DSH_REQUIRE(sattr, "tl3_ffn: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
        if (a.clk) hipLaunchKernelGGL((tl3_ffn_kernel<true, true, 1, true>), grid, block, F3_LDS, s, b);
        else hipLaunchKernelGGL((tl3_ffn_kernel<false, true, 1, true>), grid, block, F3_LDS, s, b);
    } else if (pb) {
        static const bool kattr = [] {
            auto set = [](const void* f) { return hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, F3_LDS) == hipSuccess; };
            bool ok = true;
#define F3_SETB(PBV) ok &= set(reinterpret_cast<const void*>(tl3_ffn_kernel<false, true, 1, false, PBV>)) && set(reinterpret_cast<const void*>(tl3_ffn_kernel<true, true, 1, false, PBV>))
            F3_SETB(1); F3_SETB(2); F3_SETB(3);
#undef F3_SETB
            return ok;
        }();
        DSH_REQUIRE(kattr, "tl3_ffn: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
#define F3_LAUNCHB(PBV) do { if (a.clk) hipLaunchKernelGGL((tl3_ffn_kernel<true, true, 1, false, PBV>), grid, block, F3_LDS, s, b); \
                             else hipLaunchKernelGGL((tl3_ffn_kernel<false, true, 1, false, PBV>), grid, block, F3_LDS, s, b); } while (0)
        if (pb == 1) F3_LAUNCHB(1); else if (pb == 2) F3_LAUNCHB(2); else F3_LAUNCHB(3);
#undef F3_LAUNCHB
    } else if (a.Rhi) {
        if (a.clk) F3_LAUNCH(true, true); else F3_LAUNCH(false, true);
    } else {
        if (a.clk) F3_LAUNCH(true, false); else F3_LAUNCH(false, false);
    }
#undef F3_LAUNCH
    DSH_HIP_CHECK(hipGetLastError());
    return 0;
}

void tl_pack_sty_tiles(const uint16_t* wsp, uint16_t* st16) {
    constexpr int D = 512;
    constexpr size_t CH = 16384;
    for (int t = 0; t < 16; ++t) {
        uint16_t* c = st16 + (size_t)t * CH;
        for (int n = 0; n < 32; ++n)
            for (int k = 0; k < D; ++k) c[tl2_frag_index(D, 0, n, k)] = wsp[(size_t)(32 * t + n) * D + k];
    }
}

// Weight stream of the fused FFN kernels from the pi-permuted [N, K] weights (rows already permuted, bf16 bits): version 2 = tl2
// (W3 tile-outer: chunk 64 + t = tile t), version 3 = tl3 (W3 K-outer in two passes, see the header of this file).  `st` receives
// 80 * 16384 elements.
void tl_pack_ffn_stream(int version, const uint16_t* w1p, const uint16_t* w2p, const uint16_t* w3p, uint16_t* st) {
    constexpr int D = 512, F = 1024;
    constexpr size_t CH = 16384;                       // bf16 elements per 32 KB chunk
    for (int j = 0; j < 32; ++j) {
        uint16_t* c1 = st + (size_t)(j ? 2 * j - 1 : 0) * CH;                  // W1 tile j in fragment order
        for (int n = 0; n < 32; ++n)
            for (int k = 0; k < D; ++k) c1[tl2_frag_index(D, 0, n, k)] = w1p[(size_t)(32 * j + n) * D + k];
        uint16_t* c2 = st + (size_t)(j < 31 ? 2 * j + 2 : 63) * CH;            // K chunk j of W2: fragments (output tile ot, k step ks)
        for (int ot = 0; ot < 16; ++ot)
            for (int ks = 0; ks < 2; ++ks)
                for (int ln = 0; ln < 64; ++ln)
                    for (int jj = 0; jj < 8; ++jj)
                        c2[((size_t)(2 * ot + ks) * 64 + ln) * 8 + jj] = w2p[(size_t)(32 * ot + (ln & 31)) * F + 32 * j + 16 * ks + 8 * (ln >> 5) + jj];
    }
    if (version == 2) {
        for (int t = 0; t < 16; ++t) {
            uint16_t* c3 = st + (size_t)(64 + t) * CH;
            for (int n = 0; n < 32; ++n)
                for (int k = 0; k < D; ++k) c3[tl2_frag_index(D, 0, n, k)] = w3p[(size_t)(32 * t + n) * D + k];
        }
        return;
    }
    for (int pp = 0; pp < 4; ++pp) {                     // pass A: K-outer on output tiles 0..3
        uint16_t* c3 = st + (size_t)(64 + pp) * CH;
        for (int kcl = 0; kcl < 4; ++kcl)
            for (int otl = 0; otl < 4; ++otl)
                for (int ks = 0; ks < 2; ++ks)
                    for (int ln = 0; ln < 64; ++ln)
                        for (int jj = 0; jj < 8; ++jj)
                            c3[((size_t)((kcl * 4 + otl) * 2 + ks) * 64 + ln) * 8 + jj] =
                                w3p[(size_t)(32 * otl + (ln & 31)) * D + 32 * (4 * pp + kcl) + 16 * ks + 8 * (ln >> 5) + jj];
    }
    for (int t = 4; t < 16; ++t) {                       // pass B: tile-outer, fragment order
        uint16_t* c3 = st + (size_t)(64 + t) * CH;
        for (int n = 0; n < 32; ++n)
            for (int k = 0; k < D; ++k) c3[tl2_frag_index(D, 0, n, k)] = w3p[(size_t)(32 * t + n) * D + k];
    }
}

}  // namespace dsh
