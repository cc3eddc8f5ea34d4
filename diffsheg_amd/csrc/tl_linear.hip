// "Token-per-lane" (TL) fused Linear for the K = 512 / 1024 layers of the DiffSHEG denoiser (bf16 path).
//
//   out[m, :] = epilogue( prologue(X[m, 0:K]) . W^T )        W: torch Linear weight [N, K]
//
// Why a second GEMM structure.  Profiling the 128x128 LDS-tiled kernel on the M = 167 200-token
// shapes showed waves parked on memory 60 % of the time: every K step of every tile exposes HBM
// latency on the activation panel, and the unfused pipeline round-trips LayerNorm / FiLM outputs
// through HBM.  Here the activation operand is *stationary in registers*:
//
//   * a wave owns 32 tokens; lane (m = lane & 31, h = lane >> 5) holds k = 16 s + 8 h + j (j < 8) of token m
//     as MFMA B-operand fragment s (K/16 fragments, 128 or 256 VGPRs of packed bf16).  All row loads of a
//     block are in flight at once (one HBM latency per block instead of one per K step);
//   * the row sits inside two lanes, so LayerNorm statistics, the affine, the FiLM (1+scale)/shift of
//     StylizationBlock and SiLU are a register prologue (models/transformer.py:86-97, :119);
//   * W streams through a 3-stage LDS ring as the MFMA A operand (v_mfma_f32_32x32x16_bf16,
//     D[n][m] = sum_k W[n][k] X[m][k]).
//
// Activation layouts (ablation r01: with row-major rows, the 8/16-byte-per-lane row accesses of this structure
// cost more than the MFMAs - qkv 505 -> 325 us without its stores).  Every tensor that flows between TL
// kernels is therefore stored *tiled*, so that each wave-wide load/store instruction moves one contiguous KB:
//
//   bf16 [M, Wd]  tile (tb, kt) = 32 tokens x 16 features, row-major inside (32 B per token):
//                 elem (t, n) at ((t >> 5) * (Wd >> 4) + (n >> 4)) * 512 + (t & 31) * 16 + (n & 15)
//                 -> fragment s of a token block IS tile s: one global_load_dwordx4 per fragment, 1 KB contiguous
//   fp32 [M, Wd]  (residual stream h) per (tb, nt = n >> 5): 4 lane-native 1 KB pieces qi:
//                 float index (((tb * (Wd >> 5) + nt) * 4 + qi) * 64 + lane) * 4 + e,
//                 n = 32 nt + 16 (qi >> 1) + 8 h + 4 (qi & 1) + e,  lane = (t & 31) + 32 h
//
// Output feature order.  The 32x32 accumulator gives lane (m, h) tile rows rho = 8 q + 4 h + e.  The weight
// rows of every 32-row tile are stored permuted, W'[32 nt + rho] = W[32 nt + pi(rho)] with
// pi(8q + 4h + e) = 16 (q >> 1) + 8 h + 4 (q & 1) + e, so a lane ends up with 2 x 8 consecutive features of its
// token: exactly the two 16-byte pieces of the bf16 tiles (2 nt) and (2 nt + 1).  K order is natural.
#include <algorithm>
#include <cstdlib>

#include "dsh_common.h"
#include "dsh_kernels.h"
#include "tl_common.h"

namespace dsh {

// KD = 512: 128 fragment VGPRs, 2 blocks / CU.  KD = 1024: 256 fragment VGPRs, one wave per SIMD (512-register budget),
// used for the K = 1024 Linears (ffn.linear2, feat_proj.1 on the padded concat, feat_proj.3).
// ABL (bench only): timing ablations, results are garbage.  1 = no main-loop barriers, 2 = no output stores,
// 8 = no W global loads / LDS writes in the loop, 16 = no MFMA.
// OUT: 1 = fp32 tiled, 2 = bf16 tiled, 3 = both, 4 = fp32 row-major (ldcf; last Linear of an encoder)
// HL (round 4): the residual stream as two bf16 planes (tl_common.h) — residual = hi plane (p.R reinterpreted) + lo plane (p.Rlo),
// result -> p.Ct (hi) + p.Clo (lo); instantiated for the two residual-carrying launches of a layer (OUT = 3 shape).
template <int KD, int PRO, bool HAS_R, int OUT, int ACT, int ABL = 0, bool HL = false>
__global__ __launch_bounds__(256, (KD == 512 ? 2 : 1)) void tl_linear_kernel(TlArgs p) {
    static_assert(!HL || (HAS_R && OUT == 3 && ACT == ACT_NONE && ABL == 0), "hi / lo planes: the residual-carrying instantiations only");
    constexpr int TL_K = KD, NFRAG = KD / 16, NST = KD / TL_STAGE_K;   // fragments per lane, LDS stages per 32-feature tile
    constexpr bool HAS_C = (PRO == 2 && HAS_R && ACT == ACT_NONE);   // only the StylizationBlock instantiation takes row_const
    constexpr int LDS_W = tl_lds_bytes(KD);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // clock probe (bench only): block 0 records shader-clock and 100 MHz wall-clock ticks at entry / exit
    unsigned long long clk0 = 0, rt0 = 0;
    if (p.clk) { clk0 = __builtin_readcyclecounter(); rt0 = wall_clock64(); }
    const int ml = lane & 31, h = lane >> 5;
    // Rows are NOT bounds-checked: every row-indexed buffer (X, R, Cf, Ct) must be allocated for
    // ceil(M / 128) * 128 rows.  Guarded (conditional) memory ops would make the compiler's in-order vmcnt
    // accounting conservative and drain the W prefetch queue at every epilogue.
    const int bx = tl_block_index(p.rev);
    const int tb = bx * (TL_TOK / 32) + wave;                  // 32-token block owned by this wave
    const int row = tb * 32 + ml;
    const int lane_off = ml * 32 + h * 16;                      // byte offset of this lane's 16 B inside a bf16 tile

    // ---- W staging bookkeeping: a stage is 32 rows x 512 B = 1024 16-byte chunks, 4 per thread ----
    const char* Wb = reinterpret_cast<const char*>(p.W);
    int w_goff[4], w_loff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + 256 * i;
        const int r = c >> 5, col = c & 31;
        w_goff[i] = r * (TL_K * 2) + col * 16;     // + nt * 32 rows * 1024 B + half * 512 B
        w_loff[i] = r * TL_ROW + col * 16;
    }
    // blockIdx.y selects a chunk of 32-feature tiles (small-M launches split N over the grid so that more than a
    // couple of CUs stream the weight); g counts LDS stages globally, this block runs stages [g0, nst)
    const int nt0 = blockIdx.y * p.tiles_per_block;
    const int nt1 = (nt0 + p.tiles_per_block) < (p.N / 32) ? (nt0 + p.tiles_per_block) : (p.N / 32);
    const int g0 = nt0 * NST;
    const int nst = nt1 * NST;                     // one past this block's last stage
    auto stage_src = [&](int g, int i) -> const u32x4* {
        const int gt = (p.dbg & 8) ? 0 : g / NST;   // ablation bit 8: every tile re-reads tile 0 (W stays L1/L2-hot)
        return reinterpret_cast<const u32x4*>(Wb + (size_t)gt * (32 * TL_K * 2) + (g % NST) * (TL_STAGE_K * 2) + w_goff[i]);
    };
    // ---- prologue parameters (requested first: they are the oldest entries of the in-order vmcnt queue).  They are
    //      staged in LDS, overlaid on the second W buffer (first written by the main loop, after the pre-loop barrier):
    //      PRO 1/3: LayerNorm gamma | beta;  PRO 2: the folded StylizationBlock coefficients A | B of this block's clips
    constexpr int NPRM = PRO == 2 ? TL_MAXCLIP : (PRO == 0 ? 1 : KD / 512);
    f32x4 prm[NPRM];
    int clip0 = 0;
    if (PRO == 2) {   // rows >= half_row0 are the second (conditional) CFG half, stored behind a block-aligned gap
        const int rb = bx * TL_TOK, rrb = rb >= p.half_row0 ? rb - p.half_row0 : rb;
        clip0 = rrb / p.frames;
        const int nclip = (rrb + TL_TOK - 1) / p.frames - clip0 + 1;
#pragma unroll
        for (int c = 0; c < TL_MAXCLIP; ++c) {
            const int cc = c < nclip ? c : nclip - 1;
            prm[c] = *reinterpret_cast<const f32x4*>(p.film + (size_t)((clip0 + cc) % p.bmod) * p.film_ld + p.film_off + tid * 4);
        }
    } else if (PRO != 0) {
#pragma unroll
        for (int c = 0; c < NPRM; ++c) {   // thread t takes floats [1024 c + 4 t, +4) of gamma | beta
            const int f = 1024 * c + 4 * tid;
            prm[c] = *reinterpret_cast<const f32x4*>(f < TL_K ? p.gamma + f : p.beta + (f - TL_K));
        }
    }
    // W stages in flight in registers (set = stage % NPF).  KD = 512 (two waves per SIMD share the matrix pipe): 2 stages
    // ~ 2 x 1 us.  KD = 1024 runs one wave per SIMD at full MFMA rate, where 2 stages (32 MFMAs ~ 0.5 us) is less than
    // an L2 round trip under load and every stage stalled on vmcnt: a whole tile (4 stages) is kept in flight instead.
    constexpr int NPF = KD == 1024 ? 4 : 2;
    static_assert(NST % NPF == 0, "register set of a stage must not depend on the tile");
    u32x4 wreg[NPF][4];
    u32x4 wpre[NST][4]; // the first tile, written to LDS once the row loads have been issued
#pragma unroll
    for (int hs = 0; hs < NST; ++hs)
#pragma unroll
        for (int i = 0; i < 4; ++i) wpre[hs][i] = *stage_src(g0 + hs < nst ? g0 + hs : nst - 1, i);

    // ---- activation rows -> B fragments: frag[s] = X[row][(KD/2) h + 8s .. +7] -------------------
    u32x4 frag[NFRAG];
    if (PRO == 3) {
        // un-materialised concat [latent 512 | audio_proj 256 | hubert 128 | expr_x0 128 (zero padded; absent for the
        // expression encoder)]: fragments 0..31 / 32..47 / 48..55 / 56..63 are the tiles of four tiled tensors
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
    // first stage(s) -> LDS while the row loads are in flight
    // (prefetches are unconditional with a clamped stage index: conditional loads would force the
    //  compiler's vmcnt bookkeeping to the conservative "wait for everything")
#pragma unroll
    for (int hs = 0; hs < NST; ++hs)
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(smem + ((nt0 & 1) * NST + hs) * TL_STAGE + w_loff[i]) = wpre[hs][i];
#pragma unroll
    for (int f = 0; f < NPF; ++f)
#pragma unroll
        for (int i = 0; i < 4; ++i) wreg[f][i] = *stage_src(g0 + NST + f < nst ? g0 + NST + f : nst - 1, i);

    float* sprm = reinterpret_cast<float*>(smem + (((nt0 + 1) & 1) * NST) * TL_STAGE);
    if (PRO >= 1) {
#pragma unroll
        for (int c = 0; c < NPRM; ++c) *reinterpret_cast<f32x4*>(sprm + 1024 * c + 4 * tid) = prm[c];
        __syncthreads();
        // LayerNorm statistics over the row (two lanes per token): raw moments by packed dot products (tl_common.h)
        float sum, sq;
        row_moments_bf16<NFRAG>(frag, sum, sq);
        const float kn = PRO == 3 ? (float)p.kreal : (float)TL_K;          // LayerNorm width (concat: un-padded)
        const float mean = sum / kn;
        sq = fmaxf(sq - sum * mean, 0.f);                                  // sum (x - mean)^2
        const float rstd = 1.0f / sqrtf(sq / kn + 1e-5f);
        const float nmr = -mean * rstd;
        // y = ((x - mean) rstd) ca + cb with (ca, cb) = (gamma, beta), or the per-clip folded FiLM pair
        // (A = gamma (1 + scale), B = beta (1 + scale) + shift; film_fold in rowops.hip) followed by SiLU
        const float* ca = sprm + 8 * h;
        const float* cb = sprm + TL_K + 8 * h;
        if (PRO == 2) {
            const int rr = row >= p.half_row0 ? row - p.half_row0 : row;
            int ci = rr / p.frames - clip0;                    // rows past the last clip (block padding) may exceed the staged rows
            ci = ci < TL_MAXCLIP ? ci : TL_MAXCLIP - 1;
            ca = sprm + ci * 1024 + 8 * h;
            cb = ca + 512;
        }
        f32x4 pa[2][2], pb[2][2];
#pragma unroll
        for (int q = 0; q < 2; ++q) { pa[0][q] = *reinterpret_cast<const f32x4*>(ca + 4 * q); pb[0][q] = *reinterpret_cast<const f32x4*>(cb + 4 * q); }
#pragma unroll
        for (int s = 0; s < NFRAG; ++s) {
            if (s + 1 < NFRAG) {   // next step's coefficients are in flight while this step computes
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
                    v[4 * q + e] = PRO == 2 ? y * __builtin_amdgcn_rcpf(1.0f + __expf(-y)) : y;
                }
#pragma unroll
            for (int j = 0; j < 4; ++j) frag[s][j] = pack_bf16(v[2 * j], v[2 * j + 1]);
            // one scheduling region per step: keeps hipcc from hoisting every step's coefficient reads (frags already fill the file)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // All row fragments must have landed before the main loop: otherwise the compiler's in-order vmcnt
    // bookkeeping makes every later wait (W prefetch) drain the whole queue on each trip.
#pragma unroll
    for (int s = 0; s < NFRAG; ++s) asm volatile("" ::"v"(frag[s]));
    // bias -> LDS (read back with ds_read: keeps the epilogue off the vmcnt queue)
    // (likewise the per-feature constant added to the first n_const_rows rows: CFG-null feat_proj term)
    float* sbias = reinterpret_cast<float*>(smem + LDS_W);
    float* sconst = sbias + p.N;
    for (int i = tid; i < p.N; i += 256) {
        sbias[i] = p.bias ? p.bias[i] : 0.f;
        sconst[i] = p.row_const ? p.row_const[i] : 0.f;
    }
    __syncthreads();

    // ---- main loop: one 32-feature tile of W per iteration, two LDS stages each -----------------
    char* Ctb = reinterpret_cast<char*>(p.Ct);
    const int NT = p.N / 32;
    const int a_off = ml * TL_ROW + h * 16;
    const float const_on = (p.row_const != nullptr && row < p.n_const_rows) ? 1.0f : 0.0f;
    int g = g0;
    unsigned long long bar_cyc = 0;
    for (int nt = nt0; nt < nt1; ++nt) {
        // the accumulator starts from the bias (+ the CFG-null row constant): the epilogue then touches no LDS
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
        // fragment reads are software-pipelined in groups of 4 (one group = 4 MFMAs = 128 cycles, about one
        // ds_read_b128 latency) and run ahead across the stage boundaries inside the tile
        u32x4 aw[2][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) aw[0][i] = *reinterpret_cast<const u32x4*>(smem + ((nt & 1) * NST) * TL_STAGE + a_off + i * 32);
        // (own scheduling region: these LDS reads must not be counted against the fragment-read slots pinned below)
        __builtin_amdgcn_sched_barrier(0);
        // residual for this tile is requested before the W prefetch of the tile, so that waiting for it
        // later does not drain the younger W loads (vmcnt completes in order)
        f32x4 rres[4];
        u32x4 rhi[2], rlo[2];                                                  // HL: the tile's two fragments of either plane
        const size_t fidx = (((size_t)tb * NT + nt) * 4 * 64 + lane) * 4;      // lane-native fp32 piece qi at + qi * 256
        const size_t pidx = ((size_t)tb * (2 * NT) + 2 * nt) * 1024 + lane_off;    // bf16 fragment c of this tile at + c * 1024
        if (HL) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                rhi[c] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(p.R) + pidx + c * 1024);
                rlo[c] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(p.Rlo) + pidx + c * 1024);
            }
        } else if (HAS_R) {
#pragma unroll
            for (int q = 0; q < 4; ++q) rres[q] = *reinterpret_cast<const f32x4*>(p.R + fidx + q * 256);
        }
        {
            // tile nt is read from buffer nt & 1 while tile nt + 1 is written into the other one: one barrier per tile
            const char* rbuf = smem + ((nt & 1) * NST) * TL_STAGE + a_off;
            char* wbuf = smem + (((nt + 1) & 1) * NST) * TL_STAGE;
#pragma unroll
            for (int half = 0; half < NST; ++half, ++g) {
                if (!(ABL & 8)) {
                    char* dst = wbuf + half * TL_STAGE;
#pragma unroll
                    for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(dst + w_loff[i]) = wreg[half % NPF][i];
                    const int gn = g + NST + NPF < nst ? g + NST + NPF : nst - 1;
#pragma unroll
                    for (int i = 0; i < 4; ++i) wreg[half % NPF][i] = *stage_src(gn, i);
                }
                const char* cur = rbuf + half * TL_STAGE;
#pragma unroll
                for (int grp = 0; grp < 4; ++grp) {
                    if (grp < 3) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) aw[(grp + 1) & 1][i] = *reinterpret_cast<const u32x4*>(cur + ((grp + 1) * 4 + i) * 32);
                    } else if (half + 1 < NST) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) aw[0][i] = *reinterpret_cast<const u32x4*>(cur + TL_STAGE + i * 32);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (ABL & 16) { asm volatile("" ::"v"(aw[grp & 1][i])); continue; }
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aw[grp & 1][i]),
                                                                      __builtin_bit_cast(bf16x8, frag[16 * half + grp * 4 + i]), acc, 0, 0, 0);
                    }
                }
                // pin the issue order (hipcc otherwise re-serialises each ds_read right in front of its MFMA):
                // (DSR x4 next group, MFMA x4) per group        masks: 0x100 = DS read, 0x008 = MFMA
#pragma unroll
                for (int grp = 0; grp < 4; ++grp) {
                    if (grp < 3 || half + 1 < NST) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                }
                if (half == NST - 1) {
                    if (ABL & 64) {   // probe: cycles this wave spends parked at the per-tile barrier
                        const unsigned long long b0 = __builtin_readcyclecounter();
                        __syncthreads();
                        bar_cyc += __builtin_readcyclecounter() - b0;
                    } else if (!(ABL & 1)) __syncthreads();
                }
                else __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- epilogue for features [32 nt, 32 nt + 32): accumulator quad qi of lane (m, h) holds
        //      n = 32 nt + 16 (qi >> 1) + 8 h + 4 (qi & 1) + e of token `row` (weight rows are pi-permuted)
        {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                float v8[8];
#pragma unroll
                for (int qq = 0; qq < 2; ++qq) {
                    const int qi = 2 * c + qq;
                    const int col = nt * 32 + 16 * c + 8 * h + 4 * qq;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[4 * qi + e];
                    if (ACT == ACT_GELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = gelu_fast(v[e]);
                    } else if (ACT == ACT_SILU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = v[e] * __builtin_amdgcn_rcpf(1.0f + __expf(-v[e]));
                    }
                    if (HAS_R && !HL) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += rres[qi][e]; }
#pragma unroll
                    for (int e = 0; e < 4; ++e) v8[4 * qq + e] = v[e];
                    if (ABL & 2) { asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3])); continue; }
                    if ((OUT & 5) && !HL) { f32x4 o; o.x = v[0]; o.y = v[1]; o.z = v[2]; o.w = v[3];
                        if (OUT & 4) *reinterpret_cast<f32x4*>(p.Cf + (size_t)row * p.ldcf + col) = o;
                        else *reinterpret_cast<f32x4*>(p.Cf + fidx + qi * 256) = o; }
                }
                if (HL) {
                    hl_accumulate(v8, rhi[c], rlo[c]);
                    u32x4 oh, ol;
                    hl_split(v8, oh, ol);
                    *reinterpret_cast<u32x4*>(Ctb + pidx + c * 1024) = oh;
                    *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(p.Clo) + pidx + c * 1024) = ol;
                } else if ((OUT & 2) && !(ABL & 2)) {
                    u32x4 o;
                    o.x = pack_bf16(v8[0], v8[1]); o.y = pack_bf16(v8[2], v8[3]);
                    o.z = pack_bf16(v8[4], v8[5]); o.w = pack_bf16(v8[6], v8[7]);
                    *reinterpret_cast<u32x4*>(Ctb + ((size_t)tb * (2 * NT) + 2 * nt + c) * 1024 + lane_off) = o;
                }
            }
        }
    }
    if (p.clk && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) {
        p.clk[0] = __builtin_readcyclecounter() - clk0;
        p.clk[1] = wall_clock64() - rt0;
        p.clk[2] = bar_cyc;
    }
}

int launch_tl_linear(const TlArgs& a, int pro, hipStream_t s) {
    g_tl_last_variant = 10;
    DSH_REQUIRE(a.M > 0 && a.N > 0 && a.N % 32 == 0, "tl_linear: N must be a positive multiple of 32");
    DSH_REQUIRE(a.K == 512 || a.K == 1024, "tl_linear: K must be 512 or 1024");
    DSH_REQUIRE(a.ldx == (pro == 3 ? 512 : a.K), "tl_linear: the tiled input must be exactly K features wide");
    DSH_REQUIRE(((uintptr_t)a.X % 16) == 0 && ((uintptr_t)a.W % 16) == 0, "tl_linear: operands must be 16-byte aligned");
    DSH_REQUIRE(!a.Cf || !a.cf_rowmajor || a.ldcf % 4 == 0, "tl_linear: row-major output leading dim");
    DSH_REQUIRE(!(a.cf_rowmajor && (a.R || a.Ct)), "tl_linear: the row-major fp32 output has no residual / bf16 shadow");
    DSH_REQUIRE(pro == 0 || (a.gamma && a.beta), "tl_linear: LayerNorm prologue needs gamma/beta");
    // a 128-token block stages the folded FiLM rows of at most TL_MAXCLIP clips: short windows are fine as long as the
    // launch holds few clips (the B = 1 window chain and its tail windows), long clips at any batch
    DSH_REQUIRE(pro != 2 || (a.film && a.frames > 0 && a.bmod > 0 && a.film_ld % 4 == 0 && a.film_off % 4 == 0),
                "tl_linear: FiLM prologue needs the folded film table");
    DSH_REQUIRE(pro != 2 || std::min((TL_TOK - 1) / a.frames + 2, a.bmod) <= TL_MAXCLIP,
                "tl_linear: FiLM prologue: too many clips per 128-token block (clips shorter than 26 frames need batch <= 6)");
    DSH_REQUIRE(pro != 2 || a.K == 512, "tl_linear: FiLM prologue is instantiated for K = 512");
    // N is split over grid.y only when the token blocks alone cannot fill the chip (window-chain batches)
    // (the finest split whose grid still fits one round of resident blocks: two 128-token blocks per CU at K = 512, one at
    //  K = 1024, whose fragments fill the register file)
    const int mblocks = ceil_div(a.M, TL_TOK), ntiles = a.N / 32, slots = a.K == 512 ? 512 : 256;
    int tpb = ntiles;
    if (mblocks < 256) { tpb = 1; while (tpb < ntiles && mblocks * ceil_div(ntiles, tpb) > slots) ++tpb; }
    TlArgs b = a;
    b.tiles_per_block = tpb;
    const dim3 grid(mblocks, ceil_div(ntiles, tpb)), block(256);
    const int lds = tl_lds_bytes(a.K) + 2 * a.N * 4;
    DSH_REQUIRE(lds <= 160 * 1024, "tl_linear: N too large for the LDS bias table");
    DSH_REQUIRE(pro >= 0 && pro <= 3, "tl_linear: unknown prologue");
    DSH_REQUIRE(!a.row_const || (pro == 2 && a.R && a.act == ACT_NONE), "tl_linear: row_const is only wired into the StylizationBlock instantiation");
    DSH_REQUIRE(pro != 3 || (a.K == 1024 && a.X1 && a.X2 && a.kreal > 896 - 1 && a.kreal <= 1024), "tl_linear: concat prologue arguments");
    // Straight-line epilogues only: every (prologue, residual, outputs, activation) combination used by the
    // denoiser is its own instantiation, so the compiler's vmcnt accounting stays exact (no conservative drains).
    typedef void (*kern_t)(TlArgs);
    struct Variant { int k, pro, has_r, out, act; kern_t fn; };
#define TLV(P, R, O, A) {512, P, R, O, A, tl_linear_kernel<512, P, (R) != 0, O, A>}
#define TLV1K(P, R, O, A) {1024, P, R, O, A, tl_linear_kernel<1024, P, (R) != 0, O, A>}
    static const Variant variants[] = {
        TLV(1, 0, 2, ACT_NONE),   // sa_block: LayerNorm -> q|k|v                       (bf16 out)
        TLV(2, 1, 3, ACT_NONE),   // StylizationBlock: LN+FiLM+SiLU -> Linear -> +h     (fp32 h + bf16 shadow)
        TLV(0, 0, 2, ACT_GELU),   // ffn.linear1 + GELU                                  (bf16 out)
        TLV(0, 0, 2, ACT_NONE), TLV(0, 1, 3, ACT_NONE), TLV(0, 0, 2, ACT_SILU), TLV(1, 0, 1, ACT_NONE),
        TLV(2, 0, 2, ACT_NONE), TLV(0, 1, 1, ACT_NONE), TLV(0, 0, 1, ACT_NONE),
        TLV(0, 0, 4, ACT_NONE),   // encoder `out` head: plain rows -> fp32 row-major
        TLV1K(0, 0, 2, ACT_NONE),  // ffn.linear2                                        (bf16 out)
        TLV1K(0, 0, 2, ACT_SILU),  // feat_proj.1 on the LayerNorm-ed, zero-padded concat + SiLU
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
    const bool hl = a.Rlo != nullptr || a.Clo != nullptr;
    DSH_REQUIRE(!hl || (a.Rlo && a.Clo && a.R && a.Ct && !a.Cf && a.act == ACT_NONE && ((pro == 2 && a.K == 512) || (pro == 0 && a.K == 1024))),
                "tl_linear: hi / lo planes need both residual and output planes, no fp32 output, and one of the two residual-carrying instantiations");
    const int out = hl ? 3 : ((a.Cf ? (a.cf_rowmajor ? 4 : 1) : 0) | (a.Ct ? 2 : 0)), has_r = a.R ? 1 : 0;
    kern_t fn = nullptr;
    for (int i = 0; i < NV; ++i)
        if (variants[i].k == a.K && variants[i].pro == pro && variants[i].has_r == has_r && variants[i].out == out && variants[i].act == a.act) fn = variants[i].fn;
    if (hl) {
        fn = pro == 2 ? static_cast<kern_t>(tl_linear_kernel<512, 2, true, 3, ACT_NONE, 0, true>) : static_cast<kern_t>(tl_linear_kernel<1024, 0, true, 3, ACT_NONE, 0, true>);
        static bool hattr = false;
        if (!hattr) {
            DSH_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(tl_linear_kernel<512, 2, true, 3, ACT_NONE, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            DSH_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(tl_linear_kernel<1024, 0, true, 3, ACT_NONE, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            hattr = true;
        }
    }
    if (a.dbg >> 8) {   // bench-only timing ablations of the two dominant instantiations
        const int abl = a.dbg >> 8;
        struct Abl { int pro, abl; kern_t fn; };
#define TLA(B) {1, B, tl_linear_kernel<512, 1, false, 2, ACT_NONE, B>}, {2, B, tl_linear_kernel<512, 2, true, 3, ACT_NONE, B>}
        static const Abl abls[] = {TLA(1), TLA(2), TLA(8), TLA(16), TLA(3), TLA(9), TLA(11), TLA(27), TLA(64)};
#undef TLA
        fn = nullptr;
        for (const Abl& e : abls) if (e.pro == pro && e.abl == abl) {
            fn = e.fn;
            DSH_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        }
    }
    DSH_REQUIRE(fn != nullptr, "tl_linear: this (prologue, residual, outputs, activation) combination is not instantiated");
    hipLaunchKernelGGL(fn, grid, block, lds, s, b);
    DSH_HIP_CHECK(hipGetLastError());
    return 0;
}

// source row of stored weight row r: W'[32 nt + rho] = W[32 nt + pi(rho)]   (host helper used by finalize())
int tl_weight_src_row(int r) {
    const int rho = r & 31, q = rho >> 3, hh = (rho >> 2) & 1, e = rho & 3;
    return (r & ~31) + 16 * (q >> 1) + 8 * hh + 4 * (q & 1) + e;
}

}  // namespace dsh
