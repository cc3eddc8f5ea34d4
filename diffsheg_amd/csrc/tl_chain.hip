// Chained token-per-lane kernel:  ffn.linear2  ->  StylizationBlock(ffn.proj_out)  ->  + h      (bf16 path)
//
//   y2 = g W2^T + b2                      (models/transformer.py:173, K = 1024 -> 512)
//   h  = h + Linear3( SiLU( LN(y2) (1 + scale) + shift ) )      (:86-97, :181)
//
// Run separately (tl_linear.hip) the pair moves 9 KB per token through HBM and each launch exposes a full activation
// panel load before its first MFMA.  Chained, y2 never leaves the register file (the pi-permuted accumulator layout of
// one Linear *is* the B-operand layout of the next) and g is no longer a prologue:
//
//   phase C  (output stationary): the 16 accumulators of a wave's 32 x 512 output live in registers (256) and g streams
//            through in four 256-wide K chunks, the next chunk being loaded while the current one feeds the matrix pipe;
//   phase D  LayerNorm + folded FiLM + SiLU on the packed y2 fragments in place, then the K = 512 Linear with the fp32
//            residual epilogue of tl_linear (h in -> h out + bf16 shadow, CFG-null constant of the next layer).
//
// Status (round 1): parity is exact, but the launch takes as long as the two separate ones (476 us on the bench shape:
// phase C alone 252 us vs 271 us for tl_linear's ffn.linear2, phase D alone 257 us vs 205 us for the two-wave stylization
// kernel).  With one wave per SIMD nothing hides the in-order vmcnt coupling between the activation / residual loads and
// the W stage loads (phase C drops to 196 us without its in-loop g loads), so the chain is OFF by default (DSH_CHAIN=1
// enables it); it becomes profitable once the K = 1024 side runs two waves per SIMD.
//
// W2 then W3 form one stream of 96 LDS stages (64 + 32) through the same double-buffered group-of-four-stages ring and
// one barrier per 64 MFMAs.  The arithmetic (operand order, rounding points) is that of the two separate kernels, so the
// results are bit-identical to the unchained path (tests/test_gpu_eval.py).
#include <cstdlib>

#include "dsh_common.h"
#include "dsh_kernels.h"
#include "tl_common.h"

namespace dsh {

constexpr int CH_NSLOT = 8;                       // LDS stage slots: two groups of four stages
constexpr int CH_LDS_W = CH_NSLOT * TL_STAGE;     // 135,168 B
constexpr int CH_MAXCLIP = 3;                     // clips a 128-token block may span (frames >= 64)
constexpr int CH_NQ = 64 + 32;                    // W2 stages (16 tiles x 4 chunks, chunk-major) + W3 stages (16 tiles x 2)
constexpr int CH_LDS = CH_LDS_W + CH_MAXCLIP * 1024 * 4 + 3 * 512 * 4;

__global__ __launch_bounds__(256, 1) void tl_chain2_kernel(TlChain2Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ml = lane & 31, h = lane >> 5;
    const int tb = blockIdx.x * (TL_TOK / 32) + wave;          // 32-token block owned by this wave (rows are not bounds-checked)
    const int row = tb * 32 + ml;
    const int lane_off = ml * 32 + h * 16;
    const int a_off = ml * TL_ROW + h * 16;

    const char* W2b = reinterpret_cast<const char*>(p.W2);
    const char* W3b = reinterpret_cast<const char*>(p.W3);
    int goff2[4], goff3[4], loff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + 256 * i, r = c >> 5, col = c & 31;
        goff2[i] = r * 2048 + col * 16;
        goff3[i] = r * 1024 + col * 16;
        loff[i] = r * TL_ROW + col * 16;
    }
    // stage q of the weight stream: q < 64: W2 rows of tile (q & 15), K chunk (q >> 4); else W3 tile ((q-64) >> 1), half ((q-64) & 1)
    auto stage_src = [&](int q, int i) -> const u32x4* {
        q = q < CH_NQ ? q : CH_NQ - 1;
        const int q3 = q - 64;
        const char* a2 = W2b + (size_t)(q & 15) * 65536 + (q >> 4) * 512 + goff2[i];
        const char* a3 = W3b + (size_t)(q3 >> 1) * 32768 + (q3 & 1) * 512 + goff3[i];
        return reinterpret_cast<const u32x4*>(q < 64 ? a2 : a3);
    };

    // ---- folded FiLM rows (A | B) of this block's clips: requested first, staged in LDS ---------------------------
    f32x4 prm[CH_MAXCLIP];
    int clip0;
    {
        const int rb = blockIdx.x * TL_TOK, rrb = rb >= p.half_row0 ? rb - p.half_row0 : rb;
        clip0 = rrb / p.frames;
        const int nclip = (rrb + TL_TOK - 1) / p.frames - clip0 + 1;
#pragma unroll
        for (int c = 0; c < CH_MAXCLIP; ++c) {
            const int cc = c < nclip ? c : nclip - 1;
            prm[c] = *reinterpret_cast<const f32x4*>(p.film + (size_t)((clip0 + cc) % p.bmod) * p.film_ld + p.film_off + tid * 4);
        }
    }
    u32x4 wreg[2][4], wpre[4][4];
#pragma unroll
    for (int hs = 0; hs < 4; ++hs)
#pragma unroll
        for (int i = 0; i < 4; ++i) wpre[hs][i] = *stage_src(hs, i);
    // first K chunk of g
    const char* gb = reinterpret_cast<const char*>(p.G) + (size_t)tb * 64 * 1024 + lane_off;
    u32x4 gA[16], gB[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) gA[s] = *reinterpret_cast<const u32x4*>(gb + s * 1024);
#pragma unroll
    for (int hs = 0; hs < 4; ++hs)
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(smem + hs * TL_STAGE + loff[i]) = wpre[hs][i];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int i = 0; i < 4; ++i) wreg[f][i] = *stage_src(4 + f, i);

    float* sprm = reinterpret_cast<float*>(smem + CH_LDS_W);
    float* sb2 = sprm + CH_MAXCLIP * 1024;
    float* sb3 = sb2 + 512;
    float* sconst = sb3 + 512;
#pragma unroll
    for (int c = 0; c < CH_MAXCLIP; ++c) *reinterpret_cast<f32x4*>(sprm + 1024 * c + 4 * tid) = prm[c];
    for (int i = tid; i < 512; i += 256) {
        sb2[i] = p.b2 ? p.b2[i] : 0.f;
        sb3[i] = p.b3 ? p.b3[i] : 0.f;
        sconst[i] = p.row_const ? p.row_const[i] : 0.f;
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) asm volatile("" ::"v"(gA[s]));      // the first chunk has landed: exact vmcnt accounting below
    __syncthreads();

    // one LDS stage = 16 MFMAs of `acc` against fragments fr[0..15]; also stages W (stage q + 4 -> LDS, q + 6 -> registers)
    // and keeps the fragment reads one group ahead (across the stage boundary when `more`)
    u32x4 aw[2][4];
    auto run_stage = [&](int q, int par, f32x16& acc, const u32x4* fr, bool more) {
        {
            char* dst = smem + ((q + 4) & 7) * TL_STAGE;
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(dst + loff[i]) = wreg[par][i];
#pragma unroll
            for (int i = 0; i < 4; ++i) wreg[par][i] = *stage_src(q + 6, i);
        }
        const char* cur = smem + (q & 7) * TL_STAGE + a_off;
        const char* nxt = smem + ((q + 1) & 7) * TL_STAGE + a_off;
#pragma unroll
        for (int grp = 0; grp < 4; ++grp) {
            if (grp < 3) {
#pragma unroll
                for (int i = 0; i < 4; ++i) aw[(grp + 1) & 1][i] = *reinterpret_cast<const u32x4*>(cur + ((grp + 1) * 4 + i) * 32);
            } else if (more) {
#pragma unroll
                for (int i = 0; i < 4; ++i) aw[0][i] = *reinterpret_cast<const u32x4*>(nxt + i * 32);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aw[grp & 1][i]),
                                                              __builtin_bit_cast(bf16x8, fr[grp * 4 + i]), acc, 0, 0, 0);
        }
#pragma unroll
        for (int grp = 0; grp < 4; ++grp) {
            if (grp < 3 || more) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
    };

    // ---- phase C: y2 = g W2^T + b2, output stationary ------------------------------------------------------------
    f32x16 acc[16];
#pragma unroll
    for (int nt = 0; nt < 16; ++nt)
#pragma unroll
        for (int qi = 0; qi < 4; ++qi) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(sb2 + nt * 32 + 16 * (qi >> 1) + 8 * h + 4 * (qi & 1));
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[nt][4 * qi + e] = b4[e];
        }
    auto chunk = [&](const u32x4* cur, u32x4* nx, int kc) {
        const int kn = kc + 1 < 4 ? kc + 1 : 3;
#pragma unroll
        for (int s = 0; s < 16; ++s) nx[s] = *reinterpret_cast<const u32x4*>(gb + (16 * kn + s) * 1024);
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) aw[0][i] = *reinterpret_cast<const u32x4*>(smem + ((kc * 16 + g4 * 4) & 7) * TL_STAGE + a_off + i * 32);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                run_stage(kc * 16 + g4 * 4 + j, j & 1, acc[g4 * 4 + j], cur, j < 3);
                if (j < 3) __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
        }
    };
    for (int kc = 0; kc < 4; kc += 2) {
        chunk(gA, gB, kc);
        chunk(gB, gA, kc + 1);
    }

    // ---- y2 -> packed bf16 fragments (fragment s = 2 nt + c), LayerNorm + folded FiLM + SiLU in place -----------------
    u32x4 yfr[32];
#pragma unroll
    for (int nt = 0; nt < 16; ++nt)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            u32x4 o;
            o.x = pack_bf16(acc[nt][8 * c + 0], acc[nt][8 * c + 1]); o.y = pack_bf16(acc[nt][8 * c + 2], acc[nt][8 * c + 3]);
            o.z = pack_bf16(acc[nt][8 * c + 4], acc[nt][8 * c + 5]); o.w = pack_bf16(acc[nt][8 * c + 6], acc[nt][8 * c + 7]);
            yfr[2 * nt + c] = o;
        }
    {
        float sum = 0.f;
#pragma unroll
        for (int s = 0; s < 32; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) sum += bf_lo(yfr[s][j]) + bf_hi(yfr[s][j]);
        sum += __shfl_xor(sum, 32, 64);
        const float mean = sum / 512.f;
#pragma unroll
        for (int s = 0; s < 32; ++s) asm volatile("" : "+v"(yfr[s]));
        float sq = 0.f;
#pragma unroll
        for (int s = 0; s < 32; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = bf_lo(yfr[s][j]) - mean, b = bf_hi(yfr[s][j]) - mean;
                sq += a * a + b * b;
            }
        sq += __shfl_xor(sq, 32, 64);
#pragma unroll
        for (int s = 0; s < 32; ++s) asm volatile("" : "+v"(yfr[s]));
        const float rstd = 1.0f / sqrtf(sq / 512.f + 1e-5f);
        const float nmr = -mean * rstd;
        const int rr = row >= p.half_row0 ? row - p.half_row0 : row;
        const float* ca = sprm + (rr / p.frames - clip0) * 1024 + 8 * h;
        const float* cb = ca + 512;
        f32x4 pa[2][2], pb[2][2];
#pragma unroll
        for (int q = 0; q < 2; ++q) { pa[0][q] = *reinterpret_cast<const f32x4*>(ca + 4 * q); pb[0][q] = *reinterpret_cast<const f32x4*>(cb + 4 * q); }
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            if (s + 1 < 32) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    pa[(s + 1) & 1][q] = *reinterpret_cast<const f32x4*>(ca + 16 * (s + 1) + 4 * q);
                    pb[(s + 1) & 1][q] = *reinterpret_cast<const f32x4*>(cb + 16 * (s + 1) + 4 * q);
                }
            }
            float v[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[2 * j] = bf_lo(yfr[s][j]); v[2 * j + 1] = bf_hi(yfr[s][j]); }
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = fmaf(v[4 * q + e], rstd, nmr);
                    const float y = fmaf(t, pa[s & 1][q][e], pb[s & 1][q][e]);
                    v[4 * q + e] = y * __builtin_amdgcn_rcpf(1.0f + __expf(-y));
                }
#pragma unroll
            for (int j = 0; j < 4; ++j) yfr[s][j] = pack_bf16(v[2 * j], v[2 * j + 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- phase D: h <- h + Linear3(yfr) ------------------------------------------------------------------------------------
    // One wave per SIMD cannot hide an HBM round trip per tile: a ring of four tiles of the residual h stays in flight.
    constexpr int NT = 16, RING = 4;
    f32x4 rres[RING][4];
    const size_t fbase = ((size_t)tb * NT * 4 * 64 + lane) * 4;          // + nt * 1024 floats + qi * 256
#pragma unroll
    for (int u = 0; u < RING; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) rres[u][q] = *reinterpret_cast<const f32x4*>(p.R + fbase + (size_t)u * 1024 + q * 256);
    char* Ctb = reinterpret_cast<char*>(p.Ct);
    const float const_on = (p.row_const != nullptr && row < p.n_const_rows) ? 1.0f : 0.0f;
    for (int nt0 = 0; nt0 < NT; nt0 += RING) {
#pragma unroll
        for (int u = 0; u < RING; ++u) {
            const int nt = nt0 + u;
            f32x16 a1;
#pragma unroll
            for (int qi = 0; qi < 4; ++qi) {
                const int col = nt * 32 + 16 * (qi >> 1) + 8 * h + 4 * (qi & 1);
                f32x4 b4 = *reinterpret_cast<const f32x4*>(sb3 + col);
                const f32x4 c4 = *reinterpret_cast<const f32x4*>(sconst + col);
#pragma unroll
                for (int e = 0; e < 4; ++e) a1[4 * qi + e] = fmaf(const_on, c4[e], b4[e]);
            }
            const int q0 = 64 + 2 * nt;
#pragma unroll
            for (int i = 0; i < 4; ++i) aw[0][i] = *reinterpret_cast<const u32x4*>(smem + (q0 & 7) * TL_STAGE + a_off + i * 32);
            __builtin_amdgcn_sched_barrier(0);
            run_stage(q0, 0, a1, yfr, true);
            __builtin_amdgcn_sched_barrier(0);
            run_stage(q0 + 1, 1, a1, yfr + 16, false);
            if (u & 1) __syncthreads();
            const size_t fidx = fbase + (size_t)nt * 1024;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                float v8[8];
#pragma unroll
                for (int qq = 0; qq < 2; ++qq) {
                    const int qi = 2 * c + qq;
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { o[e] = a1[4 * qi + e] + rres[u][qi][e]; v8[4 * qq + e] = o[e]; }
                    *reinterpret_cast<f32x4*>(p.Cf + fidx + qi * 256) = o;
                }
                u32x4 o;
                o.x = pack_bf16(v8[0], v8[1]); o.y = pack_bf16(v8[2], v8[3]);
                o.z = pack_bf16(v8[4], v8[5]); o.w = pack_bf16(v8[6], v8[7]);
                *reinterpret_cast<u32x4*>(Ctb + ((size_t)tb * (2 * NT) + 2 * nt + c) * 1024 + lane_off) = o;
            }
            // refill this ring slot with the tile four ahead (clamped: unconditional loads keep the vmcnt accounting exact)
            const int ntn = nt + RING < NT ? nt + RING : NT - 1;
#pragma unroll
            for (int q = 0; q < 4; ++q) rres[u][q] = *reinterpret_cast<const f32x4*>(p.R + fbase + (size_t)ntn * 1024 + q * 256);
        }
    }
}

int launch_tl_chain2(const TlChain2Args& a, hipStream_t s) {
    DSH_REQUIRE(a.M > 0 && a.G && a.W2 && a.W3 && a.film && a.R && a.Cf && a.Ct, "tl_chain2: null operand");
    DSH_REQUIRE(a.frames >= 64 && a.bmod > 0 && a.film_ld % 4 == 0 && a.film_off % 4 == 0, "tl_chain2: needs clips of >= 64 frames and the folded FiLM table");
    static bool attr = false;
    if (!attr) {
        DSH_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(tl_chain2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, CH_LDS));
        attr = true;
    }
    hipLaunchKernelGGL(tl_chain2_kernel, dim3(ceil_div(a.M, TL_TOK)), dim3(256), CH_LDS, s, a);
    DSH_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace dsh
