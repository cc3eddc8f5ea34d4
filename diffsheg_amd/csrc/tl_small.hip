// Token-per-lane Linears for WINDOW-CHAIN batches (a few hundred token rows): 32 tokens per block, one 32-feature tile per wave.
//
// The whole-chip kernels (tl_linear.hip, tl2.hip) own 128 / 256 tokens per block and stream W through LDS behind a block barrier.
// At B = 1 (344 rows under CFG, most of them block padding) they split N over the grid instead, and every N-split block repeats
// the prologue of all its 128 / 256 tokens — the LayerNorm + FiLM + SiLU pass of the StylizationBlock is 256 values per lane, about
// 6 us of an 11.4 us launch that sits 16 x 2 times on the critical path of an evaluation (profiles/r04_a_chain_trace_1.txt: 28 % of a
// window).  Here a block is ONE 32-token row block and FOUR 32-feature tiles:
//   * the four waves load a quarter of the block's rows each and exchange them through LDS (the MFMA B operand is then read from
//     LDS, 1 KB per instruction, conflict-free); the StylizationBlock conversion is done once, a quarter per wave;
//   * every wave streams the weights of ITS tile straight from the fragment-ordered copy (tl2_frag_index: 1 KB per wave
//     instruction) into registers — no LDS staging, no barrier in the MFMA sequence; the loads are requested at entry, right
//     behind the row loads (weights do not depend on the producer of the rows);
//   * padding blocks between the CFG halves are not launched.
// Arithmetic is that of the whole-chip kernels, operation for operation (row moments in the order of row_moments_bf16, the
// accumulator seeded with the bias, MFMAs in ascending k, the same epilogue expressions): a row's result does not depend on which
// kernel family its batch size selected (tests/test_gpu_eval.py::test_small_batch_kernels_are_bit_identical).
// Reference ops: transformer.py:86-97 (StylizationBlock), :106-108 (q|k|v), :172-173 (ffn), :284-289 (feat_proj).
#include "dsh_kernels.h"
#include "tl_common.h"

#include <algorithm>
#include <cstdlib>

namespace dsh {
namespace {

// 16 MFMAs of `acc`: A = 16 weight fragments in registers, B = the block's row fragments [s0, s0 + 16) in LDS (read 4 ahead)
__device__ __forceinline__ void mfma16_lds_b(f32x16& acc, const u32x4* a, const char* lds_lane) {
    u32x4 bw[2][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bw[0][i] = *reinterpret_cast<const u32x4*>(lds_lane + i * 1024);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (g + 1 < 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) bw[(g + 1) & 1][i] = *reinterpret_cast<const u32x4*>(lds_lane + ((g + 1) * 4 + i) * 1024);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[g * 4 + i]), __builtin_bit_cast(bf16x8, bw[g & 1][i]), acc, 0, 0, 0);
    }
}

__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (not __syncthreads: its vmcnt(0) would drain the weight loads in flight)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

constexpr int TLS_MAXCLIP = 4;             // FiLM prologue: clips a 32-token block may touch (clips of >= 16 frames, or batch <= 4)

// PRO / ACT as tl2_linear_kernel; HL: residual and result as hi / lo planes (tl_common.h), else one tiled bf16 output.
// Two blocks per CU (registers <= 256, LDS <= 64 KB): at a few thousand rows a launch is several rounds of blocks whose latency
// chains (rows -> LDS -> statistics -> MFMAs -> store) only overlap across co-resident blocks — feat_proj.1 at 32 chains 20.3 -> 16.0 us.
// (Two 32-token row blocks per block, every weight fragment feeding two MFMAs, was built and measured at 16 / 32 chains: q|k|v 21.5 ->
//  19.7 us, StylizationBlock 15.9 -> 18.3 us, 11.75 k vs 11.70 k frames/s — the regime is bound by those chains, not by the weight
//  re-reads; removed.)
template <int KD, int PRO, bool HL, int ACT>
__global__ __launch_bounds__(256, 2) void tls_linear_kernel(TlArgs p) {
    constexpr int NFRAG = KD / 16, OWN = NFRAG / 4, NCH = NFRAG / 16;
    // weight chunks (16 fragments = 64 registers) in flight; a K = 1024 tile's fourth chunk takes chunk 0's registers after its MFMAs.
    // (All four at entry, or the fourth right after the row registers are free: 256 + 64 registers, hipcc copies through AGPRs /
    //  scratch — feat_proj.1 11.6 vs 8.1 us, measured.)
    constexpr int RING = NCH < 3 ? NCH : 3;
    constexpr bool FOLD = PRO == 1 || PRO == 3;              // LayerNorm folded into W: the epilogue applies rstd / mean
    constexpr bool HAS_C = PRO == 2 && HL;                   // CFG-null row constant (StylizationBlock instantiation only)
    static_assert(PRO != 2 || KD == 512, "FiLM prologue: K = 512");
    // LDS: [NFRAG KB] rows (PRO 2: converted in place once every wave has its statistics) | PRO 2: FiLM rows of TLS_MAXCLIP clips
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ml = lane & 31, h = lane >> 5;
    const int bx = blockIdx.x;
    const int tb = bx < p.tls_nb0 ? bx : p.tls_tb1 + (bx - p.tls_nb0);     // 32-token block (second CFG half behind the gap)
    const int row = tb * 32 + ml;
    const int lane_off = ml * 32 + h * 16;
    const int NT = p.N / 32;
    const int nt = blockIdx.y * 4 + wave;                    // this wave's 32-feature tile

    // ---- this wave's quarter of the block's rows (requested first: the oldest entries of the in-order vmcnt queue) ----------
    u32x4 own[OWN];
    {
        const int s0 = wave * OWN;
        if (PRO == 3) {
            // un-materialised concat [latent 512 | audio_proj 256 | hubert 128 | expr_x0 128 (absent for the expression encoder)]
            const char* src;
            bool zero = false;
            if (s0 < 32) src = reinterpret_cast<const char*>(p.X) + ((size_t)tb * 32 + s0) * 1024;
            else if (s0 < 48) src = reinterpret_cast<const char*>(p.X1) + ((size_t)tb * 16 + (s0 - 32)) * 1024;
            else src = reinterpret_cast<const char*>(p.X2) + (size_t)tb * 8 * 1024;          // wave 3: hubert, then expr_x0
            const char* src3 = p.X3 ? reinterpret_cast<const char*>(p.X3) + (size_t)tb * 8 * 1024 : src;
            zero = p.X3 == nullptr;
#pragma unroll
            for (int i = 0; i < OWN; ++i) {
                const bool third = s0 >= 48 && i >= 8;
                u32x4 v = *reinterpret_cast<const u32x4*>((third ? src3 + (i - 8) * 1024 : src + i * 1024) + lane_off);
                if (third && zero) { v[0] = 0; v[1] = 0; v[2] = 0; v[3] = 0; }
                own[i] = v;
            }
        } else {
            const char* src = reinterpret_cast<const char*>(p.X) + ((size_t)tb * NFRAG + s0) * 1024 + lane_off;
#pragma unroll
            for (int i = 0; i < OWN; ++i) own[i] = *reinterpret_cast<const u32x4*>(src + i * 1024);
        }
    }
    // ---- folded FiLM rows [A | B] of the clips this block touches (PRO 2): one 16-byte piece per thread and clip, staged in LDS
    f32x4 prm[PRO == 2 ? TLS_MAXCLIP : 1];
    int clip0 = 0;
    if (PRO == 2) {
        const int rb = tb * 32, rrb = rb >= p.half_row0 ? rb - p.half_row0 : rb;
        clip0 = rrb / p.frames;
        const int nclip = (rrb + 31) / p.frames - clip0 + 1;
#pragma unroll
        for (int c = 0; c < TLS_MAXCLIP; ++c) {
            const int cc = c < nclip ? c : nclip - 1;
            prm[c] = *reinterpret_cast<const f32x4*>(p.film + (size_t)((clip0 + cc) % p.bmod) * p.film_ld + p.film_off + tid * 4);
        }
    }
    // ---- epilogue operands: residual planes, bias / folded-LayerNorm vectors --------------------------------------------------
    const size_t pidx = ((size_t)tb * (2 * NT) + 2 * nt) * 1024 + lane_off;     // bf16 fragment c of this tile at + c * 1024
    u32x4 rhi[2], rlo[2];
    if (HL) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            rhi[c] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(p.R) + pidx + c * 1024);
            rlo[c] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(p.Rlo) + pidx + c * 1024);
        }
    }
    f32x4 b4[4], c4[4];
#pragma unroll
    for (int qi = 0; qi < 4; ++qi) {
        const int col = nt * 32 + 16 * (qi >> 1) + 8 * h + 4 * (qi & 1);
        b4[qi] = *reinterpret_cast<const f32x4*>(p.bias + col);
        if (FOLD || HAS_C) c4[qi] = p.row_const ? *reinterpret_cast<const f32x4*>(p.row_const + col) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // ---- this wave's weight tile: the first RING chunks go in flight now ------------------------------------------------------
    const char* wsrc = reinterpret_cast<const char*>(p.W) + (size_t)nt * NFRAG * 1024 + lane * 16;
    u32x4 aw[RING][16];
#pragma unroll
    for (int c = 0; c < RING; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) aw[c][i] = *reinterpret_cast<const u32x4*>(wsrc + (c * 16 + i) * 1024);

    // ---- rows -> LDS, statistics, conversion ----------------------------------------------------------------------------------
    char* lds_lane = smem + lane * 16;
#pragma unroll
    for (int i = 0; i < OWN; ++i) *reinterpret_cast<u32x4*>(lds_lane + (wave * OWN + i) * 1024) = own[i];
    float* sfilm = reinterpret_cast<float*>(smem + NFRAG * 1024);
    if (PRO == 2) {
#pragma unroll
        for (int c = 0; c < TLS_MAXCLIP; ++c) *reinterpret_cast<f32x4*>(sfilm + c * 1024 + tid * 4) = prm[c];
    }
    lds_barrier();
    float rstd = 1.f, nmr = 0.f;
    if (PRO >= 1) {
        // row moments in the accumulation order of row_moments_bf16 (tl_common.h), fragments read back from LDS
        typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
        const bf16x2_t ones = __builtin_bit_cast(bf16x2_t, 0x3f803f80u);
        float sm[2] = {0.f, 0.f}, sq[2] = {0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NFRAG; ++s) {
            const u32x4 f = *reinterpret_cast<const u32x4*>(lds_lane + s * 1024);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t w = f[j];
                const bf16x2_t v = __builtin_bit_cast(bf16x2_t, w);
                sm[j & 1] = __builtin_amdgcn_fdot2_f32_bf16(v, ones, sm[j & 1], false);
                sq[j & 1] = __builtin_amdgcn_fdot2_f32_bf16(v, v, sq[j & 1], false);
            }
        }
        float sum = sm[0] + sm[1], sumsq = sq[0] + sq[1];
        sum += __shfl_xor(sum, 32, 64);
        sumsq += __shfl_xor(sumsq, 32, 64);
        const float kn = PRO == 3 ? (float)p.kreal : (float)KD;
        const float mean = sum / kn;
        sumsq = fmaxf(sumsq - sum * mean, 0.f);
        rstd = 1.0f / sqrtf(sumsq / kn + 1e-5f);
        nmr = -mean * rstd;
    }
    if (PRO == 2) {
        // y = SiLU(((x - mean) rstd) A + B) on this wave's quarter (expressions of tl_linear_kernel's prologue), written over the raw
        // rows once every wave has read them for its statistics; the raw fragment comes back from this wave's own LDS slot (keeping
        // the rows in registers across the statistics costs 32 registers)
        lds_barrier();
        const int rr = row >= p.half_row0 ? row - p.half_row0 : row;
        int ci = rr / p.frames - clip0;                      // rows past the last clip (block padding) may exceed the staged rows
        ci = ci < TLS_MAXCLIP ? ci : TLS_MAXCLIP - 1;
        const float* ca = sfilm + ci * 1024 + 16 * (wave * OWN) + 8 * h;
#pragma unroll
        for (int i = 0; i < OWN; ++i) {
            f32x4 fa[2], fb[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                fa[q] = *reinterpret_cast<const f32x4*>(ca + 16 * i + 4 * q);
                fb[q] = *reinterpret_cast<const f32x4*>(ca + 512 + 16 * i + 4 * q);
            }
            char* slot = lds_lane + (wave * OWN + i) * 1024;
            const u32x4 raw = *reinterpret_cast<const u32x4*>(slot);
            float v[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) { const uint32_t w = raw[j]; v[2 * j] = bf_lo(w); v[2 * j + 1] = bf_hi(w); }
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = fmaf(v[4 * q + e], rstd, nmr);
                    const float y = fmaf(t, fa[q][e], fb[q][e]);
                    v[4 * q + e] = y * __builtin_amdgcn_rcpf(1.0f + __expf(-y));
                }
            u32x4 o;
            o.x = pack_bf16(v[0], v[1]); o.y = pack_bf16(v[2], v[3]); o.z = pack_bf16(v[4], v[5]); o.w = pack_bf16(v[6], v[7]);
            *reinterpret_cast<u32x4*>(slot) = o;
        }
        lds_barrier();
    }

    // ---- the tile: accumulator seeded with the bias (+ CFG-null constant), K / 16 MFMAs in ascending k --------------------------
    const float const_on = (HAS_C && p.row_const != nullptr && row < p.n_const_rows) ? 1.0f : 0.0f;
    f32x16 acc;
#pragma unroll
    for (int qi = 0; qi < 4; ++qi)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[4 * qi + e] = FOLD ? 0.f : (HAS_C ? fmaf(const_on, c4[qi][e], b4[qi][e]) : b4[qi][e]);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        mfma16_lds_b(acc, aw[c % RING], lds_lane + c * 16 * 1024);
        if (c + RING < NCH) {
#pragma unroll
            for (int i = 0; i < 16; ++i) aw[c % RING][i] = *reinterpret_cast<const u32x4*>(wsrc + ((c + RING) * 16 + i) * 1024);
        }
    }
    // ---- epilogue (expressions of tl2_linear_kernel / tl_linear_kernel) -------------------------------------------------------
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        float v8[8];
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
            const int qi = 2 * c + qq;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = acc[4 * qi + e];
                if (FOLD) v = fmaf(v, rstd, fmaf(nmr, c4[qi][e], b4[qi][e]));
                if (ACT == ACT_GELU) v = gelu_fast(v);
                else if (ACT == ACT_SILU) v = v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
                v8[4 * qq + e] = v;
            }
        }
        if (HL) {
            hl_accumulate(v8, rhi[c], rlo[c]);
            u32x4 oh, ol;
            hl_split(v8, oh, ol);
            *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(p.Ct) + pidx + c * 1024) = oh;
            *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(p.Clo) + pidx + c * 1024) = ol;
        } else {
            u32x4 o;
            o.x = pack_bf16(v8[0], v8[1]); o.y = pack_bf16(v8[2], v8[3]); o.z = pack_bf16(v8[4], v8[5]); o.w = pack_bf16(v8[6], v8[7]);
            *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(p.Ct) + pidx + c * 1024) = o;
        }
    }
}

typedef void (*tls_kern_t)(TlArgs);
tls_kern_t tls_pick(const TlArgs& a, int pro) {
    const bool hl = a.Rlo != nullptr || a.Clo != nullptr;
    if (hl) {
        if (!(a.Rlo && a.Clo && a.R && a.Ct && !a.Cf && a.act == ACT_NONE)) return nullptr;
        if (a.K == 512 && pro == 2) return tls_linear_kernel<512, 2, true, ACT_NONE>;          // StylizationBlock + residual
        if (a.K == 1024 && pro == 0) return tls_linear_kernel<1024, 0, true, ACT_NONE>;        // feat_proj.3 + residual
        return nullptr;
    }
    if (a.R || a.Cf || !a.Ct) return nullptr;
    if (a.K == 512 && pro == 1 && a.act == ACT_NONE) return tls_linear_kernel<512, 1, false, ACT_NONE>;      // q|k|v
    if (a.K == 512 && pro == 0 && a.act == ACT_GELU) return tls_linear_kernel<512, 0, false, ACT_GELU>;      // ffn.linear1
    if (a.K == 1024 && pro == 0 && a.act == ACT_NONE) return tls_linear_kernel<1024, 0, false, ACT_NONE>;    // ffn.linear2
    if (a.K == 1024 && pro == 3 && a.act == ACT_SILU) return tls_linear_kernel<1024, 3, false, ACT_SILU>;    // feat_proj.1
    return nullptr;
}

}  // namespace

// a.W must be the FRAGMENT-ORDERED weight (tl2_frag_index); pro 1 / 3: a.bias = d, a.row_const = c of the folded LayerNorm
bool tls_linear_supported(const TlArgs& a, int pro) {
    if (a.M <= 0 || a.N <= 0 || a.N % 128 != 0 || !a.bias || !a.X || !a.W) return false;
    if (a.frames <= 0 || a.bmod <= 0) return false;
    const int Mc = a.frames * a.bmod;
    if (a.M != Mc && !(a.M > Mc && (a.M - Mc) % 32 == 0 && a.M - Mc >= Mc)) return false;    // one range of rows, or two CFG halves
    if ((pro == 1 || pro == 3) && !a.row_const) return false;
    if (pro == 2 && !(a.film && a.film_ld % 4 == 0 && a.film_off % 4 == 0 && std::min(31 / a.frames + 2, a.bmod) <= TLS_MAXCLIP)) return false;
    if (pro == 3 && !(a.X1 && a.X2 && a.kreal > 896 - 1 && a.kreal <= 1024)) return false;
    return tls_pick(a, pro) != nullptr;
}

int launch_tls_linear(const TlArgs& a, int pro, hipStream_t s) {
    g_tl_last_variant = 11;
    DSH_REQUIRE(tls_linear_supported(a, pro), "tls_linear: this launch is not covered by the window-chain kernels");
    DSH_REQUIRE(((uintptr_t)a.X % 16) == 0 && ((uintptr_t)a.W % 16) == 0, "tls_linear: operands must be 16-byte aligned");
    tls_kern_t fn = tls_pick(a, pro);
    static bool attr = false;
    if (!attr) {
        const tls_kern_t all[] = {tls_linear_kernel<512, 2, true, ACT_NONE>, tls_linear_kernel<1024, 0, true, ACT_NONE>,
                                  tls_linear_kernel<512, 1, false, ACT_NONE>, tls_linear_kernel<512, 0, false, ACT_GELU>,
                                  tls_linear_kernel<1024, 0, false, ACT_NONE>, tls_linear_kernel<1024, 3, false, ACT_SILU>};
        for (tls_kern_t k : all)
            DSH_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
        attr = true;
    }
    TlArgs b = a;
    const int Mc = a.frames * a.bmod, nbh = ceil_div(std::min(a.M, Mc), 32);
    b.tls_nb0 = nbh;
    b.tls_tb1 = a.M > Mc ? (a.M - Mc) / 32 : 0;
    const int nblocks = a.M > Mc ? 2 * nbh : nbh;
    const int lds = (a.K / 16) * 1024 + (pro == 2 ? TLS_MAXCLIP * 4096 : 0);
    hipLaunchKernelGGL(fn, dim3(nblocks, a.N / 128), dim3(256), lds, s, b);
    DSH_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace dsh
