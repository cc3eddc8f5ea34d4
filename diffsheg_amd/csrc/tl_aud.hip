// encoder_aud behind its attention in ONE launch (round 6): StylizationBlock of the attention branch -> + x -> FFN (128 -> 1024 -> GELU ->
// 128) -> StylizationBlock -> + h, for the D = 128 audio layer of the UniDiffuser (models/transformer.py:730-739 calling the layer of
// :300-346 with cond_proj off; :86-97 StylizationBlock, :169-181 FFN).  The layer's front (2 x audio, LayerNorm, q|k|v, attention)
// does not see the timestep and is computed once per condition (denoiser.hip aud_front); what remains per evaluation was six launches
// of LDS-tiled GEMMs and row kernels over [tokens, 128 .. 1024] row-major tensors — 425 us at 950 clips for 44 GFLOP, every
// intermediate (1024-wide hidden included) through HBM.
//
// Here a wave owns 32 tokens and keeps the whole chain in registers, token-per-lane style (tl3_ffn.hip at D = 128): the MFMA
// accumulator of one Linear, packed to bf16, IS the B operand of the next (output features leave the 32 x 32 MFMA as 2 x 8 consecutive
// features per lane because the weight rows are pi-permuted inside every 32-row tile).  A block is 8 waves = 256 tokens sharing one
// weight stream of 18 chunks of 32 KB through a three-slot LDS ring filled by LDS-DMA:
//   chunk 0: sa_block.proj_out Linear (4 tiles x 8 k steps) | chunks 1 .. 16: two hidden tiles each — [W1 tile (8 fragments) | the
//   matching K chunk of W2 (4 output tiles x 2 k steps)] x 2 | chunk 17: ffn.proj_out Linear.
// HBM traffic: 2 (y) + 4 (x) + 4 + 2 (outputs) bytes per value of a 128-wide row = 1.5 KB per token instead of ~9 KB.
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "dsh_common.h"
#include "dsh_kernels.h"
#include "tl_common.h"

namespace dsh {

struct TlAudArgs {
    const void* Y;            // bf16 row-major [Mc, 128]: attention output
    const float* X2;          // fp32 row-major [Mc, 128]: 2 x mel features (the layer's residual input)
    const void* W;            // weight stream, 18 chunks of 32 KB (tl_aud_pack_stream)
    const float* bias;        // [128 | 1024 | 128 | 128] = proj_out(sa) | linear1 | linear2 | proj_out(ffn)
    const float* film;        // FOLDED FiLM rows [A1 128 | B1 128 | A2 128 | B2 128] per embedding row (launch_film_fold, D = 128, 2 blocks)
    int film_ld, bmod, frames, Mc;
    float* out_f;             // fp32 row-major [Mc, 128]
    void* out_b; int ld_b;    // bf16 row-major, leading dimension ld_b (the right half of [audio | aud_feat]; its left half = the mel features)
    // audio_proj([audio | aud_feat]) of up to two motion encoders as further stages (transformer.py:574): chunks 18 + 4 e .. of the stream,
    // bias_ap [n_ap][256], tiled bf16 outputs [Mc, 256]; n_ap = 0: off
    int n_ap; const float* bias_ap; void* ap_out0; void* ap_out1;
};

constexpr int AUD_CHUNK = 32 * 1024;
constexpr int AUD_NCHUNK = 18;          // + 4 per audio_proj stage

__global__ __launch_bounds__(512) void tl_aud_tail_kernel(TlAudArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ml = lane & 31, h = lane >> 5;
    const int tb = blockIdx.x * 8 + wave;
    int row = tb * 32 + ml;
    const bool live = row < p.Mc;
    row = live ? row : p.Mc - 1;                                   // padding lanes compute on the last row and store nothing
    float* sbias = reinterpret_cast<float*>(smem + 3 * AUD_CHUNK);  // [1408] biases
    // weight stream: wave w moves bytes [4096 w, 4096 w + 4096) of every chunk
    const int nchunk = AUD_NCHUNK + 4 * p.n_ap;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.W), 0, nchunk * AUD_CHUNK, 0x00020000);
    const int wvoff = wave * 4096 + lane * 16;
    auto issue_chunk = [&](int c) {
        char* dst = smem + (c % 3) * AUD_CHUNK + wave * 4096;
#pragma unroll
        for (int k = 0; k < 4; ++k) dma_buf(k, wrsrc, wvoff, c * AUD_CHUNK, dst);
    };
    issue_chunk(0);
    issue_chunk(1);
    for (int i = tid; i < 1408; i += 512) sbias[i] = p.bias[i];
    // ---- rows: y fragments (bf16), residual x (fp32, lane-native quads), folded FiLM rows of this token's embedding row -------------
    u32x4 frag[8];
    const char* yr = reinterpret_cast<const char*>(p.Y) + (size_t)row * 256 + h * 16;
#pragma unroll
    for (int s = 0; s < 8; ++s) frag[s] = *reinterpret_cast<const u32x4*>(yr + s * 32);
    f32x4 res[4][4];                                                // residual stream h of this token: tile t, quad qi
    const float* xr = p.X2 + (size_t)row * 128;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int qi = 0; qi < 4; ++qi) res[t][qi] = *reinterpret_cast<const f32x4*>(xr + t * 32 + 16 * (qi >> 1) + 8 * h + 4 * (qi & 1));
    const float* fl = p.film + (size_t)((row / p.frames) % p.bmod) * p.film_ld;
    ln_frags<8, true>(frag, fl + 8 * h, fl + 128 + 8 * h, 128.f, 128.f);      // s1 = SiLU(LN(y) (1 + scale) + shift), in place
    const char* lds_lane = smem + lane * 16;
    auto phase_sync = [&](int c) {                                   // chunk c (and everything older) has landed for every wave
        asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (c + 2 < nchunk) issue_chunk(c + 2);                      // into the slot of chunk c - 1, which every wave has left
    };
    // one 128 -> 128 Linear from a 32-fragment chunk (tile t, k step s at fragment 8 t + s) on the operand `frag`
    auto linear128 = [&](const char* chunk, f32x16 (&acc)[4], const float* b) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int qi = 0; qi < 4; ++qi) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(b + t * 32 + 16 * (qi >> 1) + 8 * h + 4 * (qi & 1));
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[t][4 * qi + e] = b4[e];
            }
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const u32x4 a = *reinterpret_cast<const u32x4*>(chunk + (8 * t + s) * 1024);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, frag[s]), acc[t], 0, 0, 0);
            }
        }
    };
    // ---- phase 0: h = x + proj_out(s1); X = bf16(h) becomes the FFN's operand -----------------------------------------------------------
    phase_sync(0);
    {
        f32x16 acc[4];
        linear128(lds_lane + 0 * AUD_CHUNK, acc, sbias);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                float v[8];
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x = acc[t][4 * (2 * c + q2) + e] + res[t][2 * c + q2][e];
                        res[t][2 * c + q2][e] = x;
                        v[4 * q2 + e] = x;
                    }
                u32x4 o;
                o.x = pack_bf16(v[0], v[1]); o.y = pack_bf16(v[2], v[3]); o.z = pack_bf16(v[4], v[5]); o.w = pack_bf16(v[6], v[7]);
                frag[2 * t + c] = o;
            }
    }
    // ---- phases 1 .. 16: y2 = GELU(X W1^T + b1) W2^T + b2, two hidden tiles per chunk -------------------------------------------------
    f32x16 y2[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int qi = 0; qi < 4; ++qi) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(sbias + 128 + 1024 + t * 32 + 16 * (qi >> 1) + 8 * h + 4 * (qi & 1));
#pragma unroll
            for (int e = 0; e < 4; ++e) y2[t][4 * qi + e] = b4[e];
        }
    for (int c = 1; c <= 16; ++c) {
        phase_sync(c);
        const char* chunk = lds_lane + (c % 3) * AUD_CHUNK;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = 2 * (c - 1) + u;                          // hidden tile
            const char* cu = chunk + u * 16 * 1024;
            f32x16 hid;
#pragma unroll
            for (int qi = 0; qi < 4; ++qi) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(sbias + 128 + j * 32 + 16 * (qi >> 1) + 8 * h + 4 * (qi & 1));
#pragma unroll
                for (int e = 0; e < 4; ++e) hid[4 * qi + e] = b4[e];
            }
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const u32x4 a = *reinterpret_cast<const u32x4*>(cu + s * 1024);
                hid = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, frag[s]), hid, 0, 0, 0);
            }
            u32x4 hb[2];
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                float g[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) g[e] = gelu_fast(hid[8 * cc + e]);
                hb[cc].x = pack_bf16(g[0], g[1]); hb[cc].y = pack_bf16(g[2], g[3]); hb[cc].z = pack_bf16(g[4], g[5]); hb[cc].w = pack_bf16(g[6], g[7]);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const u32x4 a = *reinterpret_cast<const u32x4*>(cu + (8 + 2 * t + ks) * 1024);
                    y2[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, hb[ks]), y2[t], 0, 0, 0);
                }
        }
    }
    // ---- s2 = SiLU(LN(y2) (1 + scale) + shift) from the fp32 accumulators ------------------------------------------------------------------
    {
        float sum = 0.f, sq = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) { sum += y2[t][e]; sq = fmaf(y2[t][e], y2[t][e], sq); }
        sum += __shfl_xor(sum, 32, 64);
        sq += __shfl_xor(sq, 32, 64);
        const float mean = sum * (1.0f / 128.f);
        const float var = fmaxf(sq * (1.0f / 128.f) - mean * mean, 0.f);
        const float rstd = 1.0f / sqrtf(var + 1e-5f), nmr = -mean * rstd;
        const float* ca = fl + 256 + 8 * h;
        const float* cb = fl + 384 + 8 * h;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                float v[8];
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2) {
                    const f32x4 a4 = *reinterpret_cast<const f32x4*>(ca + 16 * (2 * t + c) + 4 * q2), b4 = *reinterpret_cast<const f32x4*>(cb + 16 * (2 * t + c) + 4 * q2);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float n = fmaf(y2[t][4 * (2 * c + q2) + e], rstd, nmr);
                        const float y = fmaf(n, a4[e], b4[e]);
                        v[4 * q2 + e] = y * __builtin_amdgcn_rcpf(1.0f + __expf(-y));
                    }
                }
                u32x4 o;
                o.x = pack_bf16(v[0], v[1]); o.y = pack_bf16(v[2], v[3]); o.z = pack_bf16(v[4], v[5]); o.w = pack_bf16(v[6], v[7]);
                frag[2 * t + c] = o;
            }
    }
    // ---- phase 17: aud_feat = h + proj_out(s2) -----------------------------------------------------------------------------------------
    phase_sync(17);
    {
        f32x16 acc[4];
        linear128(lds_lane + (17 % 3) * AUD_CHUNK, acc, sbias + 128 + 1024 + 128);
        float* of = p.out_f + (size_t)row * 128;
        char* ob = reinterpret_cast<char*>(p.out_b) + (size_t)row * p.ld_b * 2;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                float v[8];
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2) {
                    f32x4 o4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { o4[e] = acc[t][4 * (2 * c + q2) + e] + res[t][2 * c + q2][e]; v[4 * q2 + e] = o4[e]; }
                    if (live) *reinterpret_cast<f32x4*>(of + t * 32 + 16 * c + 8 * h + 4 * q2) = o4;
                }
                u32x4 o;
                o.x = pack_bf16(v[0], v[1]); o.y = pack_bf16(v[2], v[3]); o.z = pack_bf16(v[4], v[5]); o.w = pack_bf16(v[6], v[7]);
                if (live) *reinterpret_cast<u32x4*>(ob + (t * 32 + 16 * c + 8 * h) * 2) = o;
                frag[2 * t + c] = o;      // aud_feat as the upper half of the audio_proj operand (padding lanes only feed padding rows of the tiled outputs)
            }
    }
    if (p.n_ap == 0) return;
    // ---- audio_proj stages: [mel 128 | aud_feat 128] -> 256 per motion encoder, written in the TILED layout the layers' concat reads ----------
    u32x4 mel[8];
    {
        const char* mr = reinterpret_cast<const char*>(p.out_b) - 256 + (size_t)row * p.ld_b * 2 + h * 16;     // left half of the same rows
#pragma unroll
        for (int s = 0; s < 8; ++s) mel[s] = *reinterpret_cast<const u32x4*>(mr + s * 32);
    }
    const int lane_off = ml * 32 + h * 16;
    for (int e = 0; e < p.n_ap; ++e) {
        char* apo = reinterpret_cast<char*>(e == 0 ? p.ap_out0 : p.ap_out1);
        for (int cc = 0; cc < 4; ++cc) {
            const int c = 18 + 4 * e + cc;
            phase_sync(c);
            const char* chunk = lds_lane + (c % 3) * AUD_CHUNK;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int t = 2 * cc + u;                           // output tile (32 of the 256 features)
                f32x16 acc;
#pragma unroll
                for (int qi = 0; qi < 4; ++qi) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias_ap + e * 256 + t * 32 + 16 * (qi >> 1) + 8 * h + 4 * (qi & 1));
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[4 * qi + k] = b4[k];
                }
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const u32x4 a = *reinterpret_cast<const u32x4*>(chunk + (16 * u + s) * 1024);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, s < 8 ? mel[s & 7] : frag[s & 7]), acc, 0, 0, 0);
                }
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) {
                    u32x4 o;
                    o.x = pack_bf16(acc[8 * c2 + 0], acc[8 * c2 + 1]); o.y = pack_bf16(acc[8 * c2 + 2], acc[8 * c2 + 3]);
                    o.z = pack_bf16(acc[8 * c2 + 4], acc[8 * c2 + 5]); o.w = pack_bf16(acc[8 * c2 + 6], acc[8 * c2 + 7]);
                    *reinterpret_cast<u32x4*>(apo + ((size_t)tb * 16 + 2 * t + c2) * 1024 + lane_off) = o;
                }
            }
        }
    }
}

// The 18-chunk weight stream from the layer's row-major fp32 weights: ws1 / ws2 [128,128] (the two StylizationBlock Linears), w1 [1024,128],
// w2 [128,1024]; rows are pi-permuted inside every 32-row tile (tl_weight_src_row), values rounded to bf16.  `st` receives 18 * 16384 elements.
void tl_aud_pack_stream(const float* ws1, const float* w1, const float* w2, const float* ws2, uint16_t* st) {
    constexpr size_t CH = 16384;
    auto put = [](uint16_t* dst, float v) { *dst = f32_to_bf16(v).v; };
    auto pack128 = [&](const float* w, uint16_t* c) {            // fragment 8 t + s: tile t, k step s
        for (int t = 0; t < 4; ++t)
            for (int n = 0; n < 32; ++n) {
                const int sr = 32 * t + (tl_weight_src_row(n) & 31);
                for (int k = 0; k < 128; ++k) put(&c[(((size_t)(8 * t + (k >> 4)) * 64) + (n + 32 * ((k >> 3) & 1))) * 8 + (k & 7)], w[(size_t)sr * 128 + k]);
            }
    };
    pack128(ws1, st);
    for (int j = 0; j < 32; ++j) {
        uint16_t* c = st + (size_t)(1 + (j >> 1)) * CH + (size_t)(j & 1) * 16 * 512;        // 16 fragments (512 elements each) per hidden tile
        for (int n = 0; n < 32; ++n) {                            // W1 tile j: fragments 0 .. 7
            const int sr = 32 * j + (tl_weight_src_row(n) & 31);
            for (int k = 0; k < 128; ++k) put(&c[(((size_t)(k >> 4) * 64) + (n + 32 * ((k >> 3) & 1))) * 8 + (k & 7)], w1[(size_t)sr * 128 + k]);
        }
        for (int t = 0; t < 4; ++t)                               // K chunk j of W2: fragment 8 + 2 t + ks
            for (int ks = 0; ks < 2; ++ks)
                for (int ln = 0; ln < 64; ++ln) {
                    const int sr = 32 * t + (tl_weight_src_row(ln & 31) & 31);
                    for (int jj = 0; jj < 8; ++jj)
                        put(&c[((size_t)(8 + 2 * t + ks) * 64 + ln) * 8 + jj], w2[(size_t)sr * 1024 + 32 * j + 16 * ks + 8 * (ln >> 5) + jj]);
                }
    }
    pack128(ws2, st + (size_t)17 * CH);
}

// audio_proj [256, 256] (row-major fp32) as 4 chunks: fragment 16 (t & 1) + s of chunk t >> 1 = output tile t, k step s; `st` receives 4 * 16384 elements
void tl_aud_pack_audio_proj(const float* w, uint16_t* st) {
    for (int t = 0; t < 8; ++t)
        for (int n = 0; n < 32; ++n) {
            const int sr = 32 * t + (tl_weight_src_row(n) & 31);
            for (int k = 0; k < 256; ++k)
                st[(((size_t)(16 * t + (k >> 4)) * 64) + (n + 32 * ((k >> 3) & 1))) * 8 + (k & 7)] = f32_to_bf16(w[(size_t)sr * 256 + k]).v;
        }
}

int launch_tl_aud_tail(const void* Y, const float* X2, const void* Wst, const float* bias, const float* film, int film_ld, int bmod, int frames,
                       int Mc, float* out_f, void* out_b, int ld_b, hipStream_t s, int n_ap, const float* bias_ap, void* ap_out0, void* ap_out1) {
    DSH_REQUIRE(Y && X2 && Wst && bias && film && out_f && out_b && Mc > 0 && frames > 0 && bmod > 0, "tl_aud_tail: null operand");
    DSH_REQUIRE(film_ld % 4 == 0 && ld_b % 8 == 0 && ((uintptr_t)out_b % 16) == 0, "tl_aud_tail: alignment");
    constexpr int lds = 3 * AUD_CHUNK + 1408 * 4;
    static const bool attr = hipFuncSetAttribute(reinterpret_cast<const void*>(tl_aud_tail_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess;
    DSH_REQUIRE(attr, "tl_aud_tail: hipFuncSetAttribute failed");
    TlAudArgs a;
    a.Y = Y; a.X2 = X2; a.W = Wst; a.bias = bias; a.film = film; a.film_ld = film_ld; a.bmod = bmod; a.frames = frames; a.Mc = Mc;
    a.out_f = out_f; a.out_b = out_b; a.ld_b = ld_b;
    DSH_REQUIRE(n_ap >= 0 && n_ap <= 2 && (n_ap == 0 || (bias_ap && ap_out0 && ld_b == 256)) && (n_ap < 2 || ap_out1), "tl_aud_tail: audio_proj stages");
    a.n_ap = n_ap; a.bias_ap = bias_ap; a.ap_out0 = ap_out0; a.ap_out1 = ap_out1;
    hipLaunchKernelGGL(tl_aud_tail_kernel, dim3(ceil_div(Mc, 256)), dim3(512), lds, s, a);
    DSH_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace dsh
