// Output head of a motion encoder in ONE launch (round 6): o = out(h) for both CFG halves, eps = o_u + s (o_c - o_u), and for the
// expression encoder x0 = c1 x - c2 eps with its tiled bf16 copy for the gesture encoder's concat (models/transformer.py:582-586,
// :717-724, :749).  Replaces the `out` Linear over all rows (fp32 row-major [M, 128 | 160], 107 MB at 950 clips) + cfg_mix (reads it
// back) + tile_rows (expression x0 -> tiled bf16): three launches and 0.3 GB per encoder and evaluation.
//
// A wave owns 32 tokens of the batch: the hi-plane fragments of their CFG-null rows, then of their conditional rows, are the MFMA B
// operand of the NTO output tiles (weights straight from the fragment-ordered copy, 128 - 160 KB, L2 resident; accumulators seeded with
// the bias, ascending k — the arithmetic of the `out` Linear it replaces, operation for operation).  The mixed tile goes through a
// wave-private LDS slab so that eps / x0 leave as coalesced row segments (token-per-lane stores would touch 64 lines per instruction).
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "dsh_common.h"
#include "dsh_kernels.h"
#include "tl_common.h"

namespace dsh {

struct TlOutArgs {
    const void* H;            // hi plane of the residual stream, tiled bf16 [.., 512]
    const void* W;            // `out` in fragment order [32 NTO, 512]
    const float* bias;        // [32 NTO]
    int Mc, row1, has_null, frames, w, c0, C;   // conditional half at rows [row1, row1 + Mc); w real output columns at column c0 of the [., C] tensors
    float cond_scale;
    float* eps;               // [Mc, C] fp32
    const float* x; const float* c1; const float* c2;   // x0 = c1[clip] x - c2[clip] eps (x0 == null: off)
    float* x0; void* x0t;     // [Mc, w] fp32 row-major and tiled bf16 [Mc, 128] (zero padded)
};

template <int NTO>
__global__ __launch_bounds__(256, 1) void tl_out_mix_kernel(TlOutArgs p) {
    constexpr int LDW = NTO * 32 + 4;                          // padded slab row (floats)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ml = lane & 31, h = lane >> 5;
    const int tb = blockIdx.x * 4 + wave;
    if (tb * 32 >= p.Mc) return;
    const int lane_off = ml * 32 + h * 16;
    float* slab = reinterpret_cast<float*>(smem) + (size_t)wave * 32 * LDW;
    const char* wl = reinterpret_cast<const char*>(p.W) + lane * 16;
    f32x16 mix[NTO];
    // one CFG half: the 32 fragments of its rows, then the NTO tiles (weights 8 fragments at a time: the loads of a whole tile in flight
    // at once cost 128 registers next to the 128 of the row fragments)
    auto half_pass = [&](auto half_tag, int tbh) {
        constexpr int HALF = decltype(half_tag)::value;           // 0: CFG-null rows -> mix = o_u; 1: conditional rows -> mix = o_u + s (o_c - o_u) (or o_c)
        u32x4 frag[32];
        const char* hr = reinterpret_cast<const char*>(p.H) + (size_t)tbh * 32 * 1024 + lane_off;
#pragma unroll
        for (int s = 0; s < 32; ++s) frag[s] = *reinterpret_cast<const u32x4*>(hr + s * 1024);
        // weight fragments in groups of 8, the next group requested before the MFMAs of the current one (two register sets)
        u32x4 a[2][8];
#pragma unroll
        for (int i = 0; i < 8; ++i) a[0][i] = *reinterpret_cast<const u32x4*>(wl + i * 1024);
        f32x16 acc;
        static_for<NTO * 4>([&](auto gi_tag) {
            constexpr int gi = decltype(gi_tag)::value, nt = gi >> 2, g = gi & 3;
            if constexpr (gi + 1 < NTO * 4) {
#pragma unroll
                for (int i = 0; i < 8; ++i) a[(gi + 1) & 1][i] = *reinterpret_cast<const u32x4*>(wl + ((gi + 1) * 8 + i) * 1024);
            }
            if constexpr (g == 0) {
#pragma unroll
                for (int qi = 0; qi < 4; ++qi) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias + nt * 32 + 16 * (qi >> 1) + 8 * h + 4 * (qi & 1));
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[4 * qi + e] = b4[e];
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[gi & 1][i]), __builtin_bit_cast(bf16x8, frag[g * 8 + i]), acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (g == 3) {
                if constexpr (HALF == 0) mix[nt] = acc;
                else if (p.has_null) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) mix[nt][e] = __fadd_rn(mix[nt][e], __fmul_rn(p.cond_scale, __fsub_rn(acc[e], mix[nt][e])));
                } else mix[nt] = acc;
            }
        });
    };
    if (p.has_null) half_pass(std::integral_constant<int, 0>{}, tb);
    else {
#pragma unroll
        for (int nt = 0; nt < NTO; ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e) mix[nt][e] = 0.f;
    }
    half_pass(std::integral_constant<int, 1>{}, tb + (p.has_null ? p.row1 / 32 : 0));
    // lane-native tiles -> slab [token][feature]
#pragma unroll
    for (int nt = 0; nt < NTO; ++nt)
#pragma unroll
        for (int qi = 0; qi < 4; ++qi) {
            f32x4 v = {mix[nt][4 * qi], mix[nt][4 * qi + 1], mix[nt][4 * qi + 2], mix[nt][4 * qi + 3]};
            *reinterpret_cast<f32x4*>(slab + ml * LDW + nt * 32 + 16 * (qi >> 1) + 8 * h + 4 * (qi & 1)) = v;
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // wave-private slab: no barrier
    const int row0 = tb * 32;
    for (int r = 0; r < 32; ++r) {
        const int row = row0 + r;
        if (row >= p.Mc) break;
        const int b = row / p.frames;
        const float k1 = p.x0 ? p.c1[b] : 0.f, k2 = p.x0 ? p.c2[b] : 0.f;
        for (int col = lane; col < NTO * 32; col += 64) {
            const float e = slab[r * LDW + col];
            float z = 0.f;
            if (col < p.w) {
                p.eps[(size_t)row * p.C + p.c0 + col] = e;
                if (p.x0) {
                    z = __fsub_rn(__fmul_rn(k1, p.x[(size_t)row * p.C + p.c0 + col]), __fmul_rn(k2, e));
                    p.x0[(size_t)row * p.w + col] = z;
                }
            }
            if (p.x0) slab[r * LDW + col] = z;                 // (columns >= w: the zero padding of the tiled copy)
        }
    }
    if (p.x0t) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        char* xt = reinterpret_cast<char*>(p.x0t) + (size_t)tb * 8 * 1024 + lane_off;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (16 * s + 8 * h + j < NTO * 32 && row0 + ml < p.Mc) ? slab[ml * LDW + 16 * s + 8 * h + j] : 0.f;
            u32x4 o;
            o.x = pack_bf16(v[0], v[1]); o.y = pack_bf16(v[2], v[3]); o.z = pack_bf16(v[4], v[5]); o.w = pack_bf16(v[6], v[7]);
            *reinterpret_cast<u32x4*>(xt + s * 1024) = o;
        }
    }
}

int launch_tl_out_mix(const void* hi, const void* wfrag, const float* bias, int n_out_padded, int Mc, int row1, int has_null, int frames, int w,
                      int c0, int C, float cond_scale, float* eps, const float* x, const float* c1, const float* c2, float* x0, void* x0_tiled,
                      hipStream_t s) {
    DSH_REQUIRE(hi && wfrag && bias && eps && Mc > 0 && frames > 0 && w > 0 && w <= n_out_padded, "tl_out_mix: null operand");
    DSH_REQUIRE(n_out_padded == 128 || n_out_padded == 160, "tl_out_mix: instantiated for 4 or 5 output tiles (103 / 129 channels)");
    DSH_REQUIRE(!has_null || row1 % 32 == 0, "tl_out_mix: the conditional half starts on a 32-row boundary");
    DSH_REQUIRE(!x0 || (x && c1 && c2), "tl_out_mix: x0 needs x, c1, c2");
    DSH_REQUIRE(!x0_tiled || (x0 && n_out_padded == 128), "tl_out_mix: the tiled x0 copy is 128 columns wide");
    TlOutArgs a;
    a.H = hi; a.W = wfrag; a.bias = bias; a.Mc = Mc; a.row1 = row1; a.has_null = has_null; a.frames = frames; a.w = w; a.c0 = c0; a.C = C;
    a.cond_scale = cond_scale; a.eps = eps; a.x = x; a.c1 = c1; a.c2 = c2; a.x0 = x0; a.x0t = x0_tiled;
    const dim3 grid(ceil_div(Mc, 128)), block(256);
    if (n_out_padded == 128) {
        constexpr int lds = 4 * 32 * (4 * 32 + 4) * 4;
        static const bool attr = hipFuncSetAttribute(reinterpret_cast<const void*>(tl_out_mix_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess;
        DSH_REQUIRE(attr, "tl_out_mix: hipFuncSetAttribute failed");
        hipLaunchKernelGGL(tl_out_mix_kernel<4>, grid, block, lds, s, a);
    } else {
        constexpr int lds = 4 * 32 * (5 * 32 + 4) * 4;
        static const bool attr = hipFuncSetAttribute(reinterpret_cast<const void*>(tl_out_mix_kernel<5>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess;
        DSH_REQUIRE(attr, "tl_out_mix: hipFuncSetAttribute failed");
        hipLaunchKernelGGL(tl_out_mix_kernel<5>, grid, block, lds, s, a);
    }
    DSH_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace dsh
