// Layer-0 seed of the residual stream in ONE launch (round 6): h = joint_embed(x) + PE[:T] for both CFG halves, written straight as the
// hi / lo bf16 planes of the token-per-lane path.  Replaces pack_cols (x -> bf16 rows) + gemm_nt (joint_embed + bias + PE -> fp32
// row-major h0, 171 MB) + seed_stream (h0 -> planes of both halves): 230 us and three launches per encoder and evaluation at 950 clips.
// Reference: models/transformer.py:566-574 (joint_embed, positional table), :330-338 (the CFG-null half's first feat_proj is the
// per-layer constant feat_proj(null_cond_emb), added here to the null half as launch_seed_stream did).
//
// Token-per-lane form: a wave owns 32 tokens, its K = 16 NF input columns (112 for the 103 expression channels, 144 for the 129
// gesture channels; zero padded) are NF MFMA B fragments from the tiled bf16 copy of x; the 16 output tiles run one after the other
// with the weight fragments read straight from the fragment-ordered copy (tl2_frag_index; 512 x K bf16 = 112 - 144 KB, L2 resident),
// the next tile's fragments requested while this tile's MFMAs and epilogue run.  The launch is bound by its 4 KB of plane stores per
// token (342 MB at 950 clips); the MFMA work is 6 us.
#include <algorithm>
#include <cstdlib>

#include "dsh_common.h"
#include "dsh_kernels.h"
#include "tl_common.h"

namespace dsh {

struct TlJointArgs {
    const void* X;           // bf16 tiled [Mc, 16 NF]
    const void* W;           // fragment-ordered [512, 16 NF] (rows pi-permuted inside every 32-row tile)
    const float* bias;       // [512]
    const float* pe;         // [>= frames, 512] positional table
    const float* cnull;      // [512] or null: constant added to the CFG-null half (rows [0, Mc)); null: only rows [0, Mc) = h are written
    void* hi; void* lo;      // tiled bf16 planes [.., 512]
    int Mc, frames, row1;    // conditional half at rows [row1, row1 + Mc) (row1 a multiple of 32)
};

// TPB output tiles per block (blockIdx.y selects the group): 16 at whole-chip token counts, 4 at window-chain batches, where a wave walking
// all 16 tiles one after the other was 23 us on the critical path of every evaluation
template <int NF, int TPB>
__global__ __launch_bounds__(256, 2) void tl_joint_kernel(TlJointArgs p) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ml = lane & 31, h = lane >> 5;
    const int tb = blockIdx.x * 4 + wave;                     // 32-token block (rows past Mc: block padding, computed and stored into padding)
    const int row = tb * 32 + ml;
    const int lane_off = ml * 32 + h * 16;
    if (tb * 32 >= p.Mc) return;
    u32x4 xb[NF];
    const char* xr = reinterpret_cast<const char*>(p.X) + (size_t)tb * NF * 1024 + lane_off;
#pragma unroll
    for (int s = 0; s < NF; ++s) xb[s] = *reinterpret_cast<const u32x4*>(xr + s * 1024);
    const int nt0 = blockIdx.y * TPB;
    const char* wl = reinterpret_cast<const char*>(p.W) + lane * 16 + (size_t)nt0 * NF * 1024;
    u32x4 wa[2][NF];
#pragma unroll
    for (int s = 0; s < NF; ++s) wa[0][s] = *reinterpret_cast<const u32x4*>(wl + s * 1024);
    const float* per = p.pe + (size_t)(row % p.frames) * 512;
    char* hib = reinterpret_cast<char*>(p.hi);
    char* lob = reinterpret_cast<char*>(p.lo);
    const int tbc = tb + (p.cnull ? p.row1 / 32 : 0);         // the conditional half's token block (no CFG: the only half)
    static_for<TPB>([&](auto nt_tag) {
        constexpr int ntl = decltype(nt_tag)::value;
        const int nt = nt0 + ntl;
        if constexpr (ntl + 1 < TPB) {
#pragma unroll
            for (int s = 0; s < NF; ++s) wa[(ntl + 1) & 1][s] = *reinterpret_cast<const u32x4*>(wl + ((ntl + 1) * NF + s) * 1024);
        }
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int s = 0; s < NF; ++s)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wa[ntl & 1][s]), __builtin_bit_cast(bf16x8, xb[s]), acc, 0, 0, 0);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            float v[8], vn[8];
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2) {
                const int qi = 2 * c + q2, col = nt * 32 + 16 * (qi >> 1) + 8 * h + 4 * (qi & 1);
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias + col), p4 = *reinterpret_cast<const f32x4*>(per + col);
                f32x4 n4 = {0.f, 0.f, 0.f, 0.f};
                if (p.cnull) n4 = *reinterpret_cast<const f32x4*>(p.cnull + col);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x = (acc[4 * qi + e] + b4[e]) + p4[e];          // the GEMM epilogue's order: bias, then the residual operand (PE)
                    v[4 * q2 + e] = x;
                    vn[4 * q2 + e] = x + n4[e];
                }
            }
            u32x4 oh, ol;
            hl_split(v, oh, ol);
            const size_t ic = ((size_t)tbc * 32 + 2 * nt + c) * 1024 + lane_off;
            *reinterpret_cast<u32x4*>(hib + ic) = oh;
            *reinterpret_cast<u32x4*>(lob + ic) = ol;
            if (p.cnull) {
                hl_split(vn, oh, ol);
                const size_t in = ((size_t)tb * 32 + 2 * nt + c) * 1024 + lane_off;
                *reinterpret_cast<u32x4*>(hib + in) = oh;
                *reinterpret_cast<u32x4*>(lob + in) = ol;
            }
        }
    });
}

// x_tiled: bf16 tiled [Mc, 16 nf] (launch_tile_rows_bf16 of the encoder's channels of x), nf = 7 (K <= 112) or 9 (K <= 144)
int launch_tl_joint(const void* x_tiled, int nf, const void* wfrag, const float* bias, const float* pe, int frames, const float* cnull,
                    int Mc, int row1, void* hi, void* lo, hipStream_t s) {
    DSH_REQUIRE(x_tiled && wfrag && bias && pe && hi && lo && Mc > 0 && frames > 0, "tl_joint: null operand");
    DSH_REQUIRE(nf == 7 || nf == 9, "tl_joint: instantiated for 7 or 9 input fragments (K = 112 / 144)");
    DSH_REQUIRE(!cnull || (row1 % 32 == 0 && row1 >= Mc), "tl_joint: the conditional half starts on a 32-row boundary behind the null half");
    TlJointArgs a;
    a.X = x_tiled; a.W = wfrag; a.bias = bias; a.pe = pe; a.cnull = cnull; a.hi = hi; a.lo = lo; a.Mc = Mc; a.frames = frames; a.row1 = row1;
    const bool small = ceil_div(Mc, 128) < 64;                       // fewer than 64 token blocks: split the tiles over grid.y
    const dim3 grid(ceil_div(Mc, 128), small ? 4 : 1), block(256);
    if (nf == 7) { if (small) hipLaunchKernelGGL((tl_joint_kernel<7, 4>), grid, block, 0, s, a); else hipLaunchKernelGGL((tl_joint_kernel<7, 16>), grid, block, 0, s, a); }
    else { if (small) hipLaunchKernelGGL((tl_joint_kernel<9, 4>), grid, block, 0, s, a); else hipLaunchKernelGGL((tl_joint_kernel<9, 16>), grid, block, 0, s, a); }
    DSH_HIP_CHECK(hipGetLastError());
    return 0;
}

// audio_proj([mel | aud_feat]) of up to two motion encoders in one launch, written straight in the tiled layout the layers' concat reads
// (transformer.py:574).  Replaces, per encoder, an LDS-tiled GEMM into a row-major scratch tensor + tile_rows.  X = bf16 row-major [Mc, 256];
// W = per encoder 8 tiles x 16 k steps of 1 KB fragments (tl_aud_pack_audio_proj: fragment 16 t + s), encoder e at + e * 128 KB; bias [n][256].
__global__ __launch_bounds__(256, 2) void tl_aproj_kernel(const void* X, const void* W, const float* bias, int n_enc, void* out0, void* out1, int Mc) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ml = lane & 31, h = lane >> 5;
    const int tb = blockIdx.x * 4 + wave;
    if (tb * 32 >= Mc) return;
    int row = tb * 32 + ml;
    row = row < Mc ? row : Mc - 1;                              // padding lanes: last row, stored into padding rows of the tiled outputs
    const int lane_off = ml * 32 + h * 16;
    u32x4 xb[16];
    const char* xr = reinterpret_cast<const char*>(X) + (size_t)row * 512 + h * 16;
#pragma unroll
    for (int s = 0; s < 16; ++s) xb[s] = *reinterpret_cast<const u32x4*>(xr + s * 32);
    const char* wl = reinterpret_cast<const char*>(W) + lane * 16;
    for (int e = 0; e < n_enc; ++e) {
        char* ob = reinterpret_cast<char*>(e == 0 ? out0 : out1);
        const char* we = wl + (size_t)e * 128 * 1024;
        u32x4 wa[2][16];
#pragma unroll
        for (int s = 0; s < 16; ++s) wa[0][s] = *reinterpret_cast<const u32x4*>(we + s * 1024);
        static_for<8>([&](auto t_tag) {
            constexpr int t = decltype(t_tag)::value;
            if constexpr (t + 1 < 8) {
#pragma unroll
                for (int s = 0; s < 16; ++s) wa[(t + 1) & 1][s] = *reinterpret_cast<const u32x4*>(we + ((t + 1) * 16 + s) * 1024);
            }
            f32x16 acc;
#pragma unroll
            for (int qi = 0; qi < 4; ++qi) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias + e * 256 + t * 32 + 16 * (qi >> 1) + 8 * h + 4 * (qi & 1));
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[4 * qi + k] = b4[k];
            }
#pragma unroll
            for (int s = 0; s < 16; ++s)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wa[t & 1][s]), __builtin_bit_cast(bf16x8, xb[s]), acc, 0, 0, 0);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                u32x4 o;
                o.x = pack_bf16(acc[8 * c + 0], acc[8 * c + 1]); o.y = pack_bf16(acc[8 * c + 2], acc[8 * c + 3]);
                o.z = pack_bf16(acc[8 * c + 4], acc[8 * c + 5]); o.w = pack_bf16(acc[8 * c + 6], acc[8 * c + 7]);
                *reinterpret_cast<u32x4*>(ob + ((size_t)tb * 16 + 2 * t + c) * 1024 + lane_off) = o;
            }
        });
    }
}

int launch_tl_aproj(const void* x256, const void* wfrag, const float* bias, int n_enc, void* out0, void* out1, int Mc, hipStream_t s) {
    DSH_REQUIRE(x256 && wfrag && bias && out0 && (n_enc == 1 || (n_enc == 2 && out1)) && Mc > 0, "tl_aproj: null operand");
    hipLaunchKernelGGL(tl_aproj_kernel, dim3(ceil_div(Mc, 128)), dim3(256), 0, s, x256, wfrag, bias, n_enc, out0, out1, Mc);
    DSH_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace dsh
