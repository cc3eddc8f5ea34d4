// Row-wise (per-token) kernels of the DiffSHEG denoiser: one 64-lane wavefront owns one token row,
// loads are lane-strided (coalesced 256 B per wave instruction), statistics are wavefront
// reductions in fp32 regardless of the storage type.
//
//   ln_rows            sa_block.norm, with the CFG-null feat_proj constant folded in
//                      (models/transformer.py:119-125, :326-338)
//   ln_film_silu_rows  StylizationBlock: SiLU(LN(y)*(1+scale)+shift)        (:86-97)
//   concat_ln_rows     feat_proj.0 LayerNorm over the un-materialised concat (:304-312, :284-285)
//   im2col3_rows       Conv1d(k=3,p=1) patches for hubert_encoder           (:437-442)
//   temb_rows          timestep_embedding                                    (:42-59)
//   pack_cols          split x into gesture|expression operands              (:741)
//   cfg_mix            classifier-free mix + expression x0                   (:585-586, :717-724)
#include <algorithm>

#include "dsh_common.h"
#include "dsh_kernels.h"

namespace dsh {

constexpr int ROWS_PER_BLOCK = 4;   // 4 waves / block
constexpr int MAX_PER_LANE = 16;    // supports D <= 1024

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void ln_rows_kernel(float* h, int ldh, int M, int D, const float* pre_add,
                                                      int n_pre_rows, const float* gamma, const float* beta,
                                                      T* out, int ldo) {
    const int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= M) return;
    float* hr = h + (size_t)row * ldh;
    const bool add = pre_add != nullptr && row < n_pre_rows;
    float v[MAX_PER_LANE];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAX_PER_LANE; ++i) {
        const int c = lane + 64 * i;
        float x = 0.f;
        if (c < D) {
            x = hr[c];
            if (add) { x += pre_add[c]; hr[c] = x; }
        }
        v[i] = x;
        s += x;
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAX_PER_LANE; ++i) {
        const int c = lane + 64 * i;
        const float d = (c < D) ? v[i] - mean : 0.f;
        q += d * d;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + 1e-5f);
    T* orow = out + (size_t)row * ldo;
#pragma unroll
    for (int i = 0; i < MAX_PER_LANE; ++i) {
        const int c = lane + 64 * i;
        if (c < D) orow[c] = from_f32<T>((v[i] - mean) * rstd * gamma[c] + beta[c]);
    }
}

template <typename T>
int launch_ln_rows(float* h, int ldh, int M, int D, const float* pre_add, int n_pre_rows, const float* gamma,
                   const float* beta, T* out, int ldo, hipStream_t s) {
    DSH_REQUIRE(D <= 64 * MAX_PER_LANE, "ln_rows: D too large");
    hipLaunchKernelGGL(ln_rows_kernel<T>, dim3(ceil_div(M, ROWS_PER_BLOCK)), dim3(256), 0, s, h, ldh, M, D, pre_add,
                       n_pre_rows, gamma, beta, out, ldo);
    DSH_HIP_CHECK(hipGetLastError());
    return 0;
}
template int launch_ln_rows<float>(float*, int, int, int, const float*, int, const float*, const float*, float*, int, hipStream_t);
template int launch_ln_rows<bf16>(float*, int, int, int, const float*, int, const float*, const float*, bf16*, int, hipStream_t);

// ------------------------------------------------------------------------------------------------
template <typename TI, typename T>
__global__ __launch_bounds__(256) void ln_film_silu_rows_kernel(const TI* y, int ldy, int M, int D, const float* gamma,
                                                                const float* beta, const float* film, int film_ld,
                                                                int film_off, int frames, int bmod, T* out, int ldo) {
    const int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= M) return;
    const TI* yr = y + (size_t)row * ldy;
    float v[MAX_PER_LANE];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAX_PER_LANE; ++i) {
        const int c = lane + 64 * i;
        const float x = (c < D) ? to_f32<TI>(yr[c]) : 0.f;
        v[i] = x;
        s += x;
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAX_PER_LANE; ++i) {
        const int c = lane + 64 * i;
        const float d = (c < D) ? v[i] - mean : 0.f;
        q += d * d;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + 1e-5f);
    const int b = (row / frames) % bmod;
    const float* fs = film + (size_t)b * film_ld + film_off;   // [scale(D) | shift(D)]
    T* orow = out + (size_t)row * ldo;
#pragma unroll
    for (int i = 0; i < MAX_PER_LANE; ++i) {
        const int c = lane + 64 * i;
        if (c < D) {
            const float n = (v[i] - mean) * rstd * gamma[c] + beta[c];
            orow[c] = from_f32<T>(silu_f(n * (1.0f + fs[c]) + fs[D + c]));
        }
    }
}

template <typename TI, typename T>
int launch_ln_film_silu_rows(const TI* y, int ldy, int M, int D, const float* gamma, const float* beta,
                             const float* film, int film_ld, int film_off, int frames, int bmod, T* out, int ldo,
                             hipStream_t s) {
    DSH_REQUIRE(D <= 64 * MAX_PER_LANE, "ln_film_silu_rows: D too large");
    hipLaunchKernelGGL((ln_film_silu_rows_kernel<TI, T>), dim3(ceil_div(M, ROWS_PER_BLOCK)), dim3(256), 0, s, y, ldy, M,
                       D, gamma, beta, film, film_ld, film_off, frames, bmod, out, ldo);
    DSH_HIP_CHECK(hipGetLastError());
    return 0;
}
template int launch_ln_film_silu_rows<float, float>(const float*, int, int, int, const float*, const float*, const float*, int, int, int, int, float*, int, hipStream_t);
template int launch_ln_film_silu_rows<float, bf16>(const float*, int, int, int, const float*, const float*, const float*, int, int, int, int, bf16*, int, hipStream_t);
template int launch_ln_film_silu_rows<bf16, bf16>(const bf16*, int, int, int, const float*, const float*, const float*, int, int, int, int, bf16*, int, hipStream_t);

// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ float seg_load(const ConcatSegs& sg, int c, int row) {
    // segments are laid out back to back in the virtual concat row
    if (c < sg.w0) return sg.p0[(size_t)row * sg.ld0 + c];                       // latent h (fp32)
    c -= sg.w0;
    if (c < sg.w1) return to_f32<T>(reinterpret_cast<const T*>(sg.p1)[(size_t)row * sg.ld1 + c]);
    c -= sg.w1;
    if (c < sg.w2) return to_f32<T>(reinterpret_cast<const T*>(sg.p2)[(size_t)row * sg.ld2 + c]);
    c -= sg.w2;
    return sg.p3[(size_t)row * sg.ld3 + c];                                       // expr_x0 (fp32)
}

template <typename T>
__global__ __launch_bounds__(256) void concat_ln_rows_kernel(ConcatSegs sg, int M, const float* gamma, const float* beta,
                                                             T* out, int ldo, int Ppad) {
    const int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= M) return;
    const int P = sg.w0 + sg.w1 + sg.w2 + sg.w3;
    float v[MAX_PER_LANE];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAX_PER_LANE; ++i) {
        const int c = lane + 64 * i;
        const float x = (c < P) ? seg_load<T>(sg, c, row) : 0.f;
        v[i] = x;
        s += x;
    }
    const float mean = wave_sum(s) / (float)P;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAX_PER_LANE; ++i) {
        const int c = lane + 64 * i;
        const float d = (c < P) ? v[i] - mean : 0.f;
        q += d * d;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)P + 1e-5f);
    T* orow = out + (size_t)row * ldo;
#pragma unroll
    for (int i = 0; i < MAX_PER_LANE; ++i) {
        const int c = lane + 64 * i;
        if (c < P) orow[c] = from_f32<T>((v[i] - mean) * rstd * gamma[c] + beta[c]);
        else if (c < Ppad) orow[c] = from_f32<T>(0.f);
    }
}

template <typename T>
int launch_concat_ln_rows(const ConcatSegs& sg, int M, const float* gamma, const float* beta, T* out, int ldo, int Ppad,
                          hipStream_t s) {
    DSH_REQUIRE(Ppad <= 64 * MAX_PER_LANE, "concat_ln_rows: concat width too large");
    hipLaunchKernelGGL(concat_ln_rows_kernel<T>, dim3(ceil_div(M, ROWS_PER_BLOCK)), dim3(256), 0, s, sg, M, gamma, beta,
                       out, ldo, Ppad);
    DSH_HIP_CHECK(hipGetLastError());
    return 0;
}
template int launch_concat_ln_rows<float>(const ConcatSegs&, int, const float*, const float*, float*, int, int, hipStream_t);
template int launch_concat_ln_rows<bf16>(const ConcatSegs&, int, const float*, const float*, bf16*, int, int, hipStream_t);

// ------------------------------------------------------------------------------------------------
// out[(b,t), tap*Cin + c] = x[b, t + tap - 1, c]  (zero outside the window: padding=1 per sample)
template <typename TI, typename T>
__global__ void im2col3_rows_kernel(const TI* x, int ldx, int B, int frames, int Cin, T* out, int ldo) {
    const int row = blockIdx.x;
    const int b = row / frames, t = row % frames;
    T* orow = out + (size_t)row * ldo;
    for (int tap = 0; tap < 3; ++tap) {
        const int ts = t + tap - 1;
        const bool ok = ts >= 0 && ts < frames;
        const TI* xr = x + ((size_t)b * frames + (ok ? ts : 0)) * ldx;
        for (int c = threadIdx.x; c < Cin; c += blockDim.x)
            orow[tap * Cin + c] = ok ? from_f32<T>(to_f32<TI>(xr[c])) : from_f32<T>(0.f);
    }
}

template <typename TI, typename T>
int launch_im2col3_rows(const TI* x, int ldx, int B, int frames, int Cin, T* out, int ldo, hipStream_t s) {
    hipLaunchKernelGGL((im2col3_rows_kernel<TI, T>), dim3(B * frames), dim3(256), 0, s, x, ldx, B, frames, Cin, out, ldo);
    DSH_HIP_CHECK(hipGetLastError());
    return 0;
}
template int launch_im2col3_rows<float, float>(const float*, int, int, int, int, float*, int, hipStream_t);
template int launch_im2col3_rows<float, bf16>(const float*, int, int, int, int, bf16*, int, hipStream_t);
template int launch_im2col3_rows<bf16, bf16>(const bf16*, int, int, int, int, bf16*, int, hipStream_t);

// ------------------------------------------------------------------------------------------------
// temb[b, j] = cos(t_b f_j) (j < half), sin(t_b f_{j-half}) (j >= half); f_j = exp(-ln(1e4) j / half)
template <typename T>
__global__ void temb_rows_kernel(const int64_t* t, int B, int dim, T* out, int ldo) {
    const int b = blockIdx.x;
    const int half = dim / 2;
    const float tv = (float)t[b];
    for (int j = threadIdx.x; j < half; j += blockDim.x) {
        // the reference builds freqs in fp32: exp(-log(10000) * arange(half) / half)
        const float f = expf(-9.210340371976184f * (float)j / (float)half);
        const float a = tv * f;
        out[(size_t)b * ldo + j] = from_f32<T>(cosf(a));
        out[(size_t)b * ldo + half + j] = from_f32<T>(sinf(a));
    }
}
template <typename T>
int launch_temb_rows(const int64_t* t, int B, int dim, T* out, int ldo, hipStream_t s) {
    hipLaunchKernelGGL(temb_rows_kernel<T>, dim3(B), dim3(256), 0, s, t, B, dim, out, ldo);
    DSH_HIP_CHECK(hipGetLastError());
    return 0;
}
template int launch_temb_rows<float>(const int64_t*, int, int, float*, int, hipStream_t);
template int launch_temb_rows<bf16>(const int64_t*, int, int, bf16*, int, hipStream_t);

// ------------------------------------------------------------------------------------------------
// out[r, c] = scale * x[r, c0 + c] for c < w, 0 for w <= c < wpad; optional fp32 copy (same scale)
template <typename T>
__global__ void pack_cols_kernel(const float* x, int ldx, int M, int c0, int w, int wpad, float scale, T* out, int ldo,
                                 float* outf, int ldof) {
    const int row = blockIdx.x;
    for (int c = threadIdx.x; c < wpad; c += blockDim.x) {
        const float v = (c < w) ? scale * x[(size_t)row * ldx + c0 + c] : 0.f;
        if (out) out[(size_t)row * ldo + c] = from_f32<T>(v);
        if (outf && c < w) outf[(size_t)row * ldof + c] = v;
    }
}
template <typename T>
int launch_pack_cols(const float* x, int ldx, int M, int c0, int w, int wpad, float scale, T* out, int ldo, float* outf,
                     int ldof, hipStream_t s) {
    hipLaunchKernelGGL(pack_cols_kernel<T>, dim3(M), dim3(128), 0, s, x, ldx, M, c0, w, wpad, scale, out, ldo, outf, ldof);
    DSH_HIP_CHECK(hipGetLastError());
    return 0;
}
template int launch_pack_cols<float>(const float*, int, int, int, int, int, float, float*, int, float*, int, hipStream_t);
template int launch_pack_cols<bf16>(const float*, int, int, int, int, int, float, bf16*, int, float*, int, hipStream_t);

// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// eps[b,t,c0+c] = o_u + s (o_c - o_u)   (o rows: [0,Mc) unconditional, [cond_row0, cond_row0+Mc) conditional; has_null==0: copy)
// x0[r,c] = c1[b] * x[b,t,c0+c] - c2[b] * eps   (optional; expression branch feeding the gesture concat)
__global__ void cfg_mix_kernel(const float* o, int ldo, int cond_row0, int frames, int w, int has_null, float cond_scale,
                               float* eps, int lde, int c0, const float* x, int ldx, const float* c1, const float* c2,
                               float* x0, int ldx0) {
    const int row = blockIdx.x;
    const int b = row / frames;
    for (int c = threadIdx.x; c < w; c += blockDim.x) {
        float e;
        if (has_null) {
            const float u = o[(size_t)row * ldo + c];
            const float k = o[(size_t)(row + cond_row0) * ldo + c];
            e = __fadd_rn(u, __fmul_rn(cond_scale, __fsub_rn(k, u)));
        } else {
            e = o[(size_t)row * ldo + c];
        }
        eps[(size_t)row * lde + c0 + c] = e;
        if (x0) {
            const float a = __fmul_rn(c1[b], x[(size_t)row * ldx + c0 + c]);
            const float bb = __fmul_rn(c2[b], e);
            x0[(size_t)row * ldx0 + c] = __fsub_rn(a, bb);
        }
    }
}
int launch_cfg_mix(const float* o, int ldo, int Mc, int cond_row0, int frames, int w, int has_null, float cond_scale,
                   float* eps, int lde, int c0, const float* x, int ldx, const float* c1, const float* c2, float* x0,
                   int ldx0, hipStream_t s) {
    hipLaunchKernelGGL(cfg_mix_kernel, dim3(Mc), dim3(128), 0, s, o, ldo, cond_row0, frames, w, has_null, cond_scale, eps, lde,
                       c0, x, ldx, c1, c2, x0, ldx0);
    DSH_HIP_CHECK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Tiled activation layouts of the token-per-lane Linears (tl_linear.hip):
//   bf16: tile (tb, kt) = 32 tokens x 16 features (1 KB, 32 B per token);  fp32: per (tb, nt) four lane-native 1 KB pieces,
//   float index (((tb*NT + nt)*4 + qi)*64 + lane)*4 + e  <->  n = 32nt + 16(qi>>1) + 8h + 4(qi&1) + e, lane = (t&31) + 32h
typedef float tf32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t tile_pack2(float lo, float hi) { return pack_bf16_pair(lo, hi); }
__device__ __forceinline__ float tile_ld(const float* p) { return *p; }
__device__ __forceinline__ float tile_ld(const bf16* p) { return to_f32<bf16>(*p); }

typedef uint32_t tu32x4 __attribute__((ext_vector_type(4)));
// W'[r] = W[tl_weight_src_row(r)]: the pi-permutation of weight rows inside every 32-row tile (tl_linear.hip), on device
__global__ void tl_permute_weight_kernel(const tu32x4* __restrict__ src, tu32x4* __restrict__ dst, int N, int chunks_per_row) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)N * chunks_per_row) return;
    const int r = (int)(i / chunks_per_row), c = (int)(i % chunks_per_row);
    const int rho = r & 31, q = rho >> 3, hh = (rho >> 2) & 1, e = rho & 3;
    const int sr = (r & ~31) + 16 * (q >> 1) + 8 * hh + 4 * (q & 1) + e;
    dst[i] = src[(size_t)sr * chunks_per_row + c];
}
__global__ void silu_f32_kernel(const float* x, float* y, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = silu_f(x[i]);
}
int launch_silu_f32(const float* x, float* y, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(silu_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, y, n);
    DSH_HIP_CHECK(hipGetLastError());
    return 0;
}
int launch_tl_permute_weight(const void* W, int N, int K, void* dst, hipStream_t s) {
    DSH_REQUIRE(N % 32 == 0 && K % 8 == 0, "tl_permute_weight: N must be a multiple of 32, K of 8");
    const size_t n = (size_t)N * (K / 8);
    hipLaunchKernelGGL(tl_permute_weight_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                       reinterpret_cast<const tu32x4*>(W), reinterpret_cast<tu32x4*>(dst), N, K / 8);
    DSH_HIP_CHECK(hipGetLastError());
    return 0;
}

template <typename TS>
__global__ void tile_rows_bf16_kernel(const TS* src, int ld, int M, int w, char* dst, int Wd, size_t nchunk) {
    const int KT = Wd >> 4;
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < nchunk; c += (size_t)gridDim.x * blockDim.x) {
        const size_t tile = c >> 6;
        const int j = (int)(c & 63), m = j >> 1, hh = j & 1;
        const size_t tb = tile / KT;
        const int kt = (int)(tile % KT);
        const size_t t = tb * 32 + m;
        const int n0 = kt * 16 + hh * 8;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (t < (size_t)M && n0 + i < w) ? tile_ld(src + t * ld + n0 + i) : 0.f;
        tu32x4 o;
        o.x = tile_pack2(v[0], v[1]); o.y = tile_pack2(v[2], v[3]); o.z = tile_pack2(v[4], v[5]); o.w = tile_pack2(v[6], v[7]);
        *reinterpret_cast<tu32x4*>(dst + c * 16) = o;
    }
}
template <typename TS>
int launch_tile_rows_bf16(const TS* src, int ld, int M, int w, void* dst, int Wd, hipStream_t s) {
    DSH_REQUIRE(Wd % 16 == 0 && w <= Wd && M > 0, "tile_rows: width must be a multiple of 16");
    const size_t nchunk = (size_t)ceil_div(M, 32) * (Wd >> 4) * 64;
    const int blocks = (int)std::min<size_t>((nchunk + 255) / 256, 16384);
    hipLaunchKernelGGL(tile_rows_bf16_kernel<TS>, dim3(blocks), dim3(256), 0, s, src, ld, M, w, reinterpret_cast<char*>(dst), Wd, nchunk);
    DSH_HIP_CHECK(hipGetLastError());
    return 0;
}
template int launch_tile_rows_bf16<float>(const float*, int, int, int, void*, int, hipStream_t);
template int launch_tile_rows_bf16<bf16>(const bf16*, int, int, int, void*, int, hipStream_t);

__global__ void untile_rows_bf16_kernel(const char* src, int Wd, int M, int w, uint16_t* dst, int ld, size_t nchunk) {
    const int KT = Wd >> 4;
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < nchunk; c += (size_t)gridDim.x * blockDim.x) {
        const size_t tile = c >> 6;
        const int j = (int)(c & 63), m = j >> 1, hh = j & 1;
        const size_t tb = tile / KT;
        const int kt = (int)(tile % KT);
        const size_t t = tb * 32 + m;
        const int n0 = kt * 16 + hh * 8;
        if (t >= (size_t)M) continue;
        const tu32x4 v = *reinterpret_cast<const tu32x4*>(src + c * 16);
        const uint32_t wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (n0 + i < w) dst[t * ld + n0 + i] = (uint16_t)((i & 1) ? (wv[i >> 1] >> 16) : (wv[i >> 1] & 0xffffu));
    }
}
int launch_untile_rows_bf16(const void* src, int Wd, int M, int w, void* dst, int ld, hipStream_t s) {
    DSH_REQUIRE(Wd % 16 == 0 && w <= Wd && M > 0, "untile_rows: width must be a multiple of 16");
    const size_t nchunk = (size_t)ceil_div(M, 32) * (Wd >> 4) * 64;
    const int blocks = (int)std::min<size_t>((nchunk + 255) / 256, 16384);
    hipLaunchKernelGGL(untile_rows_bf16_kernel, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const char*>(src), Wd, M, w,
                       reinterpret_cast<uint16_t*>(dst), ld, nchunk);
    DSH_HIP_CHECK(hipGetLastError());
    return 0;
}

template <bool TO_TILED>
__global__ void tile_rows_f32_kernel(const float* src, float* dst, int ld, int M, int Wd, size_t npiece) {
    const int NT = Wd >> 5;
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < npiece; c += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(c & 63), qi = (int)((c >> 6) & 3);
        const size_t blk = c >> 8, tb = blk / NT;
        const int nt = (int)(blk % NT);
        const size_t t = tb * 32 + (lane & 31);
        const int n = 32 * nt + 16 * (qi >> 1) + 8 * (lane >> 5) + 4 * (qi & 1);
        if (TO_TILED) {
            tf32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (t < (size_t)M) v = *reinterpret_cast<const tf32x4*>(src + t * ld + n);
            *reinterpret_cast<tf32x4*>(dst + c * 4) = v;
        } else if (t < (size_t)M) {
            *reinterpret_cast<tf32x4*>(dst + t * ld + n) = *reinterpret_cast<const tf32x4*>(src + c * 4);
        }
    }
}
int launch_tile_rows_f32(const float* src, int ld, int M, float* dst, int Wd, hipStream_t s) {
    DSH_REQUIRE(Wd % 32 == 0 && ld % 4 == 0 && M > 0, "tile_rows_f32: width must be a multiple of 32");
    const size_t npiece = (size_t)ceil_div(M, 32) * (Wd >> 5) * 256;
    const int blocks = (int)std::min<size_t>((npiece + 255) / 256, 16384);
    hipLaunchKernelGGL(tile_rows_f32_kernel<true>, dim3(blocks), dim3(256), 0, s, src, dst, ld, M, Wd, npiece);
    DSH_HIP_CHECK(hipGetLastError());
    return 0;
}
int launch_untile_rows_f32(const float* src, int Wd, int M, float* dst, int ld, hipStream_t s) {
    DSH_REQUIRE(Wd % 32 == 0 && ld % 4 == 0 && M > 0, "untile_rows_f32: width must be a multiple of 32");
    const size_t npiece = (size_t)ceil_div(M, 32) * (Wd >> 5) * 256;
    const int blocks = (int)std::min<size_t>((npiece + 255) / 256, 16384);
    hipLaunchKernelGGL(tile_rows_f32_kernel<false>, dim3(blocks), dim3(256), 0, s, src, dst, ld, M, Wd, npiece);
    DSH_HIP_CHECK(hipGetLastError());
    return 0;
}

// StylizationBlock coefficient fold (models/transformer.py:86-97): the stacked FiLM table rows [scale(D) | shift(D)] per
// block become [A | B] with A = gamma (1 + scale), B = beta (1 + scale) + shift, so that the token-per-lane prologue
// computes SiLU(((x - mean) rstd) A + B) with two coefficient vectors instead of four.  In place; gamma/beta: [nblk, D].
__global__ void film_fold_kernel(float* tab, int ld, int B, int nblk, int D, const float* gamma, const float* beta) {
    const size_t n = (size_t)B * nblk * D;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % D), j = (int)((i / D) % nblk);
        const size_t b = i / ((size_t)D * nblk);
        float* r = tab + b * ld + (size_t)j * 2 * D;
        const float sc = 1.0f + r[k], sh = r[D + k];
        r[k] = gamma[j * D + k] * sc;
        r[D + k] = fmaf(beta[j * D + k], sc, sh);
    }
}
int launch_film_fold(float* tab, int ld, int B, int nblk, int D, const float* gamma, const float* beta, hipStream_t s) {
    const size_t n = (size_t)B * nblk * D;
    const int blocks = (int)std::min<size_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(film_fold_kernel, dim3(blocks), dim3(256), 0, s, tab, ld, B, nblk, D, gamma, beta);
    DSH_HIP_CHECK(hipGetLastError());
    return 0;
}

// Round 6: FiLM rows are computed for the DISTINCT (timestep, speaker) rows only and expanded per clip here: dst[b] = fold(src[idx[b]])
// (idx == null: identity; fold == 0: plain copy — the fp32 path folds nothing).  One float4 per thread and piece.
__global__ void film_expand_kernel(const float* src, int ld, const int* idx, float* dst, int B, int nblk, int D, const float* gamma,
                                   const float* beta, int fold) {
    const int q4 = nblk * D / 4;                                 // float4 pieces per (row, half): [scale | shift] per block
    const size_t n = (size_t)B * q4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / q4;
        const int r = (int)(i % q4), j = r / (D / 4), k = (r % (D / 4)) * 4;
        const float* sr = src + (size_t)(idx ? idx[b] : (int)b) * ld + (size_t)j * 2 * D;
        float* dr = dst + b * ld + (size_t)j * 2 * D;
        tf32x4 sc = *reinterpret_cast<const tf32x4*>(sr + k), sh = *reinterpret_cast<const tf32x4*>(sr + D + k);
        if (fold) {
            const tf32x4 g = *reinterpret_cast<const tf32x4*>(gamma + j * D + k), be = *reinterpret_cast<const tf32x4*>(beta + j * D + k);
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float s1 = 1.0f + sc[e]; sh[e] = fmaf(be[e], s1, sh[e]); sc[e] = g[e] * s1; }
        }
        *reinterpret_cast<tf32x4*>(dr + k) = sc;
        *reinterpret_cast<tf32x4*>(dr + D + k) = sh;
    }
}
int launch_film_expand(const float* src, int ld, const int* idx, float* dst, int B, int nblk, int D, const float* gamma, const float* beta,
                       int fold, hipStream_t s) {
    DSH_REQUIRE(D % 4 == 0 && ld % 4 == 0, "film_expand: widths must be multiples of 4");
    const size_t n = (size_t)B * (nblk * D / 4);
    const int blocks = (int)std::min<size_t>((n + 255) / 256, 8192);
    hipLaunchKernelGGL(film_expand_kernel, dim3(blocks), dim3(256), 0, s, src, ld, idx, dst, B, nblk, D, gamma, beta, fold);
    DSH_HIP_CHECK(hipGetLastError());
    return 0;
}
// dst[b, :w] = src[idx[b], :w]  (fp32 rows; the speaker embedding per clip from the distinct speakers' rows)
__global__ void gather_rows_f32_kernel(const float* src, int ld, const int* idx, float* dst, int ldd, int B, int w) {
    const size_t n = (size_t)B * w;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / w; const int k = (int)(i % w);
        dst[b * ldd + k] = src[(size_t)idx[b] * ld + k];
    }
}
int launch_gather_rows_f32(const float* src, int ld, const int* idx, float* dst, int ldd, int B, int w, hipStream_t s) {
    const size_t n = (size_t)B * w;
    hipLaunchKernelGGL(gather_rows_f32_kernel, dim3((int)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, s, src, ld, idx, dst, ldd, B, w);
    DSH_HIP_CHECK(hipGetLastError());
    return 0;
}

// one thread = 8 consecutive features of one token: two fp32 pieces (qi = 2c, 2c+1) + one 16-byte bf16 chunk, per CFG half
__global__ void seed_stream_kernel(const float* h0, int Mc, int D, const float* cadd, int has_null, int row1, float* h,
                                   char* h16, char* hlo, size_t nitem) {
    const int NT = D >> 5;
    for (size_t it = (size_t)blockIdx.x * blockDim.x + threadIdx.x; it < nitem; it += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(it & 63), c = (int)((it >> 6) & 1);
        const size_t blk = it >> 7, tb = blk / NT;
        const int nt = (int)(blk % NT);
        const size_t t = tb * 32 + (lane & 31);
        if (t >= (size_t)Mc) continue;
        const int hh = lane >> 5, n = 32 * nt + 16 * c + 8 * hh;
        const tf32x4 a = *reinterpret_cast<const tf32x4*>(h0 + t * D + n), b = *reinterpret_cast<const tf32x4*>(h0 + t * D + n + 4);
        const int lane_off = (lane & 31) * 32 + hh * 16;
        for (int half = 0; half < 1 + has_null; ++half) {
            tf32x4 va = a, vb = b;
            if (half == 0 && has_null) {
                va += *reinterpret_cast<const tf32x4*>(cadd + n);
                vb += *reinterpret_cast<const tf32x4*>(cadd + n + 4);
            }
            const size_t tbh = tb + (half ? (size_t)(row1 >> 5) : 0);
            tu32x4 o;
            o.x = tile_pack2(va.x, va.y); o.y = tile_pack2(va.z, va.w); o.z = tile_pack2(vb.x, vb.y); o.w = tile_pack2(vb.z, vb.w);
            *reinterpret_cast<tu32x4*>(h16 + (tbh * (2 * NT) + 2 * nt + c) * 1024 + lane_off) = o;
            if (hlo) {       // residual stream as (hi, lo) bf16 planes: lo = bf16(h - hi)
                const float v[8] = {va.x, va.y, va.z, va.w, vb.x, vb.y, vb.z, vb.w};
                const uint32_t hw[4] = {o.x, o.y, o.z, o.w};
                tu32x4 l;
                uint32_t lw[4];
                for (int j = 0; j < 4; ++j)
                    lw[j] = tile_pack2(v[2 * j] - __builtin_bit_cast(float, hw[j] << 16), v[2 * j + 1] - __builtin_bit_cast(float, hw[j] & 0xffff0000u));
                l.x = lw[0]; l.y = lw[1]; l.z = lw[2]; l.w = lw[3];
                *reinterpret_cast<tu32x4*>(hlo + (tbh * (2 * NT) + 2 * nt + c) * 1024 + lane_off) = l;
            } else {
                float* hp = h + (((tbh * NT + nt) * 4 + 2 * c) * 64 + lane) * 4;
                *reinterpret_cast<tf32x4*>(hp) = va;
                *reinterpret_cast<tf32x4*>(hp + 256) = vb;
            }
        }
    }
}
int launch_seed_stream(const float* h0, int Mc, int D, const float* c, int has_null, int row1, float* h, void* h16, hipStream_t s, void* hlo) {
    DSH_REQUIRE(D % 32 == 0 && (!has_null || (row1 % 32 == 0 && row1 >= Mc && c)), "seed_stream: bad arguments");
    const size_t nitem = (size_t)ceil_div(Mc, 32) * (D >> 5) * 128;
    const int blocks = (int)std::min<size_t>((nitem + 255) / 256, 16384);
    hipLaunchKernelGGL(seed_stream_kernel, dim3(blocks), dim3(256), 0, s, h0, Mc, D, c, has_null, row1, h, reinterpret_cast<char*>(h16),
                       reinterpret_cast<char*>(hlo), nitem);
    DSH_HIP_CHECK(hipGetLastError());
    return 0;
}

// row-major fp32 <-> (hi, lo) bf16 planes, tiled (one thread = 8 consecutive features of one token = one 16-byte chunk per plane)
template <bool TO_PLANES>
__global__ void hilo_rows_kernel(float* rm, int ld, int M, int w, char* hi, char* lo, int Wd, size_t nchunk) {
    const int KT = Wd >> 4;
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < nchunk; c += (size_t)gridDim.x * blockDim.x) {
        const size_t tile = c >> 6;
        const int j = (int)(c & 63), m = j >> 1, hh = j & 1;
        const size_t tb = tile / KT;
        const int kt = (int)(tile % KT);
        const size_t t = tb * 32 + m;
        const int n0 = kt * 16 + hh * 8;
        if (TO_PLANES) {
            float v[8];
            for (int i = 0; i < 8; ++i) v[i] = (t < (size_t)M && n0 + i < w) ? rm[t * ld + n0 + i] : 0.f;
            uint32_t hw[4], lw[4];
            for (int i = 0; i < 4; ++i) {
                hw[i] = tile_pack2(v[2 * i], v[2 * i + 1]);
                lw[i] = tile_pack2(v[2 * i] - __builtin_bit_cast(float, hw[i] << 16), v[2 * i + 1] - __builtin_bit_cast(float, hw[i] & 0xffff0000u));
            }
            tu32x4 a, b;
            a.x = hw[0]; a.y = hw[1]; a.z = hw[2]; a.w = hw[3]; b.x = lw[0]; b.y = lw[1]; b.z = lw[2]; b.w = lw[3];
            *reinterpret_cast<tu32x4*>(hi + c * 16) = a;
            *reinterpret_cast<tu32x4*>(lo + c * 16) = b;
        } else if (t < (size_t)M) {
            const tu32x4 a = *reinterpret_cast<const tu32x4*>(hi + c * 16), b = *reinterpret_cast<const tu32x4*>(lo + c * 16);
            const uint32_t hw[4] = {a.x, a.y, a.z, a.w}, lw[4] = {b.x, b.y, b.z, b.w};
            for (int i = 0; i < 8; ++i) {
                if (n0 + i >= w) continue;
                const uint32_t h2 = hw[i >> 1], l2 = lw[i >> 1];
                const float fh = __builtin_bit_cast(float, (i & 1) ? (h2 & 0xffff0000u) : (h2 << 16));
                const float fl = __builtin_bit_cast(float, (i & 1) ? (l2 & 0xffff0000u) : (l2 << 16));
                rm[t * ld + n0 + i] = fh + fl;
            }
        }
    }
}
int launch_tile_rows_hilo(const float* src, int ld, int M, int w, void* hi, void* lo, int Wd, hipStream_t s) {
    DSH_REQUIRE(Wd % 16 == 0 && w <= Wd && M > 0, "tile_rows_hilo: width must be a multiple of 16");
    const size_t nchunk = (size_t)ceil_div(M, 32) * (Wd >> 4) * 64;
    const int blocks = (int)std::min<size_t>((nchunk + 255) / 256, 16384);
    hipLaunchKernelGGL(hilo_rows_kernel<true>, dim3(blocks), dim3(256), 0, s, const_cast<float*>(src), ld, M, w, reinterpret_cast<char*>(hi),
                       reinterpret_cast<char*>(lo), Wd, nchunk);
    DSH_HIP_CHECK(hipGetLastError());
    return 0;
}
int launch_untile_rows_hilo(const void* hi, const void* lo, int Wd, int M, int w, float* dst, int ld, hipStream_t s) {
    DSH_REQUIRE(Wd % 16 == 0 && w <= Wd && M > 0, "untile_rows_hilo: width must be a multiple of 16");
    const size_t nchunk = (size_t)ceil_div(M, 32) * (Wd >> 4) * 64;
    const int blocks = (int)std::min<size_t>((nchunk + 255) / 256, 16384);
    hipLaunchKernelGGL(hilo_rows_kernel<false>, dim3(blocks), dim3(256), 0, s, dst, ld, M, w, reinterpret_cast<char*>(const_cast<void*>(hi)),
                       reinterpret_cast<char*>(const_cast<void*>(lo)), Wd, nchunk);
    DSH_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace dsh

// ------------------------------------------------------------------------------------------------
// Rows either side of the sampling path (SURVEY.md §8f "next"):
//   interp_time   F.interpolate(x^T, size=T, mode='linear', align_corners=True)^T along frames
//                 (datasets/show.py:98, trainers/ddpm_show_trainer.py:1082: HuBERT features -> pose frame rate)
//   affine_cols   inv_standardize: y = x * std[c] + mean[c]          (datasets/show.py:157-162)
namespace dsh {

__global__ void interp_time_kernel(const float* x, int B, int Tin, int C, float* y, int Tout) {
    const int row = blockIdx.x;                 // b * Tout + t
    const int b = row / Tout, t = row % Tout;
    // align_corners=True: src = t * (Tin - 1) / (Tout - 1); Tout == 1 samples frame 0 (ATen area_pixel_compute_scale)
    const float scale = Tout > 1 ? (float)(Tin - 1) / (float)(Tout - 1) : 0.0f;
    const float src = scale * (float)t;
    int i0 = (int)src;
    if (i0 > Tin - 1) i0 = Tin - 1;
    const int i1 = i0 + (i0 < Tin - 1 ? 1 : 0);
    const float w1 = src - (float)i0, w0 = 1.0f - w1;
    const float* x0 = x + ((size_t)b * Tin + i0) * C;
    const float* x1 = x + ((size_t)b * Tin + i1) * C;
    float* yr = y + (size_t)row * C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) yr[c] = w0 * x0[c] + w1 * x1[c];
}

int launch_interp_time(const float* x, int B, int Tin, int C, float* y, int Tout, hipStream_t s) {
    DSH_REQUIRE(B > 0 && Tin > 0 && Tout > 0 && C > 0, "interp_time: dims must be positive");
    hipLaunchKernelGGL(interp_time_kernel, dim3(B * Tout), dim3(256), 0, s, x, B, Tin, C, y, Tout);
    DSH_HIP_CHECK(hipGetLastError());
    return 0;
}

__global__ void affine_cols_kernel(const float* x, size_t n, int C, const float* mean, const float* stdv, float* y) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % (size_t)C);
        y[i] = __fadd_rn(__fmul_rn(x[i], stdv[c]), mean[c]);
    }
}

int launch_affine_cols(const float* x, size_t n, int C, const float* mean, const float* stdv, float* y, hipStream_t s) {
    const size_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(affine_cols_kernel, dim3((unsigned)(blocks < 2048 ? (blocks ? blocks : 1) : 2048)), dim3(256), 0, s, x, n, C,
                       mean, stdv, y);
    DSH_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace dsh
