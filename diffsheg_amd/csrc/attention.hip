// Linear ("efficient") temporal self-attention core of DiffSHEG
// (/root/reference/models/transformer.py:122-128):
//
//   q = softmax over the head's channels of Q[b,t,h,:]
//   k = softmax over the T frames       of K[b,:,h,d]        (src_mask == 1 at inference)
//   A[b,h] = k^T v   (hd x hd),    y[b,t,h,:] = q[b,t,h,:] A[b,h]
//
// There is no T x T score matrix; the work per (sample, head) is two tiny (hd x T x hd) products.
// v1 mapping for gfx950: one 64-lane wavefront per 64 consecutive channels of one sample (one head
// at hd=64, four heads at hd=16).  Lane l owns channel c: it computes the time-softmax of K[:,c]
// lane-locally (no cross-lane traffic), keeps column c of A (hd registers), and the per-frame
// k / q rows are broadcast through a 64-float LDS row.  Softmax over channels is a wavefront
// (hd=64) or 16-lane (hd=16) butterfly.  All math is fp32; storage type T is fp32 or bf16.
#include <stdlib.h>

#include "dsh_common.h"
#include "dsh_kernels.h"

namespace dsh {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int HD>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
    for (int o = HD / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
template <int HD>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = HD / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <typename T, int HD>
__global__ __launch_bounds__(64) void linear_attention_kernel(const T* __restrict__ qkv, int ldq, int frames, int D,
                                                              T* __restrict__ y, int ldy) {
    __shared__ float bc[2][64];
    const int lane = threadIdx.x;
    const int b = blockIdx.y;
    const int c = blockIdx.x * 64 + lane;
    const int g0 = (lane / HD) * HD;             // first lane of this lane's head
    const T* base = qkv + (size_t)b * frames * ldq;

    // ---- pass 1: online max / sum of K[:, c] over time ------------------------------------------
    float m = -INFINITY, ssum = 0.f;
    for (int t = 0; t < frames; ++t) {
        const float kv = to_f32<T>(base[(size_t)t * ldq + D + c]);
        const float mn = fmaxf(m, kv);
        ssum = ssum * expf(m - mn) + expf(kv - mn);
        m = mn;
    }
    const float inv = 1.0f / ssum;

    // ---- pass 2: A[d][c] = sum_t khat[t][d] * v[t][c] -------------------------------------------
    float A[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) A[d] = 0.f;
    for (int t = 0; t < frames; ++t) {
        const float kh = expf(to_f32<T>(base[(size_t)t * ldq + D + c]) - m) * inv;
        const float v = to_f32<T>(base[(size_t)t * ldq + 2 * D + c]);
        bc[t & 1][lane] = kh;
        __syncthreads();
        const float* row = &bc[t & 1][g0];
#pragma unroll
        for (int d = 0; d < HD; ++d) A[d] = fmaf(row[d], v, A[d]);
    }
    __syncthreads();

    // ---- pass 3: y[t][c] = sum_d qhat[t][d] * A[d][c] -------------------------------------------
    for (int t = 0; t < frames; ++t) {
        const float q = to_f32<T>(base[(size_t)t * ldq + c]);
        const float mx = group_max<HD>(q);
        const float e = expf(q - mx);
        const float sm = group_sum<HD>(e);
        bc[t & 1][lane] = e / sm;
        __syncthreads();
        const float* row = &bc[t & 1][g0];
        float acc = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) acc = fmaf(row[d], A[d], acc);
        y[((size_t)b * frames + t) * ldy + c] = from_f32<T>(acc);
    }
}


// Cross-attention form (LinearTemporalCrossAttention, models/transformer.py:146-166): queries come from the motion tokens
// (T frames), keys / values from the conditioning sequence xf (N frames, softmax of K over those N).  q [B,T,D] with row stride
// ldq; kv [B,N,2D] = (k | v) with row stride ldkv.  fp32, one wave per 64 channels, HD channels per head.
template <int HD>
__global__ __launch_bounds__(64) void linear_cross_attention_kernel(const float* __restrict__ q, int ldq, int frames,
                                                                    const float* __restrict__ kv, int ldkv, int frames_kv, int D,
                                                                    float* __restrict__ y, int ldy) {
    __shared__ float bc[2][64];
    const int lane = threadIdx.x;
    const int b = blockIdx.y;
    const int c = blockIdx.x * 64 + lane;
    const int g0 = (lane / HD) * HD;
    const float* kb = kv + (size_t)b * frames_kv * ldkv;
    float m = -INFINITY, ssum = 0.f;
    for (int t = 0; t < frames_kv; ++t) {
        const float k = kb[(size_t)t * ldkv + c];
        const float mn = fmaxf(m, k);
        ssum = ssum * expf(m - mn) + expf(k - mn);
        m = mn;
    }
    const float inv = 1.0f / ssum;
    float A[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) A[d] = 0.f;
    for (int t = 0; t < frames_kv; ++t) {
        const float kh = expf(kb[(size_t)t * ldkv + c] - m) * inv;
        const float v = kb[(size_t)t * ldkv + D + c];
        bc[t & 1][lane] = kh;
        __syncthreads();
        const float* row = &bc[t & 1][g0];
#pragma unroll
        for (int d = 0; d < HD; ++d) A[d] = fmaf(row[d], v, A[d]);
    }
    __syncthreads();
    const float* qb = q + (size_t)b * frames * ldq;
    for (int t = 0; t < frames; ++t) {
        const float qq = qb[(size_t)t * ldq + c];
        const float mx = group_max<HD>(qq);
        const float e = expf(qq - mx);
        const float sm = group_sum<HD>(e);
        bc[t & 1][lane] = e / sm;
        __syncthreads();
        const float* row = &bc[t & 1][g0];
        float acc = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) acc = fmaf(row[d], A[d], acc);
        y[((size_t)b * frames + t) * ldy + c] = acc;
    }
}

int launch_linear_cross_attention(const float* q, int ldq, int nbatch, int frames, const float* kv, int ldkv, int frames_kv, int D,
                                  int head_dim, float* y, int ldy, hipStream_t s) {
    DSH_REQUIRE(D % 64 == 0 && (head_dim == 64 || head_dim == 16 || head_dim == 32), "linear_cross_attention: D % 64, head_dim in {16, 32, 64}");
    DSH_REQUIRE(frames > 0 && frames_kv > 0, "linear_cross_attention: empty sequence");
    dim3 grid(D / 64, nbatch);
    if (head_dim == 64) hipLaunchKernelGGL((linear_cross_attention_kernel<64>), grid, dim3(64), 0, s, q, ldq, frames, kv, ldkv, frames_kv, D, y, ldy);
    else if (head_dim == 32) hipLaunchKernelGGL((linear_cross_attention_kernel<32>), grid, dim3(64), 0, s, q, ldq, frames, kv, ldkv, frames_kv, D, y, ldy);
    else hipLaunchKernelGGL((linear_cross_attention_kernel<16>), grid, dim3(64), 0, s, q, ldq, frames, kv, ldkv, frames_kv, D, y, ldy);
    DSH_HIP_CHECK(hipGetLastError());
    return 0;
}

// Same algorithm for windows of <= 96 frames with every K / V / Q column load issued up front.  The loop form above exposes one
// dependent global-load round trip per frame and pass; at chain batch sizes (two blocks on the whole chip for encoder_aud) that
// was 111 us per launch — 5 % of a batch-1 evaluation.
// (TM = 48 for the 34-frame windows of the fp32 parity configuration: 160 instead of 256 column registers)
template <typename T, int HD, int TM = 96>
__global__ __launch_bounds__(64) void linear_attention_pre_kernel(const T* __restrict__ qkv, int ldq, int frames, int D,
                                                                  T* __restrict__ y, int ldy) {
    __shared__ float bc[2][64];
    const int lane = threadIdx.x;
    const int b = blockIdx.y;
    const int c = blockIdx.x * 64 + lane;
    const int g0 = (lane / HD) * HD;
    const T* base = qkv + (size_t)b * frames * ldq;
    float kr[TM], vr[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        const size_t ro = (size_t)(t < frames ? t : frames - 1) * ldq;
        kr[t] = to_f32<T>(base[ro + D + c]);
        vr[t] = to_f32<T>(base[ro + 2 * D + c]);
    }
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < TM; ++t) m = fmaxf(m, t < frames ? kr[t] : -INFINITY);
    float ssum = 0.f;
#pragma unroll
    for (int t = 0; t < TM; ++t) { kr[t] = t < frames ? expf(kr[t] - m) : 0.f; ssum += kr[t]; }
    const float inv = 1.0f / ssum;
    float A[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) A[d] = 0.f;
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        if (t < frames) {                                   // uniform
            bc[t & 1][lane] = kr[t] * inv;
            __syncthreads();
            const float* row = &bc[t & 1][g0];
#pragma unroll
            for (int d = 0; d < HD; ++d) A[d] = fmaf(row[d], vr[t], A[d]);
        }
    }
    __syncthreads();
    // q columns into the registers K occupied
#pragma unroll
    for (int t = 0; t < TM; ++t) kr[t] = to_f32<T>(base[(size_t)(t < frames ? t : frames - 1) * ldq + c]);
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        if (t < frames) {
            const float mx = group_max<HD>(kr[t]);
            const float e = expf(kr[t] - mx);
            const float sm = group_sum<HD>(e);
            bc[t & 1][lane] = e / sm;
            __syncthreads();
            const float* row = &bc[t & 1][g0];
            float acc = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) acc = fmaf(row[d], A[d], acc);
            y[((size_t)b * frames + t) * ldy + c] = from_f32<T>(acc);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// fp32 path, 64-channel heads, T <= TM frames (round 6): the two products on the exact-fp32 matrix pipe (v_mfma_f32_32x32x2_f32) instead
// of 2 x T x 64 VALU FMAs per lane behind LDS broadcasts (linear_attention_pre_kernel: 51 us per launch at the config-2 batch, 7 % of its
// step).  One wave per (sample, head), NO LDS: every operand layout is produced by the load pattern.
//   * A = k^T v (64 x 64, reduction over time).  The MFMA takes A_op[i][k] from lane (k = l >> 5, i = l & 31) and B_op[k][j] from lane
//     (k, j): lane (hh, i) loads K[2 s + hh][i], K[2 s + hh][32 + i] (and V likewise) for s = 0 .. T / 2 — two frames per k step, the even ones
//     in the lower half-wave.  The time-softmax of a channel is then lane-local over s plus ONE exchange with lane ^ 32.  2 x 2 output tiles,
//     four independent accumulators, T / 2 steps each.
//   * y^T = A^T q^T (64 x T, reduction over the head's channels d).  The accumulator of the first product already IS the A operand of the
//     second: lane (hh, c) holds A[d = 4 hh + (r & 3) + 8 (r >> 2) (+ 32 dt)][c] in register r, i.e. register r of the two half-waves is the
//     pair (k = 0, k = 1) of one k step.  The q rows are loaded "mirrored": lane (hh, t) holds q[t][d] for exactly those d, 16-byte pieces
//     of its own frame's row — so the channel-softmax of a frame is 32 lane-local values plus one exchange with lane ^ 32, and its result
//     is the B operand as it stands.  Output tile register r' of lane (hh, t) is y[t][4 hh + (r' & 3) + 8 (r' >> 2) (+ 32 ct)]: four
//     consecutive channels per 16-byte store.
// Same math as the VALU kernels (exact fp32 products, fp32 accumulation), another summation order.
__device__ __forceinline__ float at_silu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896340736f)); }   // (as gemm_f32_pro.hip)

template <int TM>
__global__ __launch_bounds__(64) void linear_attention_f32_mfma_kernel(const float* __restrict__ qkv, int ldq, int T, int D, float* __restrict__ y, int ldy) {
    constexpr int NS = TM / 2, NTT = (TM + 31) / 32;
    const int lane = threadIdx.x, i = lane & 31, hh = lane >> 5;
    const int b = blockIdx.y, head = blockIdx.x;
    const float* base = qkv + (size_t)b * T * ldq + head * 64;
    // ---- K, V: frame 2 s + hh, channels i and 32 + i -----------------------------------------------------------------------------
    float kr[NS][2], vr[NS][2];
#pragma unroll
    for (int sx = 0; sx < NS; ++sx) {
        const int t = 2 * sx + hh, tc = t < T ? t : T - 1;
        const float* r = base + (size_t)tc * ldq;
        kr[sx][0] = r[D + i]; kr[sx][1] = r[D + 32 + i];
        vr[sx][0] = r[2 * D + i]; vr[sx][1] = r[2 * D + 32 + i];
    }
    // ---- q rows, mirrored to the accumulator layout: lane (hh, t) holds q[t][32 dt + 8 q' + 4 hh + e] ---------------------------
    f32x4 qv[NTT][2][4];
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
        const int t = 32 * tt + i, tc = t < T ? t : T - 1;
        const float* r = base + (size_t)tc * ldq + 4 * hh;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) qv[tt][dt][qq] = *reinterpret_cast<const f32x4*>(r + 32 * dt + 8 * qq);
    }
    // ---- time-softmax of K per channel: lane-local over s, then the other parity of frames in lane ^ 32 ----------------------------
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        float m = -INFINITY;
#pragma unroll
        for (int sx = 0; sx < NS; ++sx) { if (2 * sx + hh >= T) kr[sx][c] = -INFINITY; m = fmaxf(m, kr[sx][c]); }
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int sx = 0; sx < NS; ++sx) { kr[sx][c] = expf(kr[sx][c] - m); sum += kr[sx][c]; }      // exp(-inf) = 0 on masked frames
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int sx = 0; sx < NS; ++sx) kr[sx][c] *= inv;
    }
    // ---- A[d][c] = sum_t k^[t][d] v[t][c] ---------------------------------------------------------------------------------------
    f32x16 aA[2][2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) aA[dt][ct][r] = 0.f;
#pragma unroll
    for (int sx = 0; sx < NS; ++sx) {
        if (2 * sx < T) {                                       // (uniform)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) aA[dt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(kr[sx][dt], vr[sx][ct], aA[dt][ct], 0, 0, 0);
        }
    }
    // ---- channel-softmax of q per frame, y^T = A^T q^^T, store ---------------------------------------------------------------------
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
        if (32 * tt < T) {                                      // (uniform)
            float m = -INFINITY;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) m = fmaxf(fmaxf(fmaxf(m, qv[tt][dt][qq].x), fmaxf(qv[tt][dt][qq].y, qv[tt][dt][qq].z)), qv[tt][dt][qq].w);
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            float sum = 0.f;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    f32x4 e4;
                    e4.x = expf(qv[tt][dt][qq].x - m); e4.y = expf(qv[tt][dt][qq].y - m); e4.z = expf(qv[tt][dt][qq].z - m); e4.w = expf(qv[tt][dt][qq].w - m);
                    sum += (e4.x + e4.y) + (e4.z + e4.w);
                    qv[tt][dt][qq] = e4;
                }
            sum += __shfl_xor(sum, 32, 64);
            const float inv = 1.0f / sum;
            f32x16 yo[2];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) yo[ct][r] = 0.f;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const f32x4 q4 = qv[tt][dt][qq];
                    const float qe[4] = {q4.x * inv, q4.y * inv, q4.z * inv, q4.w * inv};
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct) {
                            const float av = aA[dt][ct][4 * qq + e];        // (scalar copy of the vector element: hipcc pitfall 1)
                            yo[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, qe[e], yo[ct], 0, 0, 0);
                        }
                }
            const int t = 32 * tt + i;
            if (t < T) {
                float* yr = y + ((size_t)b * T + t) * ldy + head * 64 + 4 * hh;
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) {
                        f32x4 o; o.x = yo[ct][4 * qq]; o.y = yo[ct][4 * qq + 1]; o.z = yo[ct][4 * qq + 2]; o.w = yo[ct][4 * qq + 3];
                        *reinterpret_cast<f32x4*>(yr + 32 * ct + 8 * qq) = o;
                    }
            }
        }
    }
}

// The same with the FRONT OF THE StylizationBlock behind it (models/transformer.py:86-97: LayerNorm -> FiLM -> SiLU of the attention output, the
// input of sa_block.proj_out's Linear): a block is the EIGHT heads of one sample (D = 512), so it owns whole rows of y — the row moments are
// exchanged through 6 KB of LDS (two-pass: mean, then sum (y - mean)^2), and the transform is applied to the output registers ONCE per
// element, in a launch whose matrix pipe has slack, instead of by each of the eight N tiles of the Linear's row block (gemm_f32_pro.hip, PRO 2:
// matrix pipe busy 0.40).  The Linear then runs front-less.  film: per-sample rows [scale'(D) | shift'(D)] with the LayerNorm affine folded in.
template <int TM>
__global__ __launch_bounds__(512) void linear_attention_f32_mfma_sty_kernel(const float* __restrict__ qkv, int ldq, int T, int D, float* __restrict__ y, int ldy,
                                                                            const float* __restrict__ film, int film_ld, int film_off, int bmod) {
    constexpr int NS = TM / 2, NTT = (TM + 31) / 32;
    __shared__ float red[2][NTT * 32][8];
    const int lane = threadIdx.x & 63, i = lane & 31, hh = lane >> 5;
    const int head = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x;
    const float* base = qkv + (size_t)b * T * ldq + head * 64;
    float kr[NS][2], vr[NS][2];
#pragma unroll
    for (int sx = 0; sx < NS; ++sx) {
        const int t = 2 * sx + hh, tc = t < T ? t : T - 1;
        const float* r = base + (size_t)tc * ldq;
        kr[sx][0] = r[D + i]; kr[sx][1] = r[D + 32 + i];
        vr[sx][0] = r[2 * D + i]; vr[sx][1] = r[2 * D + 32 + i];
    }
    f32x4 qv[NTT][2][4];
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
        const int t = 32 * tt + i, tc = t < T ? t : T - 1;
        const float* r = base + (size_t)tc * ldq + 4 * hh;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) qv[tt][dt][qq] = *reinterpret_cast<const f32x4*>(r + 32 * dt + 8 * qq);
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        float m = -INFINITY;
#pragma unroll
        for (int sx = 0; sx < NS; ++sx) { if (2 * sx + hh >= T) kr[sx][c] = -INFINITY; m = fmaxf(m, kr[sx][c]); }
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int sx = 0; sx < NS; ++sx) { kr[sx][c] = expf(kr[sx][c] - m); sum += kr[sx][c]; }
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int sx = 0; sx < NS; ++sx) kr[sx][c] *= inv;
    }
    f32x16 aA[2][2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) aA[dt][ct][r] = 0.f;
#pragma unroll
    for (int sx = 0; sx < NS; ++sx) {
        if (2 * sx < T) {
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) aA[dt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(kr[sx][dt], vr[sx][ct], aA[dt][ct], 0, 0, 0);
        }
    }
    f32x16 yo[NTT][2];
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) yo[tt][ct][r] = 0.f;
        if (32 * tt < T) {
            float m = -INFINITY;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) m = fmaxf(fmaxf(fmaxf(m, qv[tt][dt][qq].x), fmaxf(qv[tt][dt][qq].y, qv[tt][dt][qq].z)), qv[tt][dt][qq].w);
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            float sum = 0.f;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    f32x4 e4;
                    e4.x = expf(qv[tt][dt][qq].x - m); e4.y = expf(qv[tt][dt][qq].y - m); e4.z = expf(qv[tt][dt][qq].z - m); e4.w = expf(qv[tt][dt][qq].w - m);
                    sum += (e4.x + e4.y) + (e4.z + e4.w);
                    qv[tt][dt][qq] = e4;
                }
            sum += __shfl_xor(sum, 32, 64);
            const float inv = 1.0f / sum;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const f32x4 q4 = qv[tt][dt][qq];
                    const float qe[4] = {q4.x * inv, q4.y * inv, q4.z * inv, q4.w * inv};
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct) {
                            const float av = aA[dt][ct][4 * qq + e];
                            yo[tt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, qe[e], yo[tt][ct], 0, 0, 0);
                        }
                }
        }
    }
    // ---- row moments over the eight heads (two-pass) ---------------------------------------------------------------------------------
    float mean[NTT], rstd[NTT];
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
        float ps = 0.f;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) ps += yo[tt][ct][r];
        ps += __shfl_xor(ps, 32, 64);
        if (hh == 0) red[0][32 * tt + i][head] = ps;
    }
    __syncthreads();
    const float invD = 1.0f / (float)D;
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(&red[0][32 * tt + i][0]), c = *reinterpret_cast<const f32x4*>(&red[0][32 * tt + i][4]);
        mean[tt] = (((a.x + a.y) + (a.z + a.w)) + ((c.x + c.y) + (c.z + c.w))) * invD;
        float pq = 0.f;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float dlt = yo[tt][ct][r] - mean[tt]; pq = fmaf(dlt, dlt, pq); }
        pq += __shfl_xor(pq, 32, 64);
        if (hh == 0) red[1][32 * tt + i][head] = pq;
    }
    __syncthreads();
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(&red[1][32 * tt + i][0]), c = *reinterpret_cast<const f32x4*>(&red[1][32 * tt + i][4]);
        rstd[tt] = 1.0f / sqrtf((((a.x + a.y) + (a.z + a.w)) + ((c.x + c.y) + (c.z + c.w))) * invD + 1e-5f);
    }
    // ---- SiLU(xhat scale' + shift') on the output registers, 16-byte stores -----------------------------------------------------------
    const float* fr = film + (size_t)(b % bmod) * film_ld + film_off + head * 64 + 4 * hh;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(fr + 32 * ct + 8 * qq), sh = *reinterpret_cast<const f32x4*>(fr + D + 32 * ct + 8 * qq);
#pragma unroll
            for (int tt = 0; tt < NTT; ++tt) {
                const int t = 32 * tt + i;
                if (t < T) {
                    f32x4 o;
                    o.x = at_silu(fmaf((yo[tt][ct][4 * qq] - mean[tt]) * rstd[tt], sc.x, sh.x));
                    o.y = at_silu(fmaf((yo[tt][ct][4 * qq + 1] - mean[tt]) * rstd[tt], sc.y, sh.y));
                    o.z = at_silu(fmaf((yo[tt][ct][4 * qq + 2] - mean[tt]) * rstd[tt], sc.z, sh.z));
                    o.w = at_silu(fmaf((yo[tt][ct][4 * qq + 3] - mean[tt]) * rstd[tt], sc.w, sh.w));
                    *reinterpret_cast<f32x4*>(y + ((size_t)b * T + t) * ldy + head * 64 + 4 * hh + 32 * ct + 8 * qq) = o;
                }
            }
        }
}

// ------------------------------------------------------------------------------------------------
// bf16 MFMA version (head_dim 64, T <= 96): one wave per (sample, head).
//   * lane = channel: K[:, c] and V[:, c] are read with 128-byte coalesced wave loads; the time-softmax of K
//     is lane-local; k^ and v are written TRANSPOSED ([channel][t], bf16) to LDS so that 8 consecutive frames
//     of one channel are a 16-byte MFMA operand fragment;
//   * A = k^T v is 2x2 tiles of v_mfma_f32_32x32x16_bf16 over 6 k-steps (T padded to 96 with zeros);
//   * A stays in registers: the 32x32 accumulator layout (lane = column l, 4-row groups) is re-packed to bf16 and
//     used directly as the operand of y^T = A^T q^T; the channel order that layout implies is mirrored when q is
//     loaded (two 8-byte pieces per step), so no shuffle/LDS transpose is needed;
//   * q-softmax over the head's 64 channels: 32 values per lane + one exchange with lane ^ 32.
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int AT_TMAX = 96;
constexpr int AT_TROW = 208;                 // bytes per transposed row (96 frames * 2 B + 16 pad), 16-byte multiple
constexpr int AT_MAT = 64 * AT_TROW;         // one 64-channel matrix

__device__ __forceinline__ uint32_t pack2_bf16(float lo, float hi) { return pack_bf16_pair(lo, hi); }
__device__ __forceinline__ float bfbits_to_f32(uint16_t v) { return __builtin_bit_cast(float, (uint32_t)v << 16); }

__global__ __launch_bounds__(64) void linear_attention_mfma_kernel(const uint16_t* __restrict__ qkv, int ldq, int T, int D,
                                                                   uint16_t* __restrict__ y, int ldy) {
    __shared__ __attribute__((aligned(16))) char lds[2 * AT_MAT];
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    const int b = blockIdx.y, head = blockIdx.x;
    const uint16_t* base = qkv + (size_t)b * T * ldq + head * 64;

    // ---- K and V columns.  Every (clamped, unconditional) column load of BOTH matrices is issued before
    // anything is consumed: a branch or an early use per frame makes hipcc serialise the HBM round trips,
    // and a wave only has ~6 co-resident peers to hide them behind.
    {
        uint32_t rawk[AT_TMAX], rawv[AT_TMAX];
#pragma unroll
        for (int t = 0; t < AT_TMAX; ++t) rawk[t] = base[(size_t)(t < T ? t : T - 1) * ldq + D + lane];
#pragma unroll
        for (int t = 0; t < AT_TMAX; ++t) rawv[t] = base[(size_t)(t < T ? t : T - 1) * ldq + 2 * D + lane];
        __builtin_amdgcn_sched_barrier(0);
        // K: lane-local softmax over time, written transposed
        float m = -INFINITY;
#pragma unroll
        for (int t = 0; t < AT_TMAX; ++t) {
            const float x = (t < T) ? __builtin_bit_cast(float, rawk[t] << 16) : -INFINITY;
            rawk[t] = __builtin_bit_cast(uint32_t, x);
            m = fmaxf(m, x);
        }
        float ssum = 0.f;
#pragma unroll
        for (int t = 0; t < AT_TMAX; ++t) {
            const float e = (t < T) ? __expf(__builtin_bit_cast(float, rawk[t]) - m) : 0.f;
            rawk[t] = __builtin_bit_cast(uint32_t, e);
            ssum += e;
        }
        const float inv = 1.0f / ssum;
        char* kt = lds + lane * AT_TROW;
#pragma unroll
        for (int t = 0; t < AT_TMAX; t += 2)
            *reinterpret_cast<uint32_t*>(kt + t * 2) = pack2_bf16(__builtin_bit_cast(float, rawk[t]) * inv, __builtin_bit_cast(float, rawk[t + 1]) * inv);
        // V: raw bf16 bits, transposed
        char* vt = lds + AT_MAT + lane * AT_TROW;
#pragma unroll
        for (int t = 0; t < AT_TMAX; t += 2) {
            const uint32_t lo = (t < T) ? rawv[t] : 0u, hi = (t + 1 < T) ? rawv[t + 1] : 0u;
            *reinterpret_cast<uint32_t*>(vt + t * 2) = lo | (hi << 16);
        }
    }
    __syncthreads();

    // ---- A[d][l] = sum_t k^[t][d] v[t][l] -------------------------------------------------------------
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
#pragma unroll
    for (int s = 0; s < AT_TMAX / 16; ++s) {
        u32x4 ak[2], bv[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            ak[a] = *reinterpret_cast<const u32x4*>(lds + (32 * a + i) * AT_TROW + 32 * s + 16 * h);
            bv[a] = *reinterpret_cast<const u32x4*>(lds + AT_MAT + (32 * a + i) * AT_TROW + 32 * s + 16 * h);
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c)
                acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ak[a]), __builtin_bit_cast(bf16x8, bv[c]), acc[a][c], 0, 0, 0);
    }
    // ---- re-pack A as bf16 operand fragments: af[dt][u][lt] holds d = 32dt + 16u + 8(j>>2) + 4h + (j&3), j = 0..7
    u32x4 af[2][2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int w = 0; w < 4; ++w) af[a][u][c][w] = pack2_bf16(acc[a][c][8 * u + 2 * w], acc[a][c][8 * u + 2 * w + 1]);

    // ---- y^T[l][t] = sum_d A[d][l] q^[t][d], 32 frames at a time ----------------------------------------
#pragma unroll
    for (int tt = 0; tt < AT_TMAX / 32; ++tt) {
        const int t = 32 * tt + i;
        if (32 * tt >= T) break;                               // wave-uniform
        const int tc = t < T ? t : T - 1;
        const uint16_t* qr = base + (size_t)tc * ldq;
        float qv[2][2][8];
        float m = -INFINITY;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int d0 = 32 * a + 16 * u + 4 * h;
                const u32x2 lo = *reinterpret_cast<const u32x2*>(qr + d0);
                const u32x2 hi = *reinterpret_cast<const u32x2*>(qr + d0 + 8);
                const uint32_t w[4] = {lo.x, lo.y, hi.x, hi.y};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    qv[a][u][2 * k] = __builtin_bit_cast(float, w[k] << 16);
                    qv[a][u][2 * k + 1] = __builtin_bit_cast(float, w[k] & 0xffff0000u);
                    m = fmaxf(m, fmaxf(qv[a][u][2 * k], qv[a][u][2 * k + 1]));
                }
            }
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float ssum = 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int k = 0; k < 8; ++k) { qv[a][u][k] = __expf(qv[a][u][k] - m); ssum += qv[a][u][k]; }
        ssum += __shfl_xor(ssum, 32, 64);
        const float inv = 1.0f / ssum;
        f32x16 yacc[2];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) yacc[c][r] = 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                u32x4 qf;
#pragma unroll
                for (int w = 0; w < 4; ++w) qf[w] = pack2_bf16(qv[a][u][2 * w] * inv, qv[a][u][2 * w + 1] * inv);
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    yacc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[a][u][c]), __builtin_bit_cast(bf16x8, qf), yacc[c], 0, 0, 0);
            }
        // D[l][t]: lane (t, h) holds l = 32c + 8q + 4h + e  -> 8-byte stores of 4 consecutive channels
        if (t < T) {
            uint16_t* yr = y + ((size_t)b * T + t) * ldy + head * 64;
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    u32x2 o;
                    o.x = pack2_bf16(yacc[c][4 * q], yacc[c][4 * q + 1]);
                    o.y = pack2_bf16(yacc[c][4 * q + 2], yacc[c][4 * q + 3]);
                    *reinterpret_cast<u32x2*>(yr + 32 * c + 8 * q + 4 * h) = o;
                }
        }
    }
}

// Same algorithm on the TILED bf16 layout of the token-per-lane Linears (tl_linear.hip): element (token, n) of a
// [M, Wd] tensor lives at ((token >> 5) * (Wd >> 4) + (n >> 4)) * 512 + (token & 31) * 16 + (n & 15).
__global__ __launch_bounds__(64, 2) void linear_attention_tiled_kernel(const uint16_t* __restrict__ qkv, int half_batches, int half_row0,
                                                                    int T, int D, uint16_t* __restrict__ y, int rev) {
    __shared__ __attribute__((aligned(16))) char lds[2 * AT_MAT];
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    const int b = rev ? (int)gridDim.y - 1 - (int)blockIdx.y : (int)blockIdx.y, head = blockIdx.x;     // rev: clips in descending order (tl_block_index)
    const int tok0 = b < half_batches ? b * T : half_row0 + (b - half_batches) * T;
    const int KTQ = (3 * D) >> 4;                               // 16-feature tiles per token block of qkv
    auto tok_off = [&](int t, int ktiles) -> size_t { const int tg = tok0 + t; return ((size_t)(tg >> 5) * ktiles) * 512 + (tg & 31) * 16; };
    // ---- K and V: 16-byte loads (one token x 8 channels per lane; lane = (cl = 8-channel group, g = 12-frame block)),
    // all 24 issued before anything is consumed.  Lane cl walks its 12 frames rotated by 2 cl so that the transposed
    // LDS writes of the 8 lanes that share a frame block land on different banks (rows 8 apart alias otherwise).
    u32x2 qraw[AT_TMAX / 32][2][2][2];
    {
        const int cl = lane & 7, g = lane >> 3;
        const uint16_t* kb = qkv + (size_t)(((D + head * 64) >> 4) + (cl >> 1)) * 512 + (cl & 1) * 8;
        const uint16_t* vb = qkv + (size_t)(((2 * D + head * 64) >> 4) + (cl >> 1)) * 512 + (cl & 1) * 8;
        u32x4 rk[12], rv[12];
        int fr[12];
        // (the q rows of all three 32-frame groups are requested here as well: one exposed HBM round trip per wave)
#pragma unroll
        for (int tt = 0; tt < AT_TMAX / 32; ++tt) {
            const int t = 32 * tt + i, tc = t < T ? t : T - 1;
            const uint16_t* qr = qkv + tok_off(tc, KTQ) + (size_t)(head * 4) * 512 + 4 * h;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    qraw[tt][a][u][0] = *reinterpret_cast<const u32x2*>(qr + (2 * a + u) * 512);
                    qraw[tt][a][u][1] = *reinterpret_cast<const u32x2*>(qr + (2 * a + u) * 512 + 8);
                }
        }
#pragma unroll
        for (int j = 0; j < 12; ++j) { int r = j + 2 * cl; r = r >= 12 ? r - 12 : r; r = r >= 12 ? r - 12 : r; fr[j] = 12 * g + r; }
#pragma unroll
        for (int j = 0; j < 12; ++j) rk[j] = *reinterpret_cast<const u32x4*>(kb + tok_off(fr[j] < T ? fr[j] : T - 1, KTQ));
#pragma unroll
        for (int j = 0; j < 12; ++j) rv[j] = *reinterpret_cast<const u32x4*>(vb + tok_off(fr[j] < T ? fr[j] : T - 1, KTQ));
        __builtin_amdgcn_sched_barrier(0);
        // K: softmax over time per channel = 12 lane-local frames + the 8 lanes that share cl (lane bits 3..5)
        float e[8][12], mx[8], sm[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float m = -INFINITY;
#pragma unroll
            for (int j = 0; j < 12; ++j) {
                const uint32_t w = rk[j][c >> 1];
                const float x = fr[j] < T ? __builtin_bit_cast(float, (c & 1) ? (w & 0xffff0000u) : (w << 16)) : -INFINITY;
                e[c][j] = x;
                m = fmaxf(m, x);
            }
            mx[c] = m;
        }
#pragma unroll
        for (int sh = 8; sh <= 32; sh <<= 1)
#pragma unroll
            for (int c = 0; c < 8; ++c) mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], sh, 64));
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float ssum = 0.f;
            const float nm = -mx[c] * 1.44269504088896341f;
#pragma unroll
            for (int j = 0; j < 12; ++j) { e[c][j] = __builtin_amdgcn_exp2f(fmaf(e[c][j], 1.44269504088896341f, nm)); ssum += e[c][j]; }   // exp(-inf) = 0 on masked frames
            sm[c] = ssum;
        }
#pragma unroll
        for (int sh = 8; sh <= 32; sh <<= 1)
#pragma unroll
            for (int c = 0; c < 8; ++c) sm[c] += __shfl_xor(sm[c], sh, 64);
        // transposed rows [channel][frame]; registers (2p, 2p+1) hold frames 12 g + (2p + 2cl) % 12, + 1
        int toff[6];
#pragma unroll
        for (int pp = 0; pp < 6; ++pp) toff[pp] = fr[2 * pp] * 2;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float inv = 1.0f / sm[c];
            char* kt = lds + (8 * cl + c) * AT_TROW;
            char* vt = lds + AT_MAT + (8 * cl + c) * AT_TROW;
#pragma unroll
            for (int pp = 0; pp < 6; ++pp) {
                *reinterpret_cast<uint32_t*>(kt + toff[pp]) = pack2_bf16(e[c][2 * pp] * inv, e[c][2 * pp + 1] * inv);
                const uint32_t w0 = rv[2 * pp][c >> 1], w1 = rv[2 * pp + 1][c >> 1];
                const uint32_t lo = (c & 1) ? (w0 >> 16) : (w0 & 0xffffu), hi = (c & 1) ? (w1 >> 16) : (w1 & 0xffffu);
                // (frames >= T hold a clamped, finite duplicate of frame T - 1: their softmax weight above is exactly 0)
                *reinterpret_cast<uint32_t*>(vt + toff[pp]) = lo | (hi << 16);
            }
        }
    }
    __syncthreads();

    // ---- A[d][l] = sum_t k^[t][d] v[t][l] -------------------------------------------------------------
    // v channels enter the first product in pi order (pi(8q + 4h + e) = 16 (q >> 1) + 8h + 4 (q & 1) + e, as in
    // tl_linear.hip): the output accumulator of the second product then holds 8 consecutive channels per lane
    const int ipi = 16 * (i >> 4) + 8 * ((i >> 2) & 1) + 4 * ((i >> 3) & 1) + (i & 3);
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
#pragma unroll
    for (int s = 0; s < AT_TMAX / 16; ++s) {
        u32x4 ak[2], bv[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            ak[a] = *reinterpret_cast<const u32x4*>(lds + (32 * a + i) * AT_TROW + 32 * s + 16 * h);
            bv[a] = *reinterpret_cast<const u32x4*>(lds + AT_MAT + (32 * a + ipi) * AT_TROW + 32 * s + 16 * h);
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c)
                acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ak[a]), __builtin_bit_cast(bf16x8, bv[c]), acc[a][c], 0, 0, 0);
    }
    // ---- re-pack A as bf16 operand fragments: af[dt][u][lt] holds d = 32dt + 16u + 8(j>>2) + 4h + (j&3), j = 0..7
    u32x4 af[2][2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int w = 0; w < 4; ++w) af[a][u][c][w] = pack2_bf16(acc[a][c][8 * u + 2 * w], acc[a][c][8 * u + 2 * w + 1]);

    // ---- y^T[l][t] = sum_d A[d][l] q^[t][d], 32 frames at a time ----------------------------------------
#pragma unroll
    for (int tt = 0; tt < AT_TMAX / 32; ++tt) {
        const int t = 32 * tt + i;
        if (32 * tt >= T) break;                               // wave-uniform
        float qv[2][2][8];
        float m = -INFINITY;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const u32x2 lo = qraw[tt][a][u][0], hi = qraw[tt][a][u][1];   // features 64 head + 32a + 16u + 4h (+8)
                const uint32_t w[4] = {lo.x, lo.y, hi.x, hi.y};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    qv[a][u][2 * k] = __builtin_bit_cast(float, w[k] << 16);
                    qv[a][u][2 * k + 1] = __builtin_bit_cast(float, w[k] & 0xffff0000u);
                    m = fmaxf(m, fmaxf(qv[a][u][2 * k], qv[a][u][2 * k + 1]));
                }
            }
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float ssum = 0.f;
        const float nmq = -m * 1.44269504088896341f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int k = 0; k < 8; ++k) { qv[a][u][k] = __builtin_amdgcn_exp2f(fmaf(qv[a][u][k], 1.44269504088896341f, nmq)); ssum += qv[a][u][k]; }
        ssum += __shfl_xor(ssum, 32, 64);
        const float inv = 1.0f / ssum;
        f32x16 yacc[2];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) yacc[c][r] = 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                u32x4 qf;
#pragma unroll
                for (int w = 0; w < 4; ++w) qf[w] = pack2_bf16(qv[a][u][2 * w] * inv, qv[a][u][2 * w + 1] * inv);
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    yacc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[a][u][c]), __builtin_bit_cast(bf16x8, qf), yacc[c], 0, 0, 0);
            }
        // D[l][t]: lane (t, h), accumulator half cc holds l = 32c + 16cc + 8h + (0..7) -> one 16-byte store per (c, cc)
        if (t < T) {
            uint16_t* yr = y + tok_off(t, D >> 4) + (size_t)(head * 4) * 512 + 8 * h;
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
                    u32x4 o;
                    o.x = pack2_bf16(yacc[c][8 * cc + 0], yacc[c][8 * cc + 1]); o.y = pack2_bf16(yacc[c][8 * cc + 2], yacc[c][8 * cc + 3]);
                    o.z = pack2_bf16(yacc[c][8 * cc + 4], yacc[c][8 * cc + 5]); o.w = pack2_bf16(yacc[c][8 * cc + 6], yacc[c][8 * cc + 7]);
                    *reinterpret_cast<u32x4*>(yr + (2 * c + cc) * 512) = o;
                }
        }
    }
}

int launch_linear_attention_tiled(const void* qkv, int nbatch, int half_batches, int half_row0, int frames, int D, void* y,
                                  hipStream_t s, int rev) {
    DSH_REQUIRE(D % 64 == 0 && frames > 0 && frames <= AT_TMAX, "linear_attention_tiled: needs 64-channel heads and <= 96 frames");
    hipLaunchKernelGGL(linear_attention_tiled_kernel, dim3(D / 64, nbatch), dim3(64), 0, s, reinterpret_cast<const uint16_t*>(qkv),
                       half_batches, half_row0, frames, D, reinterpret_cast<uint16_t*>(y), rev);
    DSH_HIP_CHECK(hipGetLastError());
    return 0;
}

template <typename T>
int launch_linear_attention(const T* qkv, int ldq, int nbatch, int frames, int D, int head_dim, T* y, int ldy,
                            hipStream_t s) {
    DSH_REQUIRE(D % 64 == 0, "linear_attention: latent width must be a multiple of 64");
    DSH_REQUIRE(head_dim == 64 || head_dim == 16, "linear_attention: head_dim must be 64 or 16");
    dim3 grid(D / 64, nbatch);
    if (sizeof(T) == 2 && head_dim == 64 && frames <= AT_TMAX && ldq % 4 == 0 && ldy % 4 == 0) {
        hipLaunchKernelGGL(linear_attention_mfma_kernel, grid, dim3(64), 0, s, reinterpret_cast<const uint16_t*>(qkv), ldq,
                           frames, D, reinterpret_cast<uint16_t*>(y), ldy);
        DSH_HIP_CHECK(hipGetLastError());
        return 0;
    }
    // (fp32 path, 64-channel heads: the loop form was 62 us per launch at the config-2 batch — 10 % of its step — against
    //  a few microseconds of data; with the columns preloaded the launch is one memory round trip plus the arithmetic)
    // round 6: fp32, 64-channel heads: both products on the exact-fp32 matrix pipe (DSH_ATTN_F32_MFMA=0: the VALU kernels below)
    static const int f32_mfma = [] { const char* e = getenv("DSH_ATTN_F32_MFMA"); return e ? atoi(e) : 1; }();
    if (f32_mfma && sizeof(T) == 4 && head_dim == 64 && frames <= 96 && ldq % 4 == 0 && ldy % 4 == 0 && ((uintptr_t)qkv % 16) == 0 && ((uintptr_t)y % 16) == 0) {
        const float* qf = reinterpret_cast<const float*>(qkv);
        float* yf = reinterpret_cast<float*>(y);
        if (frames <= 32) hipLaunchKernelGGL((linear_attention_f32_mfma_kernel<32>), grid, dim3(64), 0, s, qf, ldq, frames, D, yf, ldy);
        else if (frames <= 36) hipLaunchKernelGGL((linear_attention_f32_mfma_kernel<36>), grid, dim3(64), 0, s, qf, ldq, frames, D, yf, ldy);   // (BEAT: 34 frames, 232 registers)
        else if (frames <= 64) hipLaunchKernelGGL((linear_attention_f32_mfma_kernel<64>), grid, dim3(64), 0, s, qf, ldq, frames, D, yf, ldy);
        else hipLaunchKernelGGL((linear_attention_f32_mfma_kernel<96>), grid, dim3(64), 0, s, qf, ldq, frames, D, yf, ldy);
        DSH_HIP_CHECK(hipGetLastError());
        return 0;
    }
    if (head_dim == 64 && frames <= 48)
        hipLaunchKernelGGL((linear_attention_pre_kernel<T, 64, 48>), grid, dim3(64), 0, s, qkv, ldq, frames, D, y, ldy);
    else if (head_dim == 64 && frames <= 96)
        hipLaunchKernelGGL((linear_attention_pre_kernel<T, 64, 96>), grid, dim3(64), 0, s, qkv, ldq, frames, D, y, ldy);
    else if (head_dim == 64)
        hipLaunchKernelGGL((linear_attention_kernel<T, 64>), grid, dim3(64), 0, s, qkv, ldq, frames, D, y, ldy);
    else if (frames <= 96)
        hipLaunchKernelGGL((linear_attention_pre_kernel<T, 16>), grid, dim3(64), 0, s, qkv, ldq, frames, D, y, ldy);
    else
        hipLaunchKernelGGL((linear_attention_kernel<T, 16>), grid, dim3(64), 0, s, qkv, ldq, frames, D, y, ldy);
    DSH_HIP_CHECK(hipGetLastError());
    return 0;
}
// fp32, D = 512, eight 64-channel heads, T <= 96: attention core + the StylizationBlock front behind it in one launch (see the kernel above)
bool linear_attention_sty_f32_supported(int frames, int D, int head_dim, int ldq, int ldy) {
    static const int on = [] { const char* e = getenv("DSH_ATTN_STY"); return e ? atoi(e) : 1; }();
    // (up to 64 frames: the 96-frame instantiation spills at eight waves per block — longer windows keep the two launches)
    return on && D == 512 && head_dim == 64 && frames > 0 && frames <= 64 && ldq % 4 == 0 && ldy % 4 == 0;
}
int launch_linear_attention_sty_f32(const float* qkv, int ldq, int nbatch, int frames, int D, float* s_out, int ldy, const float* film, int film_ld, int film_off,
                                    int bmod, hipStream_t s) {
    DSH_REQUIRE(linear_attention_sty_f32_supported(frames, D, 64, ldq, ldy) && film && bmod > 0 && film_ld % 4 == 0 && film_off % 4 == 0, "linear_attention_sty: unsupported shape");
    DSH_REQUIRE(((uintptr_t)qkv % 16) == 0 && ((uintptr_t)s_out % 16) == 0 && ((uintptr_t)film % 16) == 0, "linear_attention_sty: 16-byte alignment");
    const dim3 grid(nbatch);
    if (frames <= 32) hipLaunchKernelGGL((linear_attention_f32_mfma_sty_kernel<32>), grid, dim3(512), 0, s, qkv, ldq, frames, D, s_out, ldy, film, film_ld, film_off, bmod);
    else if (frames <= 36) hipLaunchKernelGGL((linear_attention_f32_mfma_sty_kernel<36>), grid, dim3(512), 0, s, qkv, ldq, frames, D, s_out, ldy, film, film_ld, film_off, bmod);
    else hipLaunchKernelGGL((linear_attention_f32_mfma_sty_kernel<64>), grid, dim3(512), 0, s, qkv, ldq, frames, D, s_out, ldy, film, film_ld, film_off, bmod);
    DSH_HIP_CHECK(hipGetLastError());
    return 0;
}
template int launch_linear_attention<float>(const float*, int, int, int, int, int, float*, int, hipStream_t);
template int launch_linear_attention<bf16>(const bf16*, int, int, int, int, int, bf16*, int, hipStream_t);

}  // namespace dsh
