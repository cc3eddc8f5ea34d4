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
#include "dsh_common.h"
#include "dsh_kernels.h"

namespace dsh {

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

template <typename T>
int launch_linear_attention(const T* qkv, int ldq, int nbatch, int frames, int D, int head_dim, T* y, int ldy,
                            hipStream_t s) {
    DSH_REQUIRE(D % 64 == 0, "linear_attention: latent width must be a multiple of 64");
    DSH_REQUIRE(head_dim == 64 || head_dim == 16, "linear_attention: head_dim must be 64 or 16");
    dim3 grid(D / 64, nbatch);
    if (head_dim == 64)
        hipLaunchKernelGGL((linear_attention_kernel<T, 64>), grid, dim3(64), 0, s, qkv, ldq, frames, D, y, ldy);
    else
        hipLaunchKernelGGL((linear_attention_kernel<T, 16>), grid, dim3(64), 0, s, qkv, ldq, frames, D, y, ldy);
    DSH_HIP_CHECK(hipGetLastError());
    return 0;
}
template int launch_linear_attention<float>(const float*, int, int, int, int, int, float*, int, hipStream_t);
template int launch_linear_attention<bf16>(const bf16*, int, int, int, int, int, bf16*, int, hipStream_t);

}  // namespace dsh
