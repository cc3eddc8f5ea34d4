// Element-wise sampler updates (HBM-bound; one fused pass per step instead of the reference's ~10
// tiny aten kernels + H2D table copies per step, /root/reference/models/gaussian_diffusion.py):
//
//   ddim_step   x0 = c1 x - c2 eps; eps' = (c1 x - x0)/c2; x <- sqrt(ab_prev) x0 + sqrt(1-ab_prev) eps'
//               (+ RePaint blend)                                     :614-622, :993-1032, :1034-1056
//   undo_step   x <- sqrt(1-beta) x + sqrt(beta) n                   :464-473
//   ddpm_step   mean = coef1 x0 + coef2 x;  x <- mean + sigma n      :598-600, :747-773
//
// Every product / sum is an individually rounded fp32 op (__fmul_rn/__fadd_rn: no FMA contraction),
// in the reference's operation order, so the update itself is bit-identical to the aten sequence
// given identical inputs.  Scalars arrive pre-rounded exactly as the reference rounds them
// (fp64 table -> fp32 at gather, gaussian_diffusion.py:1514; sqrt taken in fp32 where the
// reference takes it on an fp32 tensor).
#include <algorithm>

#include "dsh_common.h"
#include "dsh_kernels.h"

namespace dsh {

__global__ void ddim_step_kernel(DdimStepArgs a) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const int tc = a.frames * a.channels;
    const bool ranged = a.c_hi > a.c_lo;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += stride) {
        if (ranged) { const int cc = (int)(i % (size_t)a.channels); if (cc < a.c_lo || cc >= a.c_hi) continue; }
        const float x = a.x[i];
        const float e = a.eps[i];
        const float c1x = __fmul_rn(a.c1, x);
        float x0 = __fsub_rn(c1x, __fmul_rn(a.c2, e));
        if (a.clip) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
        const float e2 = __fdiv_rn(__fsub_rn(c1x, x0), a.c2);
        float s = __fadd_rn(__fmul_rn(x0, a.sqrt_ab_prev), __fmul_rn(a.coef_eps, e2));
        if (a.noise1) s = __fadd_rn(s, __fmul_rn(a.sigma, a.noise1[i]));
        if (a.x0_out) a.x0_out[i] = x0;
        const int t = (int)((i % (size_t)tc) / (size_t)a.channels);
        if (a.mask) {
            // --same_overlap_noisy (gaussian_diffusion.py:1040-1042): the out-painted frames take the previous window's saved
            // NOISY tail of this level instead of a freshly noised copy of its final tail (no Gaussian draw in that branch)
            float g;
            if (a.tail_in) {
                const size_t b = i / (size_t)tc;
                const int c = (int)(i % (size_t)a.channels);
                g = t < a.overlap_len ? a.tail_in[(b * a.overlap_len + t) * a.channels + c] : a.gt[i];
            } else g = __fadd_rn(__fmul_rn(a.sqrt_ab_prev, a.gt[i]), __fmul_rn(a.sqrt_1m_ab_prev, a.noise2[i]));
            if (a.blend) {
                if (t < a.overlap_len) {
                    // torch.linspace(0, 1, L)[t] in fp32: symmetric formula start + step*t / end - step*(L-1-t)
                    const int L = a.overlap_len;
                    float w;
                    if (L == 1) w = 0.f;
                    else {
                        const float step = __fdiv_rn(1.0f, (float)(L - 1));
                        w = (t < L / 2) ? __fmul_rn(step, (float)t) : __fsub_rn(1.0f, __fmul_rn(step, (float)(L - 1 - t)));
                    }
                    g = __fadd_rn(__fmul_rn(g, __fsub_rn(1.0f, w)), __fmul_rn(s, w));
                }
            }
            s = a.mask[i] ? g : s;
        }
        a.x[i] = s;
        if (a.tail_out && t >= a.frames - a.overlap_len) {          // saved_noisy_tail[t] = x[..., -L:, :]  (:1058-1060)
            const size_t b = i / (size_t)tc;
            const int c = (int)(i % (size_t)a.channels);
            a.tail_out[(b * a.overlap_len + (t - (a.frames - a.overlap_len))) * a.channels + c] = s;
        }
    }
}

static inline int grid_for(size_t n) {
    size_t b = (n + 255) / 256;
    return (int)(b < 2048 ? (b ? b : 1) : 2048);
}

int launch_ddim_step(const DdimStepArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(ddim_step_kernel, dim3(grid_for(a.n)), dim3(256), 0, s, a);
    DSH_HIP_CHECK(hipGetLastError());
    return 0;
}

__global__ void undo_step_kernel(float* x, const float* noise, float sa, float sb, size_t n, int channels, int c_lo, int c_hi) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const bool ranged = c_hi > c_lo;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (ranged) { const int cc = (int)(i % (size_t)channels); if (cc < c_lo || cc >= c_hi) continue; }
        x[i] = __fadd_rn(__fmul_rn(sa, x[i]), __fmul_rn(sb, noise[i]));
    }
}
int launch_undo_step(float* x, const float* noise, float sqrt_1m_beta, float sqrt_beta, size_t n, hipStream_t s, int channels, int c_lo, int c_hi) {
    DSH_REQUIRE(c_hi <= c_lo || channels > 0, "undo_step: a channel range needs the channel count");
    hipLaunchKernelGGL(undo_step_kernel, dim3(grid_for(n)), dim3(256), 0, s, x, noise, sqrt_1m_beta, sqrt_beta, n, channels, c_lo, c_hi);
    DSH_HIP_CHECK(hipGetLastError());
    return 0;
}

__global__ void ddpm_step_kernel(DdpmStepArgs a) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const bool ranged = a.c_hi > a.c_lo;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += stride) {
        if (ranged) { const int cc = (int)(i % (size_t)a.channels); if (cc < a.c_lo || cc >= a.c_hi) continue; }
        const float x = a.x[i];
        float x0 = __fsub_rn(__fmul_rn(a.c1, x), __fmul_rn(a.c2, a.eps[i]));
        if (a.clip) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
        const float mean = __fadd_rn(__fmul_rn(a.coef1, x0), __fmul_rn(a.coef2, x));
        if (a.x0_out) a.x0_out[i] = x0;
        // reference: mean + nonzero_mask * exp(0.5*logvar) * noise  (left-to-right products)
        a.x[i] = __fadd_rn(mean, __fmul_rn(a.sigma, a.noise[i]));
    }
}
int launch_ddpm_step(const DdpmStepArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(ddpm_step_kernel, dim3(grid_for(a.n)), dim3(256), 0, s, a);
    DSH_HIP_CHECK(hipGetLastError());
    return 0;
}

__global__ void fill_i64_kernel(int64_t* p, int64_t v, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
int launch_fill_i64(int64_t* p, int64_t v, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(fill_i64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, v, n);
    DSH_HIP_CHECK(hipGetLastError());
    return 0;
}
__global__ void fill_f32_kernel(float* p, float v, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
int launch_fill_f32(float* p, float v, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(fill_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, v, n);
    DSH_HIP_CHECK(hipGetLastError());
    return 0;
}

// one launch for the per-step scalars of an evaluation: t, the two x0 coefficients and the spaced level (timestep-cache slot)
__global__ void fill_step_kernel(int64_t* t, float* c1, float* c2, int64_t* level, int64_t tv, float c1v, float c2v, int64_t lv, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { t[i] = tv; c1[i] = c1v; c2[i] = c2v; }
    if (i == 0) *level = lv;
}
int launch_fill_step(int64_t* t, float* c1, float* c2, int64_t* level, int64_t tv, float c1v, float c2v, int64_t lv, int n, hipStream_t s) {
    hipLaunchKernelGGL(fill_step_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, t, c1, c2, level, tv, c1v, c2v, lv, n);
    DSH_HIP_CHECK(hipGetLastError());
    return 0;
}

// save (restore == 0) / restore the x-independent results of one evaluation to / from timestep-cache slot *level: up to four
// byte ranges (multiples of 16) of the workspace <-> slots + *level * stride + off[seg]
__global__ void level_copy_kernel(LevelCopyArgs a) {
    char* slot = a.slots + (size_t)(*a.level) * a.stride;
    const size_t stride = (size_t)gridDim.x * blockDim.x * 16;
    for (int sg = 0; sg < a.nseg; ++sg) {
        char* w = a.work[sg];
        char* c = slot + a.off[sg];
        for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16; i < a.bytes[sg]; i += stride) {
            if (a.restore) *reinterpret_cast<uint4*>(w + i) = *reinterpret_cast<const uint4*>(c + i);
            else *reinterpret_cast<uint4*>(c + i) = *reinterpret_cast<const uint4*>(w + i);
        }
    }
}
int launch_level_copy(const LevelCopyArgs& a, hipStream_t s) {
    size_t mx = 0;
    for (int i = 0; i < a.nseg; ++i) { DSH_REQUIRE(a.bytes[i] % 16 == 0 && a.off[i] % 16 == 0, "level_copy: ranges must be 16-byte multiples"); mx = std::max(mx, a.bytes[i]); }
    const size_t blocks = std::min<size_t>(std::max<size_t>((mx / 16 + 255) / 256, 1), 1024);
    hipLaunchKernelGGL(level_copy_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
    DSH_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---- Philox4x32-10 counter RNG + Box-Muller (perf runs; parity runs inject recorded noise) ------
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

// row_keys == null: one stream for the whole tensor (key `seed`, counter offset + quad index).
// row_keys != null: row b (= n_row consecutive values, n_row % 4 == 0) is its own stream: same key `seed`, the row key in the two
// HIGH words of the 128-bit Philox counter (they are zero in the whole-tensor mode) and the counter offset + quad index INSIDE
// the row in the low words — so a chain's noise does not depend on which batch / rank it is sampled in, and two (seed, row key)
// pairs share a stream only if both components are equal (round 2 XORed the row key into the seed: (s + 1) ^ 0 == s ^ 1).
__global__ void philox_randn_kernel(float* out, size_t n, uint64_t seed, uint64_t offset, const uint64_t* __restrict__ row_keys,
                                    size_t row_quads) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t nquad = (n + 3) / 4;
    for (size_t qd = (size_t)blockIdx.x * blockDim.x + threadIdx.x; qd < nquad; qd += stride) {
        uint64_t ctr = offset + qd, rk = 0;
        const uint64_t key = seed;
        if (row_keys) { const size_t b = qd / row_quads; ctr = offset + (qd - b * row_quads); rk = row_keys[b]; }
        uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)rk, (uint32_t)(rk >> 32)};
        uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            philox_round(c, k0, k1);
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        float z[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float u1 = ((float)(c[2 * h] >> 8) + 0.5f) * (1.0f / 16777216.0f);      // (0,1)
            const float u2 = ((float)(c[2 * h + 1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
            const float r = sqrtf(-2.0f * logf(u1));
            float sn, cs;
            sincosf(6.283185307179586f * u2, &sn, &cs);
            z[2 * h] = r * cs;
            z[2 * h + 1] = r * sn;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const size_t i = qd * 4 + j;
            if (i < n) out[i] = z[j];
        }
    }
}
int launch_philox_randn(float* out, size_t n, uint64_t seed, uint64_t offset, hipStream_t s) {
    hipLaunchKernelGGL(philox_randn_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, s, out, n, seed, offset,
                       (const uint64_t*)nullptr, (size_t)1);
    DSH_HIP_CHECK(hipGetLastError());
    return 0;
}
int launch_philox_randn_rows(float* out, int rows, size_t n_row, uint64_t seed, uint64_t offset, const uint64_t* row_keys,
                             hipStream_t s) {
    DSH_REQUIRE(rows > 0 && n_row % 4 == 0 && row_keys, "philox_randn_rows: row length must be a multiple of 4");
    const size_t n = (size_t)rows * n_row;
    hipLaunchKernelGGL(philox_randn_kernel, dim3(grid_for(n / 4)), dim3(256), 0, s, out, n, seed, offset, row_keys, n_row / 4);
    DSH_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace dsh
