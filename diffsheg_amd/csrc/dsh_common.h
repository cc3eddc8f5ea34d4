// Shared declarations for the DiffSHEG MI355X (gfx950) hot-path library.
// Written for CDNA4 only: wave64, MFMA, 160 KiB LDS.  No CUDA/NVIDIA compatibility paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

namespace dsh {

// ---- bf16 storage type (raw bits; round-to-nearest-even conversion) ------------------------
struct bf16 {
    uint16_t v;
};

__host__ __device__ inline float bf16_to_f32(bf16 h) {
    union { uint32_t u; float f; } c;
    c.u = ((uint32_t)h.v) << 16;
    return c.f;
}
__host__ __device__ inline bf16 f32_to_bf16(float f) {
    union { uint32_t u; float f; } c;
    c.f = f;
    uint32_t u = c.u;
    bf16 r;
    if ((u & 0x7fffffffu) > 0x7f800000u) {  // NaN
        r.v = (uint16_t)((u >> 16) | 0x40);
        return r;
    }
    u += 0x7fffu + ((u >> 16) & 1u);
    r.v = (uint16_t)(u >> 16);
    return r;
}

// two fp32 -> packed bf16 pair (lo in bits 15:0), round-to-nearest-even: one v_cvt_pk_bf16_f32 on gfx950
__device__ __forceinline__ uint32_t pack_bf16_pair(float lo, float hi) {
    typedef __bf16 bf16x2_hw __attribute__((ext_vector_type(2)));
    typedef float f32x2_hw __attribute__((ext_vector_type(2)));
    const f32x2_hw v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_hw));
}

template <typename T> __host__ __device__ inline float to_f32(T x);
template <> __host__ __device__ inline float to_f32<float>(float x) { return x; }
template <> __host__ __device__ inline float to_f32<bf16>(bf16 x) { return bf16_to_f32(x); }
template <typename T> __host__ __device__ inline T from_f32(float x);
template <> __host__ __device__ inline float from_f32<float>(float x) { return x; }
template <> __host__ __device__ inline bf16 from_f32<bf16>(float x) { return f32_to_bf16(x); }

enum Act : int { ACT_NONE = 0, ACT_SILU = 1, ACT_GELU = 2 };

__device__ inline float silu_f(float x) { return x / (1.0f + expf(-x)); }
__device__ inline float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ inline float apply_act(float x, int act) {
    if (act == ACT_SILU) return silu_f(x);
    if (act == ACT_GELU) return gelu_f(x);
    return x;
}

// ---- error plumbing (no exceptions cross the C ABI) ----------------------------------------
void set_last_error(const std::string& msg);
const char* last_error_cstr();

#define DSH_HIP_CHECK(expr)                                                                     \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            dsh::set_last_error(std::string(#expr) + " failed: " + hipGetErrorString(_e) +      \
                                " (" + __FILE__ + ":" + std::to_string(__LINE__) + ")");        \
            return -2;                                                                          \
        }                                                                                       \
    } while (0)

#define DSH_REQUIRE(cond, msg)                                                                  \
    do {                                                                                        \
        if (!(cond)) {                                                                          \
            dsh::set_last_error(std::string("invalid argument: ") + (msg) + " [" #cond "]");    \
            return -1;                                                                          \
        }                                                                                       \
    } while (0)

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline int round_up(int a, int b) { return ceil_div(a, b) * b; }

// ---- GEMM:  C[M,N] = epilogue(A[M,K] * W[N,K]^T) -------------------------------------------
// A and W are K-contiguous ("NT" layout == torch.nn.Linear weights as stored).  K must be a
// multiple of GEMM_BK_BYTES/sizeof(T) and both operands zero-padded along K; rows are clamped,
// so M and N are arbitrary.
struct GemmArgs {
    const void* A;      int lda;        // [M, K]  element type T
    const void* W;      int ldw;        // [N, K]  element type T
    const float* bias;                  // [N] or null
    const float* R;     int ldr;        // fp32 residual [*, N] or null; row = m % res_mod (res_mod>0) else m
    int res_mod;
    float* Cf;          int ldcf;       // fp32 output or null
    void* Ct;           int ldct;       // T output or null
    int M, N, K;
    int act;                            // activation on (acc + bias) ...
    int act_after_res;                  // ... or, if 1, on (acc + bias + residual)
    int nt_n, nt_m;                     // tile counts (filled by the launcher)
};
int launch_gemm_f32(const GemmArgs& a, hipStream_t s);
int launch_gemm_bf16(const GemmArgs& a, hipStream_t s);
template <typename T> inline int launch_gemm(const GemmArgs& a, hipStream_t s);
template <> inline int launch_gemm<float>(const GemmArgs& a, hipStream_t s) { return launch_gemm_f32(a, s); }
template <> inline int launch_gemm<bf16>(const GemmArgs& a, hipStream_t s) { return launch_gemm_bf16(a, s); }
constexpr int GEMM_BK_BYTES = 128;      // K tile in bytes (32 fp32 / 64 bf16)
template <typename T> constexpr int gemm_k_align() { return GEMM_BK_BYTES / (int)sizeof(T); }

}  // namespace dsh
