// Microbenchmark (measurement tool, not product code): sustained MFMA rate of this MI355X under a pure-MFMA loop — the
// ceiling a real kernel can approach at the clock the part actually holds under that load — for the two instructions the
// denoiser uses: v_mfma_f32_32x32x16_bf16 (bf16 path) and v_mfma_f32_32x32x2_f32 (exact-fp32 parity path).
//   hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, unsigned long long* clk) {
    f32x16 acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    bf16x8 ab, bb;
#pragma unroll
    for (int j = 0; j < 8; ++j) { ab[j] = (short)(0x3f80 + threadIdx.x % 3); bb[j] = (short)(0x3f80 + threadIdx.x % 5); }
    const float af = 1.0f + threadIdx.x % 3, bf = 1.0f + threadIdx.x % 5;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int a = 0; a < NACC; ++a) {
                if (KIND == 0) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[a], 0, 0, 0);
                else acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc[a], 0, 0, 0);
            }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    if (s == 12345.678f) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;
}

template <int KIND, int NACC>
static void run(const char* name, int blocks_per_cu, int cus, double flops_per_mfma) {
    float* out; unsigned long long* clk;
    CK(hipMalloc(&out, 4)); CK(hipMalloc(&clk, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = KIND == 0 ? 20000 : 10000;
    const int grid = cus * blocks_per_cu;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((mfma_loop<KIND, NACC>), dim3(grid), dim3(256), 0, 0, out, iters, clk);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long c; CK(hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost));
        const double mfmas = (double)grid * 4 * iters * 8 * NACC;
        const double tf = mfmas * flops_per_mfma / (ms * 1e-3) / 1e12;
        // s_memtime / readcyclecounter tick at a fixed 100 MHz on this part: cycles are not shader clocks; derive the shader
        // clock from the MFMA issue rate instead (one MFMA pipe per SIMD: passes * 4 cycles per instruction)
        const double passes = KIND == 0 ? 8.0 : 16.0;
        const double per_simd = (double)iters * 8 * NACC * blocks_per_cu;            // MFMAs through one SIMD's pipe
        const double ghz = per_simd * passes * 4.0 / (ms * 1e-3) / 1e9;
        printf("%-28s waves/SIMD %d  acc chains %d : %8.3f ms  %8.1f TFLOP/s  (implied shader clock if the pipe never idles: %.2f GHz)\n",
               name, blocks_per_cu, NACC, ms, tf, ghz);
    }
    CK(hipFree(out)); CK(hipFree(clk));
}

int main() {
    CK(hipSetDevice(0));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    printf("%d CUs, clockRate %d kHz\n", cus, p.clockRate);
    run<0, 4>("v_mfma_f32_32x32x16_bf16", 1, cus, 2.0 * 32 * 32 * 16);
    run<0, 4>("v_mfma_f32_32x32x16_bf16", 2, cus, 2.0 * 32 * 32 * 16);
    run<0, 1>("v_mfma_f32_32x32x16_bf16", 2, cus, 2.0 * 32 * 32 * 16);
    run<1, 4>("v_mfma_f32_32x32x2_f32", 1, cus, 2.0 * 32 * 32 * 2);
    run<1, 4>("v_mfma_f32_32x32x2_f32", 2, cus, 2.0 * 32 * 32 * 2);
    run<1, 1>("v_mfma_f32_32x32x2_f32", 4, cus, 2.0 * 32 * 32 * 2);
    return 0;
}
