// Microbenchmark (measurement tool, not product code): cost of a device-wide barrier inside one persistent kernel on MI355X
// (256 CUs / 8 XCDs, one L2 per XCD), and of one "stage" = every wave stores a 2 KB tile, barrier, every wave loads 32 KB that
// OTHER blocks wrote (checks cross-XCD visibility).  Decides whether a persistent small-batch evaluator can beat ~5.4 us per
// dependent kernel launch.   hipcc --offload-arch=gfx950 -O3 gridsync_bench.hip -o gridsync_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

constexpr long SPIN_LIMIT = 4000000;   // bounded spin: a lost barrier ends the kernel with an error flag instead of hanging the GPU

__device__ __forceinline__ bool wait_ge(unsigned* p, unsigned target, int* err) {
    long spins = 0;
    while (__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (++spins > SPIN_LIMIT) { *err = 1; return false; }
        __builtin_amdgcn_s_sleep(1);
    }
    return true;
}

// flat: one counter, every block adds 1
__device__ __forceinline__ void barrier_flat(unsigned* ctr, unsigned& target, unsigned nblocks, int* err) {
    __syncthreads();
    if (threadIdx.x == 0) {
        target += nblocks;
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        wait_ge(ctr, target, err);
    }
    __syncthreads();
}

// hierarchical: group g = blockIdx % 8 (the XCD a block lands on under round-robin dispatch) has its own arrival counter; the last
// arriver of a group adds 1 to the global counter; everybody polls the global counter
__device__ __forceinline__ void barrier_hier(unsigned* ctrs, unsigned& gen, unsigned nblocks, int* err) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned g = blockIdx.x & 7, per = (nblocks + 7 - g) / 8;
        gen += 1;
        const unsigned old = __hip_atomic_fetch_add(ctrs + 32 * (1 + g), 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == gen * per) __hip_atomic_fetch_add(ctrs, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        wait_ge(ctrs, gen * 8, err);
    }
    __syncthreads();
}

template <int MODE>
__global__ __launch_bounds__(256) void barrier_only(unsigned* ctrs, int iters, int* err) {
    unsigned t = 0;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) barrier_flat(ctrs, t, gridDim.x, err); else barrier_hier(ctrs, t, gridDim.x, err);
        if (*(volatile int*)err) return;
    }
}

// stage: wave w of block b writes 512 floats (2 KB) of buffer[i&1] = f(i, global wave id); barrier; reads 8192 floats (32 KB) written by
// the 16 waves following it cyclically (other blocks, other XCDs) and checks them
template <int MODE>
__global__ __launch_bounds__(256) void stage_loop(unsigned* ctrs, float* buf0, float* buf1, int iters, int* err, unsigned* bad) {
    unsigned t = 0;
    const int lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
    unsigned nbad = 0;
    for (int i = 0; i < iters; ++i) {
        float* w = (i & 1) ? buf1 : buf0;
        float4 v; v.x = v.y = v.z = v.w = (float)(i * 7 + wave);
        reinterpret_cast<float4*>(w + (size_t)wave * 512)[lane] = v;
        reinterpret_cast<float4*>(w + (size_t)wave * 512)[lane + 64] = v;
        if (MODE == 0) barrier_flat(ctrs, t, gridDim.x, err); else barrier_hier(ctrs, t, gridDim.x, err);
        if (*(volatile int*)err) return;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int src = (wave + 37 * (k + 1)) % nw;
            const float4 a = reinterpret_cast<const float4*>(w + (size_t)src * 512)[lane];
            const float4 b = reinterpret_cast<const float4*>(w + (size_t)src * 512)[lane + 64];
            const float e = (float)(i * 7 + src);
            nbad += (a.x != e) + (a.w != e) + (b.y != e) + (b.z != e);
            acc += a.x + b.w;
        }
        if (acc == -1.f) bad[1] = 1;   // keep the loads
    }
    if (nbad) atomicAdd(bad, nbad);
}

int main() {
    int dev = 0; CK(hipSetDevice(dev));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, dev));
    printf("device %s, %d CUs\n", p.name, p.multiProcessorCount);
    unsigned* ctrs; int* err; unsigned* bad; float *b0, *b1;
    CK(hipMalloc(&ctrs, 4096)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&bad, 8));
    const int maxw = 1024 * 4;
    CK(hipMalloc(&b0, (size_t)maxw * 2048)); CK(hipMalloc(&b1, (size_t)maxw * 2048));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000;
    for (int nb : {64, 128, 256, 512}) {
        if (nb > p.multiProcessorCount * 2) continue;
        for (int mode = 0; mode < 2; ++mode) {
            for (int kind = 0; kind < 2; ++kind) {
                float best = 1e30f; int herr = 0; unsigned hbad[2] = {0, 0};
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipMemset(ctrs, 0, 4096)); CK(hipMemset(err, 0, 4)); CK(hipMemset(bad, 0, 8));
                    CK(hipEventRecord(e0));
                    if (kind == 0) {
                        if (mode == 0) hipLaunchKernelGGL(barrier_only<0>, dim3(nb), dim3(256), 0, 0, ctrs, iters, err);
                        else hipLaunchKernelGGL(barrier_only<1>, dim3(nb), dim3(256), 0, 0, ctrs, iters, err);
                    } else {
                        if (mode == 0) hipLaunchKernelGGL(stage_loop<0>, dim3(nb), dim3(256), 0, 0, ctrs, b0, b1, iters, err, bad);
                        else hipLaunchKernelGGL(stage_loop<1>, dim3(nb), dim3(256), 0, 0, ctrs, b0, b1, iters, err, bad);
                    }
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best) best = ms;
                    CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hbad, bad, 8, hipMemcpyDeviceToHost));
                    if (herr) break;
                }
                printf("blocks %4d  %s  %s : %7.3f us / iteration   spin_timeout=%d  stale_reads=%u\n", nb, mode ? "hier" : "flat",
                       kind ? "store+barrier+load32KB" : "barrier only          ", best * 1000.f / iters, herr, hbad[0]);
                fflush(stdout);
            }
        }
    }
    // reference point: the same number of dependent trivial kernel launches
    {
        CK(hipMemset(err, 0, 4));
        hipStream_t s; CK(hipStreamCreate(&s));
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(barrier_only<0>, dim3(256), dim3(256), 0, s, ctrs, 0, err);
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < 10; ++r) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("hipGraph of 200 dependent empty 256-block kernels: %7.3f us / kernel\n", ms * 1000.f / 2000);
    }
    return 0;
}
