// Optional per-kernel-class timing with HIP events on the context stream (bench.py roofline leg).
// Disabled by default: zero events are recorded unless dsh_profile_enable(ctx, 1) was called.
#pragma once
#include <stdlib.h>
#include <hip/hip_runtime.h>

#include <vector>

namespace dsh {

// 0-3: kernel families; 4-9: the token-per-lane Linear instantiations the denoiser launches
enum ProfClass : int {
    PROF_GEMM = 0, PROF_ATTN = 1, PROF_ROWOPS = 2, PROF_SAMPLER = 3,
    PROF_TL_QKV = 4, PROF_TL_STY = 5, PROF_TL_FFN1 = 6, PROF_TL_FFN2 = 7, PROF_TL_FEAT1 = 8, PROF_TL_FEAT3 = 9, PROF_RESERVED10 = 10, PROF_TL_FFN = 11,
    PROF_NCLASS = 16
};

// kernel (rocprofv3 name without the trailing ablation argument) and role of every class; "" = class unused
struct ProfClassInfo { const char* kernel; const char* role; };
inline const ProfClassInfo& prof_class_info(int cls, bool fp32) {
    static const ProfClassInfo none = {"", ""};
    // (default dispatch of the bf16 path: LDS-DMA kernels (tl2.hip) for the MFMA-bound instantiations, first-generation
    //  tl_linear.hip for the two HBM-bound ones; names as rocprofv3 prints them, minus a trailing default ablation argument)
    static const ProfClassInfo tab[PROF_NCLASS] = {
        {"gemm_nt_kernel<dsh::bf16, 1, MI, NJ>", "small GEMMs (joint_embed, audio_proj, hubert conv, encoder_aud; embeddings / FiLM above 16 rows)"},
        {"linear_attention_tiled_kernel", "linear self-attention core"},
        {"row kernels", "layout edges / LayerNorm rows"},
        {"sampler kernels", "ddim / ddpm / undo updates, Philox"},
        {"tl2_linear_kernel<512, 1, false, 2, 0, false>", "sa_block LayerNorm (folded into W) + q|k|v"},
        {"tl_linear_kernel<512, 2, true, 3, 0>", "StylizationBlock (LN+FiLM+SiLU) Linear + residual"},
        {"tl2_linear_kernel<512, 0, false, 2, 2, false>", "ffn.linear1 + GELU (unfused path)"},
        {"tl2_linear_kernel<1024, 0, false, 2, 0, false>", "ffn.linear2 (unfused path)"},
        {"tl2_linear_kernel<1024, 3, false, 2, 1, false>", "feat_proj concat + LayerNorm (folded) + Linear + SiLU"},
        {"tl_linear_kernel<1024, 0, true, 3, 0>", "feat_proj.3 + residual"},
        none,   // (slot of the round-1 chained kernel, removed: superseded by the fused FFN)
        {"tl2_ffn_kernel<false>", "ffn.linear1 -> GELU -> ffn.linear2 -> StylizationBlock(ffn) -> + h (one launch)"},
        none, none, none, none};
    static const ProfClassInfo gemm32 = {"gemm_f32_pro_kernel<PRO> + gemm_nt_kernel<float, 1, MI, NJ>", "fp32 path: every Linear (exact-fp32 v_mfma_f32_32x32x2_f32, 64 x 64 tiles; round 6: the large launches on the software-pipelined LDS-DMA loop of gemm_f32_pro.hip, LayerNorms folded in, the FFN branch's StylizationBlock front in the operand staging)"};
    static const ProfClassInfo attn32 = {"linear_attention_f32_mfma(_sty)_kernel<TM>", "fp32 path: linear self-attention core on the exact-fp32 matrix pipe (round 6; with the attention branch's StylizationBlock front behind it for windows of up to 64 frames)"};
    static const ProfClassInfo ffn3 = {"tl3_ffn_kernel<false>", "ffn.linear1 -> GELU -> ffn.linear2 -> StylizationBlock(ffn) -> + h (one launch; tl3_ffn.hip)"};
    if (cls < 0 || cls >= PROF_NCLASS) return none;
    if (cls == PROF_TL_FFN) { const char* fv = getenv("DSH_FFN_V"); if (!(fv && atoi(fv) == 2)) return ffn3; }   // (as Denoiser reads it)
    if (fp32 && cls == PROF_ATTN) return attn32;
    return (fp32 && cls == PROF_GEMM) ? gemm32 : tab[cls];
}

struct Profiler {
    bool on = false;
    hipStream_t st = nullptr;
    struct Rec { hipEvent_t a, b; int cls; };
    std::vector<Rec> pool;
    size_t used = 0;
    double flops[PROF_NCLASS] = {};
    double bytes[PROF_NCLASS] = {};   // algorithmic HBM bytes (operands read once + results written once)

    ~Profiler() { for (auto& r : pool) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); } }
    void reset() { used = 0; for (double& f : flops) f = 0; for (double& f : bytes) f = 0; }
    void begin(int cls) {
        if (!on) return;
        if (used == pool.size()) {
            Rec r; r.cls = cls;
            (void)hipEventCreate(&r.a); (void)hipEventCreate(&r.b);
            pool.push_back(r);
        }
        pool[used].cls = cls;
        (void)hipEventRecord(pool[used].a, st);
    }
    void end(double fl = 0.0, double by = 0.0) {
        if (!on) return;
        (void)hipEventRecord(pool[used].b, st);
        flops[pool[used].cls] += fl;
        bytes[pool[used].cls] += by;
        ++used;
    }
    // ms / launches per class; synchronises the stream
    void read(double* ms, long long* launches) {
        (void)hipStreamSynchronize(st);
        for (int c = 0; c < PROF_NCLASS; ++c) { ms[c] = 0; launches[c] = 0; }
        for (size_t i = 0; i < used; ++i) {
            float t = 0.f;
            (void)hipEventElapsedTime(&t, pool[i].a, pool[i].b);
            ms[pool[i].cls] += t;
            launches[pool[i].cls] += 1;
        }
    }
};

}  // namespace dsh
