// Device-side UniDiffuser denoiser (models/transformer.py:590-770) for one (device, stream).
#pragma once
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "dsh_common.h"
#include "dsh_kernels.h"
#include "profiler.h"

namespace dsh {

struct ModelConfig {
    int dim_pose = 129, expression_dim = 103, style_dim = 4;
    int classifier_free = 1;
    float cond_scale = 1.25f;
    int latent_dim = 512, ff_size = 1024, num_layers = 8, num_heads = 8;
    int audio_dim = 128, aud_latent_dim = 256, hubert_dim = 1024, hubert_enc_dim = 128;
    int precision = 0;   // 0 = fp32 (exact-fp32 MFMA), 1 = bf16 storage + bf16 MFMA, fp32 accumulate
    int cfg_active() const { return classifier_free && cond_scale != 1.0f; }
    int channels() const { return dim_pose + expression_dim; }
    int time_embed_dim() const { return 4 * latent_dim; }
};

struct HostTensor {
    std::vector<int64_t> shape;
    std::vector<float> data;
    size_t numel() const { return data.size(); }
};

// Type-erased interface; the implementation is templated on the storage type (float / bf16).
class DenoiserBase {
  public:
    virtual ~DenoiserBase() {}
    virtual int finalize(const std::map<std::string, HostTensor>& w) = 0;
    // step-invariant conditioning: audio [B,T,128] fp32, person_id [B,S] fp32, hubert [B,T,1024] fp32 (device)
    virtual int set_condition(int B, int T, const float* audio, const float* person_id, const float* hubert) = 0;
    // eps[B,T,C] = model(x[B,T,C], t[B]) with c1/c2 [B] (device fp32) feeding the expression x0
    virtual int eval(const float* x, const int64_t* t, const float* c1, const float* c2, float* eps) = 0;
    virtual double issued_flops_per_eval() const = 0;   // MFMA GEMM flops actually launched for the current (B,T)
    virtual size_t weight_bytes() const = 0;
    // debug taps (device -> caller device buffer, fp32): "aud_feat" [B,T,128], "expr_x0" [B,T,E]
    virtual int debug_copy(const std::string& what, float* out) = 0;
    // second instance sharing the finalized weights, working on another stream (null if not supported / not finalized)
    virtual DenoiserBase* clone_shared(hipStream_t) { return nullptr; }
    // record `ev` on this instance's stream after the n-th token-per-lane launch of every eval (phase offset of a twin)
    virtual void notify_after_launches(hipEvent_t, int) {}
    int batch = 0, frames = 0;
    Profiler* prof = nullptr;   // owned by the context; may be null
};

DenoiserBase* make_denoiser(const ModelConfig& cfg, hipStream_t stream);

}  // namespace dsh
