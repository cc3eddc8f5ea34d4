// Device-side UniDiffuser denoiser (models/transformer.py:590-770) for one (device, stream).
#pragma once
#include <cstdlib>
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
    // 1: the model is ONE MotionTransformer over all dim_pose + expression_dim channels (runner.py:46-57, opt.unidiffuser =
    // False, model_base transformer_encoder): no encoder_aud, audio_proj on the 128 mel features, no expression -> gesture flow
    int single_transformer = 0;
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
    // Timestep-level cache.  In every sampling loop all rows of an evaluation share one timestep, and part of an evaluation
    // does not depend on x at all: the time / speaker / FiLM embeddings, encoder_aud and audio_proj are functions of
    // (condition, t) only (transformer.py:730-739, :555-559, :574).  Loops that revisit levels (the out-painting jump schedule:
    // 63 evaluations over 16 levels) compute that part once per level:
    //   level_cache_prepare(n): slots for n levels of the CURRENT condition; 0 if available (small batches only), else -1
    //   eval_level(mode, level): mode 0 = eval(); 1 = eval() and save the x-independent results into slot *level (device
    //   int64); 2 = restore them from slot *level instead of recomputing them.  Slots die with the next set_condition().
    //   mode 3 = compute and save the x-independent results only (no evaluation; x / c1 / c2 / eps may be null): what a second
    //   instance on a side stream runs AHEAD of the loop.
    //   level_prefetch(t_values, n_levels, order, n_order, begin): enqueue that side-stream computation for the levels in `order`
    //   (t_values[level] = model timestep).  begin = 1 starts a run (conditions the side instance; -1 if prefetching is not
    //   available: then use modes 1 / 2 inline), begin = 0 appends further levels of the same run — the sampler enqueues each
    //   level one evaluation ahead of its first use, so the host never queues more than one level in front of the main chain.
    //   Every evaluation of such a run uses mode 2 and calls level_wait(level) before the first use of a level.
    //   sub >= 0 (round 4): the same for sub-batch `sub` of a large batch (sub_get): its own side instance and side stream feed the
    //   sub-batch instance's slots, and that instance is then evaluated with mode 2 directly.
    virtual int level_prefetch(const int64_t* /*t_values_host*/, int /*n_levels*/, const int* /*order*/, int /*n_order*/, int /*begin*/, int /*sub*/ = -1) { return -1; }
    virtual int level_wait(int /*level*/, int /*sub*/ = -1) { return 0; }
    // abandon the side-stream run of sub-batch `sub` (or of the whole batch): the evaluating stream waits for whatever the side
    // stream has queued — it writes the cache slots an inline fallback would write too — and the run is marked over
    virtual int level_prefetch_cancel(int /*sub*/ = -1) { return 0; }
    // helpers of the prefetch path (implemented by the per-stream instances)
    virtual int set_condition_light(int /*B*/, int /*T*/, const float* /*audio*/, const float* /*person_id*/) { return -1; }
    virtual int level_slots(char** /*slots*/, size_t* /*stride*/, int* /*n*/) { return -1; }
    virtual int adopt_level_slots(char* /*slots*/, size_t /*stride*/, int /*n*/) { return -1; }
    virtual int level_cache_prepare(int /*n_levels*/) { return -1; }
    virtual int eval_level(const float* x, const int64_t* t, const float* c1, const float* c2, float* eps, int /*mode*/,
                           const int64_t* /*level*/) { return eval(x, t, c1, c2, eps); }
    virtual double issued_flops_per_eval() const = 0;   // MFMA GEMM flops actually launched for the current (B,T)
    virtual size_t weight_bytes() const = 0;
    // debug taps (device -> caller device buffer, fp32): "aud_feat" [B,T,128], "expr_x0" [B,T,E]
    virtual int debug_copy(const std::string& what, float* out) = 0;
    // Sub-batch streams.  Large batches are conditioned as several independent sub-batches, each with its own instance (shared
    // weights) and stream.  eval() forks / joins them around ONE evaluation; a sampling loop can do better: clips never interact,
    // so it drives every sub-batch through ALL its steps on that sub-batch's stream and joins once at the end (sampler.hip).
    // sub_count() is valid after set_condition(); sub_get(i): the instance, its stream, and its clips [first, first + n).
    virtual int sub_count() const { return 1; }
    virtual int sub_get(int /*i*/, DenoiserBase** /*inst*/, hipStream_t* /*stream*/, int* /*first_clip*/, int* /*n_clips*/) { return -1; }
    // Pipelined small-batch loop (round 6).  In the UniDiffuser the expression encoder sees only the expression channels of x, the gesture
    // encoder the gesture channels plus the expression encoder's x0 estimate of the SAME step (transformer.py:741-768), and every sampler update
    // is element-wise: the expression chain E_0 -> E_1 -> ... never waits for a gesture evaluation, only G_k waits for E_k.  A launch-bound
    // loop therefore runs the two encoders on two streams, G one step behind E: (n + 1) encoder times instead of 2 n.
    //   set_part(p): 0 whole evaluation, 1 expression encoder only, 2 gesture encoder only (both need eval_level mode 2: the x-independent
    //                head restored from the timestep cache, of which each part restores its own encoder's share)
    //   import_expr(src, s): copy src's expression x0 estimate (what the gesture encoder reads) into this instance, on stream s
    //   pipe_begin(&twin, &stream): the gesture-side twin of the whole-batch instance (shared weights, own workspace and stream,
    //                conditioned like it, reading its timestep-cache slots) and puts both into their part; -1 if not available
    //                (single transformer, split batch, no timestep cache, DSH_PIPE=0).  pipe_end(): both back to whole evaluations.
    //   level_wait_stream(level, s): as level_wait(), for the twin's stream
    virtual int set_part(int /*part*/) { return -1; }
    virtual int import_expr(DenoiserBase* /*src*/, hipStream_t /*s*/) { return -1; }
    virtual int pipe_begin(DenoiserBase** /*twin*/, hipStream_t* /*stream*/) { return -1; }
    virtual int pipe_end() { return 0; }
    virtual int level_wait_stream(int /*level*/, hipStream_t /*s*/) { return -1; }
    //   loop_begin(kind): called by the sampler before it asks for sub-batches.  A DDIM or DDPM loop (kind 0 / 1) of a batch below pipe_rows token rows
    //                (default 64 499: where the sub-batch split would use two streams) is conditioned as ONE batch — the two encoder chains
    //                on two streams beat two sub-batch streams (313 clips: 131.7 k -> 148.8 k frames/s) — and the next set_condition of the
    //                same shape skips the split too; plain evaluations (dsh_eval) keep the sub-batch streams.
    virtual int loop_begin(int /*kind*/) { return 0; }
    virtual int loop_end() { return 0; }
    virtual int gesture_channels() const { return -1; }          // channels [0, g) belong to the gesture encoder, [g, C) to the expression encoder
    // second instance sharing the finalized weights, working on another stream (null if not supported / not finalized)
    virtual DenoiserBase* clone_shared(hipStream_t) { return nullptr; }
    // record `ev` on this instance's stream after the n-th token-per-lane launch of every eval (phase offset of a twin)
    virtual void notify_after_launches(hipEvent_t, int) {}
    int batch = 0, frames = 0;
    // Hint for the next evaluations: every row of `t` holds the SAME timestep (true in every sampling loop: gaussian_diffusion.py:
    // 1125, :792 build t = [i] * batch).  The time / speaker / FiLM embedding Linears (transformer.py:446-457, :77) then run on the
    // DISTINCT (timestep, speaker) rows only — the distinct speakers found at set_condition() — and are expanded per clip.
    // dsh_eval() checks the caller's tensor; the sampler sets it.
    bool t_uniform = false;
    Profiler* prof = nullptr;   // owned by the context; may be null
};

// DSH_EMB_DEDUP=0 (A/B switch): the embedding Linears run on every clip's row as before round 6
inline bool emb_dedup_enabled() { const char* e = getenv("DSH_EMB_DEDUP"); return !(e && atoi(e) == 0); }

DenoiserBase* make_denoiser(const ModelConfig& cfg, hipStream_t stream);

}  // namespace dsh
