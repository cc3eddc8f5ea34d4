// Sampling-loop driver (host) — see sampler.hip for the reference citations.
#pragma once
#include <string>
#include <vector>

#include "denoiser.h"

namespace dsh {

struct DiffusionTables {
    std::vector<double> betas, ac, ac_prev, c1, c2, post_var, post_logvar, coef1, coef2;
    std::vector<int> tmap;   // spaced index -> original timestep
};
void build_tables(const std::vector<double>& betas, DiffusionTables& t);
std::vector<double> linear_betas(int n);
int make_tables(int steps, int respacing, DiffusionTables& out, std::string& err);
std::vector<int> jump_schedule(int respacing, int jump_length, int jump_n_sample);

struct SamplerOpts {
    int kind = 0, diffusion_steps = 1000, respacing = 25, jump_length = 3, jump_n_sample = 5, overlap_len = 10,
        add_blend = 1, no_resample = 0, no_repaint = 0, clip_denoised = 0, noise_mode = 0;
    uint64_t seed = 0;
    int same_overlap_noisy = 0, clip_idx = 0;     // gaussian_diffusion.py:1040-1060: window index inside a chain
    float eta = 0.f;                              // DDIM eta (gaussian_diffusion.py:1011-1032)
};
enum StepKind { STEP_DDIM = 0, STEP_UNDO = 1, STEP_DDPM = 2 };
struct SamplerStep { StepKind kind; int level; };

int64_t sampler_num_draws(const SamplerOpts& o, bool masked, bool init_from_x);
int64_t sampler_num_steps(const SamplerOpts& o, bool masked);

class Sampler {
  public:
    Sampler(hipStream_t s, int channels_) : st(s), channels(channels_) {}
    ~Sampler();
    Profiler* prof = nullptr;
    // Philox mode: per-row keys (one per batch row; empty = one stream for the whole batch).  A chain keyed by its
    // global id draws the same noise whatever batch / rank it is sampled in (sharded long-audio path).
    int set_row_keys(const uint64_t* keys_host, int n);
    int run(DenoiserBase* den, const SamplerOpts& o, float* x, bool init_from_x, const float* gt, const uint8_t* mask,
            bool masked, const float* noise_stack, int64_t n_draws, float* trace);

  private:
    int ensure(size_t n, int B);
    hipStream_t st;
    int channels;
    std::vector<void*> bufs;
    size_t cap_n = 0; int cap_b = 0;
    float *eps = nullptr, *nz1 = nullptr, *c1buf = nullptr, *c2buf = nullptr;
    float* nz_eta = nullptr; size_t cap_eta = 0;      // Philox scratch of the step's own randn_like (eta != 0 only)
    int64_t* tbuf = nullptr; int64_t* lvlbuf = nullptr;
    DiffusionTables tb; int tb_steps = -1, tb_resp = -1;
    uint64_t* row_keys = nullptr; int n_row_keys = 0, cap_row_keys = 0;
    // --same_overlap_noisy: the noisy tail x[..., -L:, :] saved after every DDIM step, one slot per spaced level; persists
    // across sample() calls like the reference's self.saved_noisy_tail (the dict the next window receives IS this object)
    float* tails = nullptr; float* tail_tmp = nullptr; size_t tails_blc = 0; int tails_levels = 0;
    // hipGraph replay of one denoiser evaluation for launch-bound (small-batch / window-chain) runs
    // (one graph per timestep-cache mode, denoiser.h: 0 plain, 1 compute + save level, 2 restore level)
    hipGraphExec_t graph_exec[3] = {nullptr, nullptr, nullptr};
    hipGraph_t graph[3] = {nullptr, nullptr, nullptr};
    void drop_graph();
    int eval_step(DenoiserBase* den, float* x, int n_eval, bool use_graph, int mode);
    // pipelined small-batch loop (denoiser.h: set_part / pipe_begin): the gesture encoder's chain on its own stream, one step behind the
    // expression encoder's — its own per-step scalars, noise scratch and evaluation graph
    float *nz1G = nullptr, *nz_etaG = nullptr, *c1bufG = nullptr, *c2bufG = nullptr;
    int64_t* tbufG = nullptr; int64_t* lvlbufG = nullptr; size_t capG_n = 0;
    hipGraphExec_t graph_execG = nullptr; hipGraph_t graphG = nullptr;
    hipEvent_t ev_pE = nullptr, ev_pC = nullptr, ev_pG = nullptr;       // E_k done / E_k's expression estimate copied / gesture chain done
    int eval_step_twin(DenoiserBase* twin, hipStream_t s, float* x, int n_eval, bool use_graph, int mode);
    // free-running sub-batch streams of large batches (run(): one fork before the loop, one join after it)
    hipEvent_t ev_fork = nullptr;
    std::vector<hipEvent_t> ev_sub;      // [2 i] = "sub-batch i has queued its first launches" (stagger), [2 i + 1] = "sub-batch i done"
};

}  // namespace dsh
