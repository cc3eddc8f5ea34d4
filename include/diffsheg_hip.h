/*
 * diffsheg_hip.h — C ABI of the MI355X-native DiffSHEG sampling hot path (libdiffsheg_hip.so).
 *
 * The reference is pure Python/PyTorch; it has no FFI for this path.  The entry points below are
 * what a binding for the three in-process boundaries of the reference would call:
 *
 *   boundary 1  model(x, ts, **model_kwargs)          models/gaussian_diffusion.py:536
 *               (through _WrappedModel.__call__,       models/respace.py:119-124;
 *                UniDiffuser.forward                   models/transformer.py:728-770)   -> dsh_eval
 *   boundary 2  ddim_sample_loop / p_sample_loop       models/gaussian_diffusion.py:1106-1159, :776-841
 *               (SpacedDiffusion tables                models/respace.py:68-82,
 *                jump schedule                         models/scheduler.py:178-208)     -> dsh_sample
 *   weights     UniDiffuser.state_dict() key names     trainers/ddpm_show_trainer.py:259-292 -> dsh_load_tensor
 *
 * Conventions: every pointer marked "device" is caller-owned HIP device memory on the context's
 * device; tensors are dense row-major float32 [B,T,channels] unless stated; timesteps are int64.
 * Functions return 0 on success, negative on error (-1 invalid argument, -2 HIP runtime error,
 * -3 a C++ exception such as std::bad_alloc caught at the boundary); dsh_last_error() returns a
 * thread-local description.  No exceptions cross this boundary.  A
 * context is bound to one (device, stream) and is not thread-safe; distinct contexts are
 * independent (one process per GPU, as runner.py:86 mp.spawn does).
 */
#ifndef DIFFSHEG_HIP_H
#define DIFFSHEG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dsh_ctx dsh_ctx;

enum { DSH_PRECISION_FP32 = 0, DSH_PRECISION_BF16 = 1 };
enum { DSH_SAMPLER_DDIM = 0, DSH_SAMPLER_DDPM = 1 };
enum { DSH_NOISE_STACK = 0, DSH_NOISE_PHILOX = 1 };

/* Static model configuration: the attributes UniDiffuser reads off the reference's `opt`
 * namespace (runner.py:124-222, options/base_options.py), baked at creation. */
typedef struct dsh_model_config {
    int32_t dim_pose;        /* gesture channels (129 SHOW / 141 BEAT)   */
    int32_t expression_dim;  /* expression channels (103 / 51)           */
    int32_t style_dim;       /* one-hot speaker width (4 / 30)           */
    int32_t classifier_free; /* model has null_cond_emb                  */
    float   cond_scale;      /* CFG scale; != 1 doubles the batch inside */
    int32_t latent_dim;      /* 512  */
    int32_t ff_size;         /* 1024 */
    int32_t num_layers;      /* 8    */
    int32_t num_heads;       /* 8    */
    int32_t audio_dim;       /* 128  */
    int32_t aud_latent_dim;  /* 256  */
    int32_t hubert_dim;      /* 1024 */
    int32_t hubert_enc_dim;  /* 128  */
    int32_t precision;       /* DSH_PRECISION_*                          */
    int32_t single_transformer; /* 0: UniDiffuser (encoder_aud + encoder_exp -> encoder_ges), state-dict keys as saved by  */
                             /* runner.py:32-45.  1: ONE MotionTransformer over all dim_pose + expression_dim channels     */
                             /* (runner.py:46-57, opt.unidiffuser = False, model_base 'transformer_encoder'): keys without */
                             /* the encoder_* prefix, audio_proj on the 128 mel features; c1 / c2 of dsh_eval are unused   */
} dsh_model_config;

/* Sampler options = the `opt` attributes read by gaussian_diffusion.py / respace.py / scheduler.py. */
typedef struct dsh_sampler_opts {
    int32_t kind;            /* DSH_SAMPLER_DDIM (spaced) or DSH_SAMPLER_DDPM (ancestral)            */
    int32_t diffusion_steps; /* 1000                                                                  */
    int32_t respacing;       /* K of 'ddimK' (25); ignored for DDPM                                   */
    int32_t jump_length;     /* RePaint jump schedule (options/base_options.py:127-128)               */
    int32_t jump_n_sample;
    int32_t overlap_len;     /* frames cross-faded when add_blend (gaussian_diffusion.py:1051-1054)   */
    int32_t add_blend;
    int32_t no_resample;     /* jump schedule with jump_length = jump_n_sample = 1                    */
    int32_t no_repaint;      /* plain 25-step loop even when a mask is present                        */
    int32_t clip_denoised;   /* clamp pred_xstart to [-1,1] (gaussian_diffusion.py:575-580); harness: 0 */
    int32_t noise_mode;      /* DSH_NOISE_STACK: consume caller-provided draws in reference order;    */
                             /* DSH_NOISE_PHILOX: on-device Philox4x32-10 + Box-Muller                */
    uint64_t seed;           /* Philox key                                                            */
    int32_t same_overlap_noisy; /* --same_overlap_noisy (gaussian_diffusion.py:1040-1060): out-painted frames take the  */
                             /* previous window's noisy tail of the same level, saved in the context after every DDIM   */
                             /* step (the reference's self.saved_noisy_tail); DDIM only                                 */
    int32_t clip_idx;        /* window index inside the chain (model_kwargs['y']['clip_idx']); 0 = first window         */
    float eta;               /* DDIM eta (gaussian_diffusion.py:985,1011-1032); the harness uses 0                       */
} dsh_sampler_opts;

const char* dsh_last_error(void);
const char* dsh_version(void);

/* ---- lifecycle ------------------------------------------------------------------------------ */
/* `hip_stream` is a hipStream_t (NULL = default stream) on the current device. */
int dsh_create(const dsh_model_config* cfg, void* hip_stream, dsh_ctx** out);
int dsh_destroy(dsh_ctx* ctx);

/* ---- weights (state-dict key names of UniDiffuser; fp32 host memory, copied) ---------------- */
int dsh_load_tensor(dsh_ctx* ctx, const char* name, const float* host_data, const int64_t* shape, int32_t ndim);
/* Builds the device-side layout (fused qkv, stacked FiLM, folded BatchNorm, K-padded tiles,
 * feat_proj(null_cond_emb) constants) and releases the host staging copies. */
int dsh_finalize_weights(dsh_ctx* ctx);
int64_t dsh_weight_bytes(const dsh_ctx* ctx);

/* ---- boundary 1: one denoiser evaluation ------------------------------------------------------ */
/* Step-invariant conditioning (model_kwargs audio_emb / person_id / add_cond['pretrain_aud_feat']):
 *   audio_emb [B,T,audio_dim], person_id [B,style_dim], hubert [B,T,hubert_dim]; all device fp32.
 * Runs hubert_encoder and pid_embed once; must be called before dsh_eval / dsh_sample and again
 * whenever the conditioning or (B,T) changes.  The three tensors are copied into context-owned
 * buffers in stream order: they may be freed / overwritten as soon as the call has returned,
 * provided that happens on (or is ordered after) the context stream. */
int dsh_set_condition(dsh_ctx* ctx, int32_t batch, int32_t frames, const float* audio_emb, const float* person_id,
                      const float* hubert);
/* eps[B,T,C] = UniDiffuser(x[B,T,C], t[B]; sqrt_alphas = (c1[B], c2[B])).  t holds ORIGINAL-scale
 * timesteps (what _WrappedModel passes); c1/c2 are sqrt(1/abar_t), sqrt(1/abar_t - 1) per sample
 * (gaussian_diffusion.py:527-532).  All device pointers; asynchronous on the context stream. */
int dsh_eval(dsh_ctx* ctx, const float* x, const int64_t* t, const float* c1, const float* c2, float* eps);
/* GEMM + attention flops actually launched by the last dsh_eval (work skipped is not counted). */
double dsh_eval_flops(const dsh_ctx* ctx);
/* Per-kernel-class HIP-event timing on the context stream (bench.py roofline leg).  enable=1 resets and
 * starts recording; dsh_profile_read synchronises and returns, per class, the summed milliseconds, launch
 * counts, algorithmic flops and algorithmic HBM bytes; every output array has 16 entries.
 * dsh_profile_class_info names class `cls`: the kernel as rocprofv3 prints it (empty string = unused class)
 * and its role in the denoiser (static strings). */
int dsh_profile_enable(dsh_ctx* ctx, int32_t enable);
int dsh_profile_read(dsh_ctx* ctx, double* ms16, int64_t* launches16, double* flops16, double* bytes16);
int dsh_profile_class_info(const dsh_ctx* ctx, int32_t cls, const char** kernel, const char** role);
/* debug taps after dsh_eval: "aud_feat" [B,T,audio_dim], "expr_x0" [B,T,expression_dim] (device out). */
int dsh_debug_copy(dsh_ctx* ctx, const char* what, float* out);

/* ---- boundary 2: full sampling loops ----------------------------------------------------------- */
/* Number of Gaussian tensors [B,T,C] the loop consumes, in the reference's draw order
 * (SURVEY §8a S7): x_T first (unless init_from_x), then per step.  Returns < 0 on error. */
int64_t dsh_sample_num_draws(const dsh_sampler_opts* opts, int32_t masked, int32_t init_from_x);
/* Number of (denoise + undo) steps, i.e. rows a trace buffer needs. */
int64_t dsh_sample_num_steps(const dsh_sampler_opts* opts, int32_t masked);
/* x[B,T,C] (device, in/out): start sample if init_from_x, final sample on return.
 * gt / mask (device, [B,T,C] fp32 / uint8) may be NULL; `masked` is the caller's
 * `True in outpainting_mask` (gaussian_diffusion.py:1126).  noise_stack: device fp32
 * [n_draws, B*T*C] for DSH_NOISE_STACK, else NULL.  trace (device, nullable): [n_steps, B*T*C],
 * receives the sample after every step.  Asynchronous on the context stream.  (Large batches are sampled as two or three
 * independent sub-batches, each running the whole loop on an internal stream forked from and joined to the context stream:
 * for the caller everything stays ordered on the context stream, and the result is bit-identical to one stream.) */
int dsh_sample(dsh_ctx* ctx, const dsh_sampler_opts* opts, float* x, int32_t init_from_x, const float* gt,
               const uint8_t* mask, int32_t masked, const float* noise_stack, int64_t n_draws, float* trace);
/* DSH_NOISE_PHILOX only: give every batch row its own generator key (host array of n = B entries; n = 0 restores
 * the single whole-batch stream).  Row b then draws from key `seed` with the 128-bit Philox counter
 * (draw index and position inside the row, keys[b]): distinct (seed, key) pairs never share a stream, and the
 * counters depend only on the draw index and the position inside the row, so a chain identified by a global id
 * receives the same noise
 * whatever batch, stream split or rank it is sampled in (sharded test_arbitrary_len, ddpm_show_trainer.py:743-750:
 * the reference instead draws from each rank's global torch RNG).  Sticky until changed; frames*channels % 4 == 0. */
int dsh_sample_set_row_keys(dsh_ctx* ctx, const uint64_t* keys_host, int32_t n);

/* ---- schedule / table introspection (host; parity tests for S1-S3) ---------------------------- */
/* name in {betas, alphas_cumprod, alphas_cumprod_prev, sqrt_recip_alphas_cumprod,
 * sqrt_recipm1_alphas_cumprod, posterior_variance, posterior_log_variance_clipped,
 * posterior_mean_coef1, posterior_mean_coef2}; respacing 0 = full chain.  Returns entries written. */
int32_t dsh_diffusion_table(int32_t diffusion_steps, int32_t respacing, const char* name, double* out, int32_t cap);
int32_t dsh_timestep_map(int32_t diffusion_steps, int32_t respacing, int32_t* out, int32_t cap);
int32_t dsh_jump_schedule(int32_t respacing, int32_t jump_length, int32_t jump_n_sample, int32_t* out, int32_t cap);

/* ---- rows either side of the path ---------------------------------------------------- */
/* y[B,frames_out,C] = F.interpolate(x^T, size=frames_out, mode='linear', align_corners=True)^T of x[B,frames_in,C]:
 * HuBERT hidden states resampled to the pose frame rate (datasets/show.py:98, ddpm_show_trainer.py:1082). */
int dsh_interp_time(void* hip_stream, const float* x, int32_t batch, int32_t frames_in, int32_t channels, float* y,
                    int32_t frames_out);
/* y = x * std[c] + mean[c] over n contiguous fp32 values with `channels` innermost (datasets/show.py:157-162). */
int dsh_inv_standardize(void* hip_stream, const float* x, int64_t n, int32_t channels, const float* mean, const float* stdv,
                        float* y);

/* ---- unit kernels (device pointers; used by the kernel-level parity tests) -------------------- */
/* C[M,N] = act(A[M,K] W[N,K]^T + bias) (+ R); dtype 0: fp32 operands, 1: bf16 operands (uint16 bits).
 * K must be a multiple of 32 (fp32) / 64 (bf16).  Cf: fp32 out (nullable); Ct: operand-typed out (nullable). */
int dsh_op_gemm(void* hip_stream, int32_t dtype, const void* A, const void* W, const float* bias, const float* R,
                float* Cf, void* Ct, int32_t M, int32_t N, int32_t K, int32_t act);
/* fp32 parity path: a Linear with what the reference computes in front of it, in ONE launch (gemm_f32_pro.hip).
 *   pro 1: C = act(LayerNorm(concat(x0 .. x3)) W^T + b) with the LayerNorm affine FOLDED by the caller: W = gamma (.) W_ref, bias = b + W_ref beta,
 *          fc[n] = sum_k W[n][k]  (models/transformer.py:106-108,119-125 q|k|v; :284-289,304-312 feat_proj.0/.1).  Segment j is fp32 [M, ld_j], w_j
 *          columns wide (multiples of 32; zero padded where the real width is smaller); sum w_j = K; LayerNorm over the first k_real columns.
 *   pro 2: C = Linear(SiLU(LN(x0) scale' + shift')) + R, the StylizationBlock (models/transformer.py:86-97), film [nb, >= film_off + 2K] holding
 *          (scale' | shift') = (gamma (1 + scale) | beta (1 + scale) + shift) per sample, sample = (row / frames) % nb; K = ld-free width of x0.
 *   pro 0: C = act(x0 W^T + b) (+ R): the same software-pipelined main loop without a front.
 * Row moments travel between launches instead of being recomputed: stats_out (nullable, N % 64 == 0, N <= 512) receives [M][N / 32] pairs
 * (mean_g, sum (c - mean_g)^2) over groups of 32 output columns; stats (pro 2, nullable) takes such pairs for the INPUT rows, stat_groups groups
 * of K / stat_groups columns each, combined in a fixed order — the launch then makes no pass over its rows for the LayerNorm.
 * All pointers device fp32; W [N, K] row-major; R nullable [M, N]; C [M, N]. */
int dsh_op_gemm_f32_pro(void* hip_stream, int32_t pro, const float* x0, int32_t ld0, int32_t w0, const float* x1, int32_t ld1, int32_t w1,
                        const float* x2, int32_t ld2, int32_t w2, const float* x3, int32_t ld3, int32_t w3, int32_t k_real, const float* W,
                        const float* bias, const float* fc, const float* film, int32_t film_ld, int32_t film_off, int32_t frames, int32_t nb,
                        const float* R, float* C, int32_t M, int32_t N, int32_t act, const float* stats, int32_t stat_groups, float* stats_out);
/* Token-per-lane fused Linear (bf16, K = 512 or 1024): out = act(prologue(X) W^T + bias) (+ R).  X bf16 [M,K]
 * with M padded to a multiple of 128 rows, W bf16 [N,K] in natural k order (permuted internally into a scratch
 * copy), pro 0 plain / 1 LayerNorm / 2 LayerNorm+FiLM+SiLU with film [nb, 2K] = (scale | shift) per sample,
 * sample = (row / frames) % nb; pro 3 (K = 1024): X is the row-major concat row [h 512 | audio_proj 256 | hubert 128 | expr 128]
 * of feat_proj.0 (transformer.py:304-312), LayerNorm over its first `frames` (= real width, 896 .. 1024) columns, gamma / beta
 * [1024] zero beyond them. */
/* Test helper: which kernel family the LAST token-per-lane Linear launch of this process selected — 0 = tl2 round-2 loop, 1 = tl2 rolling
 * loop, 2 = tl2 rolling loop on hi / lo residual planes, 3 = tl2 out-of-phase epilogues, 4 = tl4 LDS-tiled (8 waves), 5 = tl4 (4 waves),
 * 10 = tl_linear (first generation), 11 = tl_small (window chain); -1 before any launch.  Lets the bit-identity tests assert that the
 * kernel they mean to check is the one that ran. */
int32_t dsh_debug_last_tl_variant(void);
int dsh_op_tl_linear(void* hip_stream, int32_t pro, const void* X, const void* W, const float* bias, const float* R,
                     float* Cf, void* Ct, int32_t M, int32_t N, int32_t act, const float* gamma, const float* beta,
                     const float* film, int32_t frames, int32_t nb, int32_t K);
/* FFN branch of a decoder layer in one launch (bf16 path, latent 512 / ff 1024; models/transformer.py:169-181, :86-97):
 *   Cf = Hres + Linear3(SiLU(LN(y2; gamma, beta) * (1 + scale) + shift)) (+ row_const on rows < n_const_rows),
 *   y2 = GELU(X W1^T + b1) W2^T + b2;  Ct = bf16(Cf).  X bf16 [M,512], Hres / Cf fp32 [M,512], W1 [1024,512], W2 [512,1024],
 * W3 [512,512] bf16 row-major (natural order), film [nb, 1024] = (scale | shift) per sample, sample = (row / frames) % nb. */
int dsh_op_tl2_ffn(void* hip_stream, const void* X, const float* Hres, const void* W1, const float* b1, const void* W2, const float* b2,
                   const void* W3, const float* b3, const float* gamma, const float* beta, const float* film, int32_t frames, int32_t nb,
                   const float* row_const, int32_t n_const_rows, float* Cf, void* Ct, int32_t M);
/* LinearTemporalCrossAttention (models/transformer.py:133-166; the `transformer_decoder` layer's ca_block): device fp32
 * weights under the module's own parameter names.  y = x + proj_out(softmax_ch(Wq LN(x)) (softmax_N(Wk tn(xf))^T Wv tn(xf)), emb),
 * x [B,T,D], xf [B,N,L], emb [B,E] (the raw embedding: proj_out.emb_layers applies SiLU first), y [B,T,D]. */
typedef struct dsh_cross_attn_weights {
    const float *norm_g, *norm_b;             /* norm            [D]      */
    const float *text_norm_g, *text_norm_b;   /* text_norm       [L]      */
    const float *wq, *bq;                     /* query           [D,D],[D] */
    const float *wk, *bk;                     /* key             [D,L],[D] */
    const float *wv, *bv;                     /* value           [D,L],[D] */
    const float *sty_norm_g, *sty_norm_b;     /* proj_out.norm   [D]      */
    const float *sty_emb_w, *sty_emb_b;       /* proj_out.emb_layers.1 [2D,E],[2D] */
    const float *sty_out_w, *sty_out_b;       /* proj_out.out_layers.2 [D,D],[D]   */
} dsh_cross_attn_weights;
int dsh_op_cross_attention(void* hip_stream, const dsh_cross_attn_weights* w, const float* x, const float* xf, const float* emb,
                           int32_t B, int32_t T, int32_t N, int32_t D, int32_t L, int32_t E, int32_t num_head, float* y);
/* y[nb,T,D] = linear attention core on qkv[nb,T,3D] (fp32), head_dim in {16,64}. */
int dsh_op_linear_attention(void* hip_stream, const float* qkv, int32_t nb, int32_t frames, int32_t D, int32_t head_dim,
                            float* y);
/* same on bf16 storage (uint16 bits in/out); head_dim 64 and frames <= 96 take the MFMA kernel. */
int dsh_op_linear_attention_bf16(void* hip_stream, const void* qkv, int32_t nb, int32_t frames, int32_t D, int32_t head_dim,
                                 void* y);
/* out = LayerNorm(x[M,D]) * gamma + beta */
int dsh_op_layernorm(void* hip_stream, const float* x, int32_t M, int32_t D, const float* gamma, const float* beta,
                     float* out);
/* standard normals from the on-device Philox generator */
int dsh_op_philox_randn(void* hip_stream, float* out, int64_t n, uint64_t seed, uint64_t offset);
/* the per-row streams dsh_sample draws from after dsh_sample_set_row_keys: out[rows, n_row] (device), row b = key `seed`,
 * counter (offset + position inside the row, row_keys_host[b]); n_row % 4 == 0.  Synchronises the stream. */
int dsh_op_philox_randn_rows(void* hip_stream, float* out, int32_t rows, int64_t n_row, uint64_t seed, uint64_t offset,
                             const uint64_t* row_keys_host);

#ifdef __cplusplus
}
#endif
#endif /* DIFFSHEG_HIP_H */
