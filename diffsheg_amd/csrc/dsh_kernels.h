// Launcher declarations for the non-GEMM kernels (rowops.hip, attention.hip, sampler_kernels.hip).
#pragma once
#include "dsh_common.h"

namespace dsh {

// virtual concat row [h | audio_proj | hubert128 | expr_x0]  (transformer.py:304-312)
struct ConcatSegs {
    const float* p0; int ld0, w0;   // latent h, fp32
    const void* p1;  int ld1, w1;   // element type T
    const void* p2;  int ld2, w2;   // element type T
    const float* p3; int ld3, w3;   // fp32 (w3 may be 0)
};

// fp32 parity path: Linear with the LayerNorm (folded, pro 1) or StylizationBlock front (LN -> FiLM -> SiLU, pro 2) in front of it
// in one launch (gemm_f32_pro.hip).  All pointers fp32.
struct GemmProArgs {
    int pro;                            // 1: folded LayerNorm over up to four concat segments; 2: StylizationBlock front
    const float* seg[4]; int seg_ld[4]; // A rows: K tile kt (32 floats) comes from segment j with seg_end[j - 1] <= kt < seg_end[j], at column 32 (kt - seg_end[j - 1])
    int seg_end[4];                     //   (pro 2: seg[0] only)
    int k_real;                         // LayerNorm width (K - k_real trailing columns are zero padding)
    const float* W; int ldw;            // [N, K]  (pro 1: gamma folded in)
    const float* bias;                  // [N]     (pro 1: b + W beta)
    const float* fc;                    // [N]     pro 1: row sums of the folded weight
    const float* film; int film_ld, film_off, frames, bmod;   // pro 2: per-clip [scale'(K) | shift'(K)] with the LayerNorm affine folded in; clip = (row / frames) % bmod
    const float* R; int ldr;            // residual or null
    float* C; int ldc;                  // output
    int M, N, K, act;
    int nt_n, nt_m;                     // (launcher)
    // per-row group moments (mean_g, sum (x - mean_g)^2) over groups of consecutive columns, [M][groups] float2:
    const float* stats; int stat_groups, stat_gs;   // pro 2, nullable: moments of the INPUT rows left by its producer (no pass over the rows here)
    float* stats_out;                   // nullable: moments of the OUTPUT rows in groups of 32 columns ([M][N / 32] float2; N % 64 == 0)
    const float* row_const; int n_const_rows;   // epilogue: + row_const[n] on rows < n_const_rows, after the residual (the CFG-null constant of the next layer)
    int abl;                            // ablation bits (DSH_GP_ABL, bench only; results are garbage): 1 no global loads in the loop, 2 no LDS writes, 4 no fragment reads, 8 no MFMAs, 16 no barrier
};
int launch_gemm_f32_pro(const GemmProArgs& a, hipStream_t s);

template <typename T>
int launch_ln_rows(float* h, int ldh, int M, int D, const float* pre_add, int n_pre_rows, const float* gamma,
                   const float* beta, T* out, int ldo, hipStream_t s);
template <typename TI, typename T>
int launch_ln_film_silu_rows(const TI* y, int ldy, int M, int D, const float* gamma, const float* beta,
                             const float* film, int film_ld, int film_off, int frames, int bmod, T* out, int ldo,
                             hipStream_t s);
template <typename T>
int launch_concat_ln_rows(const ConcatSegs& sg, int M, const float* gamma, const float* beta, T* out, int ldo, int Ppad,
                          hipStream_t s);
template <typename TI, typename T>
int launch_im2col3_rows(const TI* x, int ldx, int B, int frames, int Cin, T* out, int ldo, hipStream_t s);
template <typename T>
int launch_temb_rows(const int64_t* t, int B, int dim, T* out, int ldo, hipStream_t s);
template <typename T>
int launch_pack_cols(const float* x, int ldx, int M, int c0, int w, int wpad, float scale, T* out, int ldo, float* outf,
                     int ldof, hipStream_t s);
int launch_cfg_mix(const float* o, int ldo, int Mc, int cond_row0, int frames, int w, int has_null, float cond_scale,
                   float* eps, int lde, int c0, const float* x, int ldx, const float* c1, const float* c2, float* x0,
                   int ldx0, hipStream_t s);

// ---- token-per-lane fused Linear, K = 512, bf16 (tl_linear.hip) ------------------------------
// Row-indexed tensors are TILED (layouts at the top of tl_linear.hip; helpers below) and padded to 128-row blocks.
struct TlArgs {
    const void* X; int ldx;        // bf16 tiled [M, K] input (ldx = its width in features)
    const void* W;                 // bf16 [N, K], rows pi-permuted inside every 32-row tile (tl_weight_src_row)
    const float* bias;             // [N] or null
    const float* R; int ldr;       // fp32 tiled residual [M, N] or null (ldr unused)
    float* Cf; int ldcf;           // fp32 out or null: tiled [M, N], or row-major with ldcf when cf_rowmajor
    int cf_rowmajor;
    void* Ct; int ldct;            // bf16 tiled out [M, N] or null (ldct unused)
    int half_row0;                 // FiLM prologue: rows >= half_row0 index the batch as (row - half_row0) / frames
    int M, N, K, act;              // K = 512 or 1024
    const float* gamma; const float* beta;                          // prologue LayerNorm affine [K] (pro 1 / 3)
    const float* film; int film_ld, film_off, frames, bmod;         // pro 2: FOLDED FiLM table rows [A | B] (launch_film_fold)
    const float* row_const; int n_const_rows;                       // epilogue: + row_const[n] for rows < n_const_rows
    // prologue 3 (K = 1024 only): the row is the virtual concat [X(512) | X1(256) | X2(128) | X3(128, may be null)]
    // of four tiled tensors (transformer.py:304-312), LayerNorm over the first kreal columns; gamma/beta zero-padded
    const void* X1; int ld1; const void* X2; int ld2; const void* X3; int ld3; int kreal;
    int tiles_per_block;                                            // 32-feature tiles per blockIdx.y (set by the launcher)
    int stag_groups, stag_sleep;                                    // first-round start stagger (set by the launcher), as Tl2FfnArgs
    int rev;                                                        // 1: token blocks in descending order (tl_block_index, tl_common.h)
    int rot;                                                        // tl2 rolling loop: block b starts its weight stream at tile (b / 8) % tiles (set by the launcher)
    // residual stream as two bf16 planes (tl_common.h): Rlo != null -> the residual is (R reinterpreted as the bf16 hi plane) + Rlo;
    // Clo != null -> the result leaves as Ct (hi plane) + Clo (lo plane) and Cf is not written.  Planes are tiled bf16 [M, N].
    const void* Rlo; void* Clo;
    int tls_nb0, tls_tb1;                                           // window-chain kernels (tl_small.hip): 32-token blocks of the first row
                                                                    // range / first block of the second one (set by the launcher)
    int dbg;                                                        // ablation bits (bench only)
    unsigned long long* clk;                                        // clock probe output {shader cycles, 100 MHz ticks} or null
    unsigned long long* trace;                                      // block timeline (bench only): 4 words per block, or null
};
// pro: 0 = plain rows, 1 = LayerNorm, 2 = LayerNorm -> FiLM -> SiLU (StylizationBlock), 3 = concat + LayerNorm (feat_proj.0)
int launch_tl_linear(const TlArgs& a, int pro, hipStream_t s);
// window-chain batches (tl_small.hip): 32 tokens per block, one tile per wave, weights straight from the fragment-ordered copy
// (a.W as for launch_tl2_linear; pro 1 / 3: a.bias = d, a.row_const = c of the folded LayerNorm).  Rows = frames * bmod, or two
// CFG halves of that many rows with the second one starting at row M - frames * bmod.
bool tls_linear_supported(const TlArgs& a, int pro);
int launch_tls_linear(const TlArgs& a, int pro, hipStream_t s);
int tl_weight_src_row(int r);
// device-side application of the same row permutation to a bf16 [N, K] weight (test / bench helper of capi.hip)
int launch_tl_permute_weight(const void* W, int N, int K, void* dst, hipStream_t s);

// ---- second generation (tl2.hip): LDS-DMA weight stream from FRAGMENT-ORDERED weights -------------------------------
// same arguments as launch_tl_linear, except that a.W is the fragment-ordered copy of the weight (tl2_frag_index)
int launch_tl2_linear(const TlArgs& a, int pro, hipStream_t s);
// fourth form (tl4.hip, round 6): weights AND activations through an LDS ring, 64 x 128 outputs per wave; instantiated for feat_proj.1
// (pro 3), feat_proj.3 on hi / lo planes (pro 0, K = 1024) and q|k|v (pro 1); same arguments and bit-identical results
extern int g_tl_last_variant;          // test helper (dsh_debug_last_tl_variant): family the last token-per-lane Linear launch selected
bool tl4_linear_supported(const TlArgs& a, int pro);
int launch_tl4_linear(const TlArgs& a, int pro, hipStream_t s);
// element index of stored row n (inside 32-row tile nt; rows pi-permuted as for tl_linear) / column k of a [N, K] weight
size_t tl2_frag_index(int K, int nt, int n, int k);
// FFN branch of a decoder layer in one launch: h <- h + Sty(GELU(h16 W1^T + b1) W2^T + b2)   (transformer.py:169-181)
struct Tl2FfnArgs {
    const void* X;                       // bf16 tiled [M, 512]: the layer's residual stream (bf16 shadow)
    const void* Wffn;                    // weight stream: 80 chunks of 32 KB in phase order (W1 tiles / W2 K chunks interleaved, W3 tiles; tl2.hip)
    const float* b1; const float* b2; const float* b3;
    const float* film; int film_ld, film_off, frames, bmod, half_row0;   // folded FiLM rows [A | B] of ffn.proj_out
    const float* R; float* Cf; void* Ct; // h in (fp32 tiled), h out, bf16 shadow out
    const float* row_const; int n_const_rows;
    int M;
    unsigned long long* trace;           // block timeline (bench only) or null
    unsigned long long* clk;             // phase probe (bench only): 8 words per block, or null
    int stag_groups, stag_sleep;         // first-round start stagger (set by the launcher): block b < 256 sleeps (b % groups) * sleep * 8 k cycles
    int rev;                             // 1: token blocks in descending order (tl_block_index, tl_common.h)
    const void* Rhi; const void* Rlo; void* Clo;   // hi / lo planes of the residual stream (tl3_ffn_kernel only): Rhi != null replaces R / Cf
    // round 5 (plane form only): Y != null -> the attention branch's StylizationBlock runs as a first stage of the launch (tl3_ffn.hip, STY):
    // Y = bf16 tiled attention output [M, 512], bs1 = bias of its Linear, film_off1 = offset of its folded FiLM rows in `film`; Wffn then
    // starts with its 16 weight tiles (tl_pack_ffn_stream version 4), X is not read, and Rhi / Rlo are updated in place (Ct = Rhi, Clo = Rlo)
    const void* Y; const float* bs1; int film_off1;
};
void tl_stagger_config(int which, int* groups, int* sleep);   // DSH_STAGGER (tl2.hip)
bool tl2_ffn_supported(int M, int frames, int bmod);
int launch_tl2_ffn(const Tl2FfnArgs& a, hipStream_t s);
// third generation (tl3_ffn.hip): same arguments, Wffn in the version-3 stream order (K-outer Linear3 in two passes)
int launch_tl3_ffn(const Tl2FfnArgs& a, hipStream_t s);
bool tl3_ffn_supported(int M, int frames, int bmod, bool planes);   // the gate Denoiser::run_encoder uses for generation 3
// the 80-chunk weight stream of the fused FFN kernels from pi-permuted row-major bf16 weights ([1024,512], [512,1024], [512,512]);
// version 2: tl2_ffn_kernel, 3: tl3_ffn_kernel; `st` receives 80 * 16384 elements
void tl_pack_ffn_stream(int version, const uint16_t* w1p, const uint16_t* w2p, const uint16_t* w3p, uint16_t* st);
// the 16 chunks of the attention branch's StylizationBlock Linear ([512,512], pi-permuted rows) in front of a version-3 stream: tile t in
// fragment order = chunk t; `st16` receives 16 * 16384 elements
void tl_pack_sty_tiles(const uint16_t* wsp, uint16_t* st16);

// ---- tiled-layout helpers (rowops.hip).  bf16 tiles: 32 tokens x 16 features; fp32: lane-native 32 x 32 blocks ----
// row-major [M, w] (ld, element type TS = float or bf16) -> bf16 tiled [Mpad, Wd]; columns >= w are zero filled
template <typename TS>
int launch_tile_rows_bf16(const TS* src, int ld, int M, int w, void* dst, int Wd, hipStream_t s);
int launch_untile_rows_bf16(const void* src, int Wd, int M, int w, void* dst_bf16, int ld, hipStream_t s);
int launch_tile_rows_f32(const float* src, int ld, int M, float* dst, int Wd, hipStream_t s);
int launch_untile_rows_f32(const float* src, int Wd, int M, float* dst, int ld, hipStream_t s);
// FiLM table rows [scale | shift] -> folded [A | B] coefficients of the token-per-lane StylizationBlock prologue (in place)
int launch_film_fold(float* tab, int ld, int B, int nblk, int D, const float* gamma, const float* beta, hipStream_t s);
// round 6: the same from the DISTINCT embedding rows: dst[b] = fold(src[idx[b]]) for b < B (idx == null: identity; fold == 0: plain copy)
int launch_film_expand(const float* src, int ld, const int* idx, float* dst, int B, int nblk, int D, const float* gamma, const float* beta,
                       int fold, hipStream_t s);
int launch_gather_rows_f32(const float* src, int ld, const int* idx, float* dst, int ldd, int B, int w, hipStream_t s);
// layer-0 seed of the tiled residual stream from the row-major joint_embed output h0 [Mc, 512]:
// rows [0, Mc) (CFG-null half) = h0 + c, rows [row1, row1 + Mc) (conditional half) = h0; fp32 tiled + bf16 tiled shadow.
// has_null == 0: only rows [0, Mc) = h0.
int launch_seed_stream(const float* h0, int Mc, int D, const float* c, int has_null, int row1, float* h, void* h16, hipStream_t s,
                       void* hlo = nullptr);      // hlo != null: the stream is seeded as (h16 = hi, hlo = lo) planes and h is not written
// round 6 (tl_embed.hip): the same seed straight from the tiled bf16 channels of x — joint_embed + bias + PE + null constant + plane split
// in one launch; x_tiled [Mc, 16 nf] (nf = 7 or 9), wfrag = fragment-ordered [512, 16 nf] weight
int launch_tl_joint(const void* x_tiled, int nf, const void* wfrag, const float* bias, const float* pe, int frames, const float* cnull,
                    int Mc, int row1, void* hi, void* lo, hipStream_t s);
// round 6 (tl_aud.hip): encoder_aud behind its attention (two StylizationBlocks + FFN at D = 128) in one launch; Y bf16 [Mc,128] and X2 fp32
// [Mc,128] row-major, Wst = tl_aud_pack_stream, bias = [proj_out(sa) 128 | linear1 1024 | linear2 128 | proj_out(ffn) 128], film = FOLDED rows
// [A1 | B1 | A2 | B2] (128 each) of embedding row (token / frames) % bmod
void tl_aud_pack_stream(const float* ws1, const float* w1, const float* w2, const float* ws2, uint16_t* st);
void tl_aud_pack_audio_proj(const float* w, uint16_t* st);     // audio_proj [256,256] -> 4 more chunks per motion encoder behind the 18
// n_ap > 0: audio_proj([mel | aud_feat]) of n_ap motion encoders as further stages (bias_ap [n_ap][256]; tiled bf16 outputs [Mc, 256]);
// out_b must then be the right half of a [Mc, 256] bf16 buffer whose left half holds the mel features (ld_b = 256)
int launch_tl_aud_tail(const void* Y, const float* X2, const void* Wst, const float* bias, const float* film, int film_ld, int bmod, int frames,
                       int Mc, float* out_f, void* out_b, int ld_b, hipStream_t s, int n_ap = 0, const float* bias_ap = nullptr,
                       void* ap_out0 = nullptr, void* ap_out1 = nullptr);
// round 6 (tl_out.hip): the `out` head for both CFG halves + CFG mix (+ x0 = c1 x - c2 eps and its tiled bf16 copy) in one launch;
// hi = hi plane of the residual stream (null half at rows [0, Mc), conditional half at [row1, row1 + Mc)), wfrag / bias = the `out` Linear
// in fragment order padded to n_out_padded (128 or 160) rows, eps / x [Mc, C] with the encoder's w channels at column c0
int launch_tl_out_mix(const void* hi, const void* wfrag, const float* bias, int n_out_padded, int Mc, int row1, int has_null, int frames, int w,
                      int c0, int C, float cond_scale, float* eps, const float* x, const float* c1, const float* c2, float* x0, void* x0_tiled,
                      hipStream_t s);
// audio_proj of up to two motion encoders from the row-major bf16 [Mc, 256] operand straight into their tiled [Mc, 256] concat operands
// (tl_embed.hip); wfrag = tl_aud_pack_audio_proj per encoder (128 KB apart), bias [n_enc][256]
int launch_tl_aproj(const void* x256, const void* wfrag, const float* bias, int n_enc, void* out0, void* out1, int Mc, hipStream_t s);
// row-major fp32 [M, w] <-> hi / lo bf16 planes in the tiled layout (test helpers of capi.hip)
int launch_tile_rows_hilo(const float* src, int ld, int M, int w, void* hi, void* lo, int Wd, hipStream_t s);
int launch_untile_rows_hilo(const void* hi, const void* lo, int Wd, int M, int w, float* dst, int ld, hipStream_t s);

int launch_interp_time(const float* x, int B, int Tin, int C, float* y, int Tout, hipStream_t s);
int launch_affine_cols(const float* x, size_t n, int C, const float* mean, const float* stdv, float* y, hipStream_t s);

// linear ("efficient") self-attention core: y = softmax_ch(Q) (softmax_time(K)^T V)   (transformer.py:122-128)
template <typename T>
int launch_linear_attention(const T* qkv, int ldq, int nbatch, int frames, int D, int head_dim, T* y, int ldy,
                            hipStream_t s);
// cross-attention core (LinearTemporalCrossAttention, transformer.py:146-166): q [B,T,D] (ldq), kv [B,N,2D] = (k | v) (ldkv), fp32
int launch_linear_cross_attention(const float* q, int ldq, int nbatch, int frames, const float* kv, int ldkv, int frames_kv, int D,
                                  int head_dim, float* y, int ldy, hipStream_t s);
int launch_silu_f32(const float* x, float* y, size_t n, hipStream_t s);
// bf16 tiled qkv [M, 3D] -> bf16 tiled y [M, D]; batch b starts at row b * frames (b < half_batches) or
// half_row0 + (b - half_batches) * frames
bool linear_attention_sty_f32_supported(int frames, int D, int head_dim, int ldq, int ldy);
int launch_linear_attention_sty_f32(const float* qkv, int ldq, int nbatch, int frames, int D, float* s_out, int ldy, const float* film, int film_ld, int film_off,
                                    int bmod, hipStream_t s);
int launch_linear_attention_tiled(const void* qkv, int nbatch, int half_batches, int half_row0, int frames, int D, void* y,
                                  hipStream_t s, int rev = 0);

// ---- sampler element-wise kernels (sampler_kernels.hip) ------------------------------------
struct DdimStepArgs {
    float* x;              // [n] in/out sample
    const float* eps;      // [n] model output
    float* x0_out;         // [n] pred_xstart or null
    float c1, c2;          // sqrt(1/abar), sqrt(1/abar - 1)          (fp64 table -> fp32)
    float sqrt_ab_prev;    // sqrt_f32(f32(abar_prev))
    float sqrt_1m_ab_prev; // sqrt_f32(1 - f32(abar_prev))
    // eta != 0 (gaussian_diffusion.py:1011-1032): coef_eps = sqrt(1 - abar_prev - sigma^2) multiplies the re-derived eps (eta = 0:
    // = sqrt_1m_ab_prev), and sigma * noise1 is added (sigma already carries the t != 0 mask); noise1 == null: no such term
    float coef_eps, sigma;
    const float* noise1;   // [n] N(0,1): the randn_like of the step
    // RePaint blend (gaussian_diffusion.py:1034-1056); mask == null disables it
    const uint8_t* mask;   // [n] bool
    const float* gt;       // [n]
    const float* noise2;   // [n] N(0,1) for the gt branch
    int blend;             // 1: linear cross-fade on the first overlap_len frames
    int clip;              // clamp x0 to [-1,1] before re-deriving eps (clip_denoised)
    int overlap_len, frames, channels;
    size_t n;
    // --same_overlap_noisy: tail_in [B, overlap_len, C] replaces the noised gt on the out-painted frames (null: off / first
    // window); tail_out [B, overlap_len, C] receives the last overlap_len frames of the updated sample (null: off)
    const float* tail_in; float* tail_out;
    // channel range [c_lo, c_hi) of the update (0 / 0 = all): the expression and the gesture channels of a sample never interact in the sampler's
    // element-wise updates, so the two encoders' chains may advance on their own streams (sampler.hip, pipelined small-batch loop)
    int c_lo = 0, c_hi = 0;
};
int launch_ddim_step(const DdimStepArgs& a, hipStream_t s);
int launch_undo_step(float* x, const float* noise, float sqrt_1m_beta, float sqrt_beta, size_t n, hipStream_t s, int channels = 0, int c_lo = 0, int c_hi = 0);
struct DdpmStepArgs {
    float* x; const float* eps; const float* noise; float* x0_out;
    float c1, c2, coef1, coef2, sigma;   // sigma = exp(0.5*logvar) or 0 at t == 0
    int clip;
    size_t n;
    int channels = 0, c_lo = 0, c_hi = 0;   // channel range of the update (c_hi <= c_lo: all), as DdimStepArgs
};
int launch_ddpm_step(const DdpmStepArgs& a, hipStream_t s);
int launch_fill_i64(int64_t* p, int64_t v, size_t n, hipStream_t s);
int launch_fill_f32(float* p, float v, size_t n, hipStream_t s);
int launch_fill_step(int64_t* t, float* c1, float* c2, int64_t* level, int64_t tv, float c1v, float c2v, int64_t lv, int n, hipStream_t s);
struct LevelCopyArgs {
    char* work[4]; size_t bytes[4]; size_t off[4]; int nseg;
    char* slots; size_t stride; const int64_t* level; int restore;
};
int launch_level_copy(const LevelCopyArgs& a, hipStream_t s);
// Philox4x32-10 + Box-Muller standard normals; element i uses counter (offset + i/4)
int launch_philox_randn(float* out, size_t n, uint64_t seed, uint64_t offset, hipStream_t s);
// per-row streams: row b of `rows` x n_row values uses key `seed`, counter = (in-row quad index, row_keys[b]) (device array)
int launch_philox_randn_rows(float* out, int rows, size_t n_row, uint64_t seed, uint64_t offset, const uint64_t* row_keys,
                             hipStream_t s);

}  // namespace dsh
