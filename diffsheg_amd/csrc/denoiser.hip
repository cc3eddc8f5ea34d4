// UniDiffuser forward on MI355X: host-side orchestration of the HIP kernels, no host syncs inside
// eval().  Restates /root/reference/models/transformer.py:728-770 (UniDiffuser.forward), :496-587
// (MotionTransformer.forward) and :300-346 (layer) with the structural savings the reference leaves
// on the table:
//   * hubert_encoder (Conv-BN-GELU-Conv) and pid_embed are step-invariant -> set_condition(), once
//   * the 32+2 FiLM Linears (StylizationBlock.emb_layers) are one stacked GEMM per encoder
//   * q/k/v share one LayerNorm and one [512 -> 1536] GEMM
//   * the feat_proj concat is never materialised un-normalised; its LayerNorm writes the GEMM operand
//   * CFG: the unconditional half's concat row is the constant null_cond_emb, so feat_proj(null) is a
//     per-layer constant vector (computed at finalize()) added inside the next LayerNorm pass; the
//     concat/LN/feat_proj GEMMs run on the conditional half only; FiLM/time/speaker embeddings are
//     computed for B rows and indexed mod B.
#include <math.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <type_traits>

#include "denoiser.h"

namespace dsh {

namespace {

template <typename T>
class Denoiser final : public DenoiserBase {
  public:
    Denoiser(const ModelConfig& c, hipStream_t s) : cfg(c), st(s) {
        const char* t2 = getenv("DSH_TL2");
        tl2_on = !(t2 && atoi(t2) == 0);          // LDS-DMA token-per-lane kernels (tl2.hip); DSH_TL2=0: first generation
        tl2_all = t2 && atoi(t2) == 2;            // DSH_TL2=2: also for the HBM-bound (residual) instantiations
        const char* ff = getenv("DSH_FFN_FUSE");
        ffn_fuse = tl2_on && !(ff && atoi(ff) == 0);
        // alternate the row order of consecutive token-per-lane launches (tl_block_index, tl_common.h): measured 613.2 -> 609.6 ms per
        // step on three streams, 687 -> 673 ms on one (round 4); DSH_REV=0 disables it
        const char* rv = getenv("DSH_REV");
        rev_on = !(rv && atoi(rv) == 0);
        const char* fv = getenv("DSH_FFN_V");     // fused FFN kernel generation: 3 (default, tl3_ffn.hip) or 2 (tl2.hip); fixes the weight stream order
        ffn_ver = (fv && atoi(fv) == 2) ? 2 : 3;
        // residual stream of the token-per-lane path as two bf16 planes (hi = the old bf16 shadow, lo = bf16(h - hi); tl_common.h)
        // instead of fp32 + shadow: 4 instead of 6 bytes per value written by every residual-carrying launch.  DSH_HILO=0: fp32.
        const char* hl = getenv("DSH_HILO");
        hilo = ffn_ver == 3 && !tl2_all && !(hl && atoi(hl) == 0);
        // window-chain batches: 32-token blocks, one tile per wave (tl_small.hip) up to a few thousand token rows per launch; DSH_TLS=0: off
        const char* ts = getenv("DSH_TLS");
        const char* tr = getenv("DSH_TLS_ROWS");
        tls_on = tl2_on && hilo && !(ts && atoi(ts) == 0);
        const char* sk = getenv("DSH_DBG_SKIP");   // bench experiment only (results are garbage): skip launches of a layer, bit 0 feat_proj.1, 1 feat_proj.3,
        dbg_skip = sk ? atoi(sk) : 0;              // 2 q|k|v, 3 attention, 4 StylizationBlock (attention branch), 5 fused FFN — what each launch costs the STEP
        if (dbg_skip || getenv("DSH_SPLIT_AT")) {
            static bool warned = false;
            if (!warned) { fprintf(stderr, "[diffsheg_hip] WARNING: bench-only switches are set (DSH_DBG_SKIP=%d%s): %s\n", dbg_skip, getenv("DSH_SPLIT_AT") ? ", DSH_SPLIT_AT" : "",
                                   dbg_skip ? "launches are skipped, results are GARBAGE" : "sub-batch boundaries are moved"); warned = true; }
        }
        // the attention branch's StylizationBlock as the first stage of the fused FFN launch (tl3_ffn_kernel<..., STY>): bit-identical, built and
        // measured in round 5 — 577.3 ms per 950-clip step against 557.7 with the separate launch (profiles/r05_k_ab_ffn_sty.txt; the fused launch
        // 753 us against 486 + 182): with every CU in the stage at once its 16 phases run at the HBM wall (2.9 k cycles per phase, like pass B)
        // instead of in the shadow of other CUs' compute phases.  Off unless DSH_FFN_STY=1.
        const char* fs = getenv("DSH_FFN_STY");
        ffn_sty = fs && atoi(fs) != 0;
        const char* th = getenv("DSH_TL2_HL");
        tl2_hl = tl2_on && hilo && !(th && atoi(th) == 0);
        if (tr && atoi(tr) > 0) tls_rows = atoi(tr);
        // fp32 parity path (round 6): LayerNorm folded into q|k|v and feat_proj.1 with the row moments taken in the GEMM's own staging, the
        // StylizationBlock front (LN -> FiLM -> SiLU) in the A-operand staging of its Linear (gemm_f32_pro.hip) instead of four row kernels per
        // layer.  DSH_F32_FUSE=0: the separate row kernels of rounds 1 - 5.
        const char* f3 = getenv("DSH_F32_FUSE");
        // Bits (measurement): 1 folded LayerNorms, 2 StylizationBlock fronts, 4 the front-less Linears on the same software-pipelined main loop,
        // 8 (with 2) the attention branch's StylizationBlock front inside the attention launch (attention.hip) instead of its Linear's staging.
        f32_bits = (std::is_same<T, float>::value && c.latent_dim == 512) ? (f3 ? atoi(f3) & 15 : 15) : 0;
        f32_fuse = f32_bits != 0;
        // ... above the few-row GEMM's range only (gemm.hip: K split over the waves of a block up to DSH_GEMM_KSPLIT = 512 rows — at 34 rows the 64 x 64
        // tile launches measured 3.50 vs 2.62 ms per configs[0] evaluation); DSH_GEMM_KSPLIT=0, the reproducible mode, puts every batch size on them
        { const char* ks = getenv("DSH_GEMM_KSPLIT"); f32_min_rows = ks ? atoi(ks) : 512; }
    }
    // second instance on another stream that shares (does not own) the finalized weights; own workspace
    Denoiser(const Denoiser& o, hipStream_t s)
        : cfg(o.cfg), st(s), wbytes(o.wbytes), finalized(o.finalized), aud_te0(o.aud_te0), aud_te2(o.aud_te2),
          aud_film(o.aud_film), aud_stream(o.aud_stream), aud_ap_bias(o.aud_ap_bias), aud_bias(o.aud_bias), aud_film_g(o.aud_film_g), aud_film_b(o.aud_film_b), aud(o.aud), exp_(o.exp_), ges_(o.ges_), tl2_on(o.tl2_on), tl2_all(o.tl2_all), ffn_fuse(o.ffn_fuse), ffn_ver(o.ffn_ver), hilo(o.hilo), tls_on(o.tls_on), tl2_hl(o.tl2_hl), f32_bits(o.f32_bits), f32_min_rows(o.f32_min_rows), f32_fuse(o.f32_fuse), dbg_skip(o.dbg_skip), ffn_sty(o.ffn_sty), tls_rows(o.tls_rows), rev_on(o.rev_on) {
        for (Encoder* E : {&exp_, &ges_}) { E->pid_part = nullptr; E->pid_part_s = nullptr; E->hub = nullptr; E->film_tab = nullptr; E->aproj_buf = nullptr; }
    }
    DenoiserBase* clone_shared(hipStream_t s) override { return finalized ? new Denoiser(*this, s) : nullptr; }
    void notify_after_launches(hipEvent_t ev, int n) override { notify_ev = ev; notify_at = n; }
    ~Denoiser() override {
        for (void* p : allocs) (void)hipFree(p);
        for (void* p : ws_allocs) (void)hipFree(p);
        if (lvl_slots) (void)hipFree(lvl_slots);
    }

    int finalize(const std::map<std::string, HostTensor>& w) override;
    int set_condition(int B, int T_, const float* audio, const float* person_id, const float* hubert) override;
    int eval(const float* x, const int64_t* t, const float* c1, const float* c2, float* eps) override {
        return eval_level(x, t, c1, c2, eps, 0, nullptr);
    }
    int level_cache_prepare(int n_levels) override;
    int set_condition_light(int B, int T_, const float* audio, const float* person_id) override;
    int set_part(int p) override {
        DSH_REQUIRE(p >= 0 && p <= 2 && (p == 0 || !cfg.single_transformer), "set_part: 0 whole / 1 expression / 2 gesture (UniDiffuser only)");
        part = p;
        return 0;
    }
    int import_expr(DenoiserBase* src_, hipStream_t s) override {
        Denoiser<T>* src = dynamic_cast<Denoiser<T>*>(src_);
        DSH_REQUIRE(src && src != this && src->batch == batch && src->frames == frames && expr_x0 && src->expr_x0, "import_expr: incompatible instances");
        const size_t Mc = (size_t)batch * frames;
        DSH_HIP_CHECK(hipMemcpyAsync(expr_x0, src->expr_x0, Mc * expr_ld() * sizeof(float), hipMemcpyDeviceToDevice, s));
        if (tl_path()) DSH_HIP_CHECK(hipMemcpyAsync(expr16, src->expr16, (size_t)round_up((int)Mc, 32) * 128 * sizeof(T), hipMemcpyDeviceToDevice, s));
        return 0;
    }
    int level_slots(char** slots, size_t* stride, int* n) override { *slots = lvl_slots; *stride = lvl_stride; *n = lvl_n; return lvl_n > 0 ? 0 : -1; }
    int adopt_level_slots(char* slots, size_t stride, int n) override { lvl_borrowed = slots; lvl_borrowed_stride = stride; lvl_borrowed_n = n; return 0; }
    int eval_level(const float* x, const int64_t* t, const float* c1, const float* c2, float* eps, int mode, const int64_t* level) override;
    double issued_flops_per_eval() const override { return flops_last_eval; }
    size_t weight_bytes() const override { return wbytes; }
    int debug_copy(const std::string& what, float* out) override;

  private:
    struct Lin {
        T* w = nullptr; float* b = nullptr; int N = 0, K = 0, Kp = 0;
        T* wf = nullptr;                 // token-per-lane operands: fragment-ordered copy for the LDS-DMA kernels (tl2.hip);
        float* fd = nullptr;             //   when a LayerNorm precedes the Linear it is folded in: wf = gamma (.) W, fd[n] = b[n] +
        float* fc = nullptr;             //   sum_k beta[k] W[n][k], fc[n] = sum_k wf[n][k]  (tl2.hip, PRO 1 / 3)
        std::vector<T> hperm;            // host copy of the pi-permuted rows (only while finalize() builds the FFN stream)
        float* wfold = nullptr;          // fp32 parity path (gemm_f32_pro.hip): [N, Kp] = gamma (.) W, with fd / fc as above in natural feature order
    };
    struct LNp { float* g = nullptr; float* b = nullptr; int D = 0; };
    struct Sty { LNp ln; Lin out; };
    struct Layer {
        bool has_feat = false;
        int P = 0, Pp = 0;
        LNp ln0; Lin f1, f3; float* null_const = nullptr;
        LNp sa_ln; Lin qkv; Sty sty1; Lin ffn1, ffn2; Sty sty2;
        bool tl = false;             // qkv / sty*.out / ffn1 hold K-permuted weights for tl_linear (bf16, D = 512)
        T* ffn_stream = nullptr;     // W1 tiles / W2 K-chunks interleaved + W3 tiles: operand of the fused FFN kernel (tl2.hip)
    };
    struct Encoder {
        int cin = 0, cin_p = 0;
        Lin joint, aproj, conv1, conv2, te0, te2, pe0, pe2, film, out;
        Lin out_tl;                  // `out` as a tl_linear operand (bf16 path)
        T* joint_wf = nullptr; int joint_nf = 0;   // joint_embed in fragment order, K padded to 16 joint_nf (tl_embed.hip; bf16 path)
        float* pe = nullptr;
        std::vector<Layer> layers;
        // per-condition state
        float* pid_part = nullptr;   // [B, E] fp32 (per clip)
        float* pid_part_s = nullptr; // [n_spk, E] fp32 (per distinct speaker)
        T* hub = nullptr;            // [Mc, 128] (tiled on the token-per-lane path)
        float* film_tab = nullptr;   // [B, L*2*2D]
        T* aproj_buf = nullptr;      // [Mc, aud_latent] audio_proj([audio | aud_feat]) of the current evaluation (tiled on the token-per-lane path)
        float* film_g = nullptr;     // [2L, D] StylizationBlock LayerNorm gamma / beta stacked in FiLM-table order
        float* film_b = nullptr;     //         (token-per-lane path: folded into the table by launch_film_fold)
    };

    ModelConfig cfg;
    hipStream_t st;
    std::vector<void*> allocs, ws_allocs;
    size_t wbytes = 0;
    double flops_acc = 0, flops_last_eval = 0;
    bool finalized = false, conditioned = false;
    hipEvent_t notify_ev = nullptr;   // recorded on `st` after the notify_at-th token-per-lane launch of an eval
    int notify_at = 0, tl_launches = 0;

    Lin aud_te0, aud_te2, aud_film;
    T* aud_stream = nullptr; float* aud_ap_bias = nullptr; bool aproj_in_tail = false; float* aud_bias = nullptr; float* aud_film_g = nullptr; float* aud_film_b = nullptr;   // fused encoder_aud tail (tl_aud.hip; bf16 path)
    Layer aud;
    Encoder exp_, ges_;
    bool tl2_on = true, tl2_all = false, ffn_fuse = true;
    int ffn_ver = 3;
    bool hilo = false;
    bool tls_on = false;
    bool tl2_hl = false;                 // residual-carrying launches on the rolling LDS-DMA loop (round 5)
    int f32_bits = 0, f32_min_rows = 512;
    int f32_now() const { return batch * frames > f32_min_rows ? f32_bits : 0; }   // the bits that apply to the current condition's batch
    bool f32_fuse = false;               // fp32 path: LayerNorm / StylizationBlock fronts inside the GEMM launches (round 6, gemm_f32_pro.hip)
    int expr_ld() const { return f32_fuse ? round_up(cfg.expression_dim, 32) : cfg.expression_dim; }   // row stride of expr_x0 (zero padded to whole K tiles for the fused concat)
    int dbg_skip = 0;
    bool ffn_sty = false;                // the attention branch's StylizationBlock as the first stage of the fused FFN launch (round 5, off)
    static constexpr size_t FFN_STREAM_OFF = (size_t)16 * 16384;   // elements of L.ffn_stream in front of the FFN's own 80 chunks
    int tls_rows = 0;                // DSH_TLS_ROWS: one row limit for every instantiation (0: the measured per-instantiation limits in tl())
    bool rev_on = false; int rev_ctr = 0;
    int next_rev() { return rev_on ? (rev_ctr++ & 1) : 0; }

    // ---- workspace (grow-only) ----
    int capB = 0, capT = 0;
    float *audio_f = nullptr, *h = nullptr, *o = nullptr, *expr_x0 = nullptr, *film_aud_tab = nullptr, *aud_feat_f = nullptr;
    T *temb = nullptr, *hid = nullptr, *semb = nullptr, *pid_in = nullptr, *audio256 = nullptr,
      *x_in = nullptr, *h16 = nullptr, *n = nullptr, *y = nullptr, *s = nullptr, *qkv = nullptr, *U = nullptr,
      *g = nullptr, *y2 = nullptr, *col = nullptr, *z = nullptr, *expr16 = nullptr, *aproj_rm = nullptr, *hub_rm = nullptr, *qkv_rm = nullptr, *y_rm = nullptr;
    // round 6: distinct speakers of the current condition (exact comparison of the style rows on the host, once per set_condition)
    int n_spk = 0;                   // number of distinct person_id rows
    int* spk_idx = nullptr;          // [B] device: clip -> distinct row
    float* pid_rep = nullptr;        // [n_spk, style] device: the distinct rows
    float* film_small = nullptr;     // [rows, film.N] FiLM Linear output on the distinct rows of one encoder (expanded into E.film_tab)
    std::vector<float> pid_host; std::vector<int> spk_idx_host;
    int emb_rows() const { return t_uniform ? n_spk : batch; }      // rows the embedding Linears run on
    // round 6: the timestep-independent front of encoder_aud — x = 2 audio, y = attention(q|k|v(LayerNorm(x))) (transformer.py:302-303,
    // :119-128: nothing in front of the first StylizationBlock sees the embedding) — is computed once per condition
    float* aud_x2 = nullptr;         // [Mc, 128] fp32: 2 * mel features (encoder_aud's residual input)
    T* aud_y = nullptr;              // [Mc, 128]: its self-attention output, in front of sa_block.proj_out
    float* h0 = nullptr;             // row-major joint_embed output, seed of the tiled residual stream (token-per-lane path)
    T* hlo = nullptr;                // lo plane of the residual stream (hilo; the hi plane is h16)
    bool tl_path() const { return !ges_.layers.empty() && ges_.layers[0].tl; }
    std::vector<Encoder*> encs() { return cfg.single_transformer ? std::vector<Encoder*>{&ges_} : std::vector<Encoder*>{&exp_, &ges_}; }

    static constexpr int KA = gemm_k_align<T>();
    static int kpad(int k) { return round_up(k, KA); }

    template <typename U_> int dalloc(U_** p, size_t nelem, std::vector<void*>& pool) {
        void* q = nullptr;
        DSH_HIP_CHECK(hipMalloc(&q, std::max<size_t>(nelem, 1) * sizeof(U_)));
        pool.push_back(q);
        *p = reinterpret_cast<U_*>(q);
        return 0;
    }
    int upload_f32(float** dst, const float* src, size_t nelem) {
        if (int e = dalloc(dst, nelem, allocs)) return e;
        DSH_HIP_CHECK(hipMemcpy(*dst, src, nelem * sizeof(float), hipMemcpyHostToDevice));
        wbytes += nelem * sizeof(float);
        return 0;
    }
    // weight [N,K] fp32 host -> T device, zero padded along K to the 128-byte tile.  tl_perm: operand of tl_linear —
    // rows zero-padded to a multiple of 32 and pi-permuted inside every 32-row tile (tl_weight_src_row); L.N = padded N
    int make_lin(Lin& L, const float* W, const float* bias, int N, int K, bool tl_perm = false, int force_kp = 0, bool keep_host = false,
                 const float* fold_gamma = nullptr, const float* fold_beta = nullptr) {
        const int Np = tl_perm ? round_up(N, 32) : N;
        L.N = Np; L.K = K; L.Kp = force_kp ? force_kp : kpad(K);
        std::vector<T> tmp((size_t)Np * L.Kp);
        for (int r = 0; r < Np; ++r) {
            const int sr = tl_perm ? tl_weight_src_row(r) : r;
            for (int k = 0; k < L.Kp; ++k)
                tmp[(size_t)r * L.Kp + k] = from_f32<T>((sr < N && k < K) ? W[(size_t)sr * K + k] : 0.f);
        }
        if (int e = dalloc(&L.w, tmp.size(), allocs)) return e;
        DSH_HIP_CHECK(hipMemcpy(L.w, tmp.data(), tmp.size() * sizeof(T), hipMemcpyHostToDevice));
        wbytes += tmp.size() * sizeof(T);
        if (tl_perm && (L.Kp == 512 || L.Kp == 1024)) {
            std::vector<T> fr(tmp.size());
            if (fold_gamma) {
                // LayerNorm folded into the operand of the LDS-DMA kernel: LN(x) W^T + b = rstd (x W'^T - mean c) + d
                std::vector<float> fc(Np, 0.f), fd(Np, 0.f);
                for (int r = 0; r < Np; ++r) {
                    const int sr = tl_weight_src_row(r);
                    double c = 0, d = (sr < N && bias) ? bias[sr] : 0.0;
                    for (int k = 0; k < L.Kp; ++k) {
                        const float w = (sr < N && k < K) ? W[(size_t)sr * K + k] : 0.f;
                        const T wq = from_f32<T>(w * (k < K ? fold_gamma[k] : 0.f));
                        tmp[(size_t)r * L.Kp + k] = wq;
                        c += (double)to_f32<T>(wq);
                        d += (double)(k < K ? fold_beta[k] : 0.f) * w;
                    }
                    if (sr < Np) { fc[sr] = (float)c; fd[sr] = (float)d; }      // indexed by output feature (natural order), like the bias
                }
                if (int e = upload_f32(&L.fc, fc.data(), Np)) return e;
                if (int e = upload_f32(&L.fd, fd.data(), Np)) return e;
            }
            for (int r = 0; r < Np; ++r)
                for (int k = 0; k < L.Kp; ++k) fr[tl2_frag_index(L.Kp, r >> 5, r & 31, k)] = tmp[(size_t)r * L.Kp + k];
            if (int e = dalloc(&L.wf, fr.size(), allocs)) return e;
            DSH_HIP_CHECK(hipMemcpy(L.wf, fr.data(), fr.size() * sizeof(T), hipMemcpyHostToDevice));
            wbytes += fr.size() * sizeof(T);
            if (keep_host) L.hperm = std::move(tmp);
        }
        if (bias) {
            std::vector<float> bp(Np, 0.f);
            std::copy(bias, bias + N, bp.begin());
            if (int e = upload_f32(&L.b, bp.data(), Np)) return e;
        }
        if (sizeof(T) == 4 && !tl_perm && fold_gamma && fold_beta && bias) {
            // fp32 parity path: LayerNorm folded into the Linear behind it, LN(x) W^T + b = rstd (x W'^T - mean c) + d (sums in fp64, one rounding)
            std::vector<float> wf((size_t)N * L.Kp, 0.f), fc(N), fd(N);
            for (int r = 0; r < N; ++r) {
                double c = 0, d = bias[r];
                for (int k = 0; k < K; ++k) {
                    const float w = W[(size_t)r * K + k], wq = w * fold_gamma[k];
                    wf[(size_t)r * L.Kp + k] = wq;
                    c += (double)wq;
                    d += (double)fold_beta[k] * w;
                }
                fc[r] = (float)c; fd[r] = (float)d;
            }
            if (int e = upload_f32(&L.wfold, wf.data(), wf.size())) return e;
            if (int e = upload_f32(&L.fc, fc.data(), N)) return e;
            if (int e = upload_f32(&L.fd, fd.data(), N)) return e;
        }
        return 0;
    }
    const HostTensor* find(const std::map<std::string, HostTensor>& w, const std::string& k) {
        auto it = w.find(k);
        if (it == w.end()) { set_last_error("missing weight '" + k + "'"); return nullptr; }
        return &it->second;
    }
    int lin_from(const std::map<std::string, HostTensor>& w, const std::string& p, Lin& L, int N, int K, bool tl_perm = false, int force_kp = 0,
                 bool keep_host = false, const float* fold_gamma = nullptr, const float* fold_beta = nullptr) {
        const HostTensor* W = find(w, p + ".weight"); if (!W) return -1;
        const HostTensor* B = find(w, p + ".bias"); if (!B) return -1;
        DSH_REQUIRE((int64_t)W->numel() == (int64_t)N * K && (int)B->numel() == N, ("shape mismatch for " + p).c_str());
        return make_lin(L, W->data.data(), B->data.data(), N, K, tl_perm, force_kp, keep_host, fold_gamma, fold_beta);
    }
    int ln_from(const std::map<std::string, HostTensor>& w, const std::string& p, LNp& l, int D) {
        const HostTensor* G = find(w, p + ".weight"); if (!G) return -1;
        const HostTensor* B = find(w, p + ".bias"); if (!B) return -1;
        DSH_REQUIRE((int)G->numel() == D && (int)B->numel() == D, ("shape mismatch for " + p).c_str());
        l.D = D;
        if (int e = upload_f32(&l.g, G->data.data(), D)) return e;
        return upload_f32(&l.b, B->data.data(), D);
    }
    int sty_from(const std::map<std::string, HostTensor>& w, const std::string& p, Sty& s_, int D, bool tl_perm, bool keep_host = false) {
        if (int e = ln_from(w, p + ".norm", s_.ln, D)) return e;
        return lin_from(w, p + ".out_layers.2", s_.out, D, D, tl_perm, 0, keep_host);
    }
    int layer_from(const std::map<std::string, HostTensor>& w, const std::string& p, Layer& L, int D, int P,
                   const float* null_emb);
    int encoder_from(const std::map<std::string, HostTensor>& w, const std::string& p, Encoder& E, int cin, int P, int audio_k);
    int film_from(const std::map<std::string, HostTensor>& w, const std::vector<std::string>& prefixes, Lin& L, int D);

    int gemm(const Lin& L, const T* A, int lda, int M, int act, bool act_after, const float* R, int ldr, int res_mod,
             float* Cf, int ldcf, T* Ct, int ldct) {
        GemmArgs a;
        a.A = A; a.lda = lda; a.W = L.w; a.ldw = L.Kp; a.bias = L.b; a.R = R; a.ldr = ldr; a.res_mod = res_mod;
        a.Cf = Cf; a.ldcf = ldcf; a.Ct = Ct; a.ldct = ldct; a.M = M; a.N = L.N; a.K = L.Kp; a.act = act;
        a.act_after_res = act_after ? 1 : 0;
        const double fl = 2.0 * M * (double)L.N * L.K;
        flops_acc += fl;
        if (prof) prof->begin(PROF_GEMM);
        int rc;
        if ((f32_bits & 4) && sizeof(T) == 4 && M > f32_min_rows && L.N % 4 == 0 && res_mod == 0 && !act_after && L.b && !(Cf && Ct) && (Cf || Ct) &&
            lda % 4 == 0 && (!R || ldr % 4 == 0) && (Cf ? ldcf : ldct) % 4 == 0) {
            // (above the few-row kernels' range: the software-pipelined 64 x 64 main loop of gemm_f32_pro.hip, 5 - 11 % faster per launch at M = 8704)
            GemmProArgs q = one_seg(reinterpret_cast<const float*>(A), lda, L.Kp);
            q.pro = 0; q.k_real = L.Kp; q.K = L.Kp; q.W = reinterpret_cast<const float*>(L.w); q.ldw = L.Kp; q.bias = L.b; q.R = R; q.ldr = ldr;
            q.C = Cf ? Cf : reinterpret_cast<float*>(Ct); q.ldc = Cf ? ldcf : ldct; q.M = M; q.N = L.N; q.act = act; q.frames = 1; q.bmod = 1;
            rc = launch_gemm_f32_pro(q, st);
        } else rc = launch_gemm<T>(a, st);
        if (prof) prof->end(fl);
        return rc;
    }
    // fp32 path, round 6: Linear + the LayerNorm (pro 1, folded; up to four concat segments) or StylizationBlock front (pro 2) before it (gemm_f32_pro.hip)
    int gemm_pro(const Lin& L, int pro, const GemmProArgs& segs, int k_real, int M, int act, const float* film, int film_ld, int film_off, int fr, int bmod,
                 const float* R, float* C, int ldc, const float* row_const = nullptr, int n_const_rows = 0) {
        GemmProArgs a = segs;
        a.row_const = row_const; a.n_const_rows = n_const_rows;
        a.pro = pro; a.k_real = k_real; a.K = L.Kp;
        a.W = reinterpret_cast<const float*>(pro == 1 ? (const void*)L.wfold : (const void*)L.w); a.ldw = L.Kp;
        a.bias = pro == 1 ? L.fd : L.b; a.fc = pro == 1 ? L.fc : nullptr;
        a.film = film; a.film_ld = film_ld; a.film_off = film_off; a.frames = fr > 0 ? fr : 1; a.bmod = bmod > 0 ? bmod : 1;
        a.R = R; a.ldr = L.N; a.C = C; a.ldc = ldc; a.M = M; a.N = L.N; a.act = act; a.nt_n = a.nt_m = 0;
        DSH_REQUIRE(pro != 1 || (L.wfold && L.fc && L.fd), "folded LayerNorm operands are missing");
        const double fl = 2.0 * M * (double)L.N * L.K;
        flops_acc += fl;
        if (prof) prof->begin(PROF_GEMM);
        const int rc = launch_gemm_f32_pro(a, st);
        if (prof) prof->end(fl);
        return rc;
    }
    static GemmProArgs one_seg(const float* x, int ld, int K) {
        GemmProArgs a{};
        a.seg[0] = x; a.seg_ld[0] = ld;
        a.seg_end[0] = a.seg_end[1] = a.seg_end[2] = a.seg_end[3] = K / 32;
        return a;
    }
    int run_block_tail_fused(const Layer& L, int M, int D, int nbatch, int fr, const float* film, int film_ld, int film_off0, int bmod, int has_null, int r0,
                             bool const_done, const float* next_const);
    // token-per-lane fused Linear (bf16, K = 512): prologue pro (0 plain / 1 LN / 2 LN+FiLM+SiLU) on X
    int tl(const Lin& L, int pro, const T* X, int M, int act, const LNp* ln, const float* film, int film_ld, int film_off,
           int fr, int bmod, const float* R, float* Cf, T* Ct, const float* row_const, int n_const_rows,
           const T* cat1 = nullptr, const T* cat2 = nullptr, const T* cat3 = nullptr, int kreal = 0,
           int half_row0 = 0x7fffffff, int cf_rowmajor_ld = 0, const T* Rlo = nullptr, T* Clo = nullptr) {
        TlArgs a;
        a.Rlo = Rlo; a.Clo = Clo;            // hi / lo planes of the residual stream: R is then the hi plane (bf16), Cf is null
        a.X = X; a.ldx = L.Kp; a.K = L.Kp; a.W = L.w; a.bias = L.b; a.R = R; a.ldr = L.N; a.Cf = Cf; a.ldcf = L.N; a.Ct = Ct; a.ldct = L.N;
        a.half_row0 = half_row0; a.cf_rowmajor = cf_rowmajor_ld > 0 ? 1 : 0;
        if (cf_rowmajor_ld > 0) a.ldcf = cf_rowmajor_ld;
        a.M = M; a.N = L.N; a.act = act; a.gamma = ln ? ln->g : nullptr; a.beta = ln ? ln->b : nullptr;
        a.film = film; a.film_ld = film_ld; a.film_off = film_off; a.frames = fr > 0 ? fr : 1; a.bmod = bmod > 0 ? bmod : 1;
        a.row_const = row_const; a.n_const_rows = n_const_rows; a.dbg = 0; a.clk = nullptr;
        a.X1 = cat1; a.ld1 = cfg.aud_latent_dim; a.X2 = cat2; a.ld2 = cfg.hubert_enc_dim; a.X3 = cat3; a.ld3 = 128; a.kreal = kreal;
        if (pro == 3) { a.ldx = cfg.latent_dim; }
        const double fl = 2.0 * M * (double)L.N * L.K;
        flops_acc += fl;
        // algorithmic HBM bytes of this launch: input rows + weight once + residual + outputs
        const double by = (double)M * L.K * 2 + (double)L.N * L.K * 2 + (R ? (double)M * L.N * 4 : 0.0) +
                          (Cf ? (double)M * L.N * 4 : 0.0) + (Ct ? (double)M * L.N * 2 : 0.0) + (Clo ? (double)M * L.N * 2 : 0.0);
        int cls = PROF_TL_QKV;
        if (pro == 2) cls = PROF_TL_STY;
        else if (pro == 3) cls = PROF_TL_FEAT1;
        else if (L.Kp == 1024) cls = R ? PROF_TL_FEAT3 : PROF_TL_FFN2;
        else if (pro == 0) cls = PROF_TL_FFN1;
        a.trace = nullptr;
        a.rev = (M >= 4096) ? next_rev() : 0;
        // LDS-DMA kernels for the MFMA-bound instantiations; the HBM-bound ones (fp32 residual in / out: StylizationBlock,
        // feat_proj.3) stay on the first generation, whose two independent 128-token blocks per CU ride out memory stalls
        // better than one 256-token block behind a single barrier (measured: 219 vs 269 us, 162 vs 184 us)
        // window-chain batches: 32-token blocks with one tile per wave (tl_small.hip); same arithmetic, operation for operation
        bool small = false;
        // up to a per-instantiation row count (measured per launch at 16 / 32 window chains = 2944 / 5888 rows, conditional half 1408 /
        // 2816, whole-chip kernel vs this family, us): StylizationBlock 19.7 / 20.1 vs 11.8 / 15.9; ffn.linear1 10.5 / 13.4 vs < 8.7 / 13.2;
        // ffn.linear2 13.0 / 12.5 vs 10.7 / 15.4; q|k|v 13.7 / 17.5 vs 14.2 / 21.5; feat_proj.1 14.0 / 15.3 vs 12.2 / 16.0; feat_proj.3
        // 10.4 / 12.1 vs 8.9 / 11.2.  DSH_TLS_ROWS=n replaces all six limits by n.
        int tls_limit = tls_rows;
        if (tls_limit <= 0) {
            if (pro == 2) tls_limit = 6144;
            else if (pro == 1) tls_limit = 2560;
            else if (pro == 3) tls_limit = 2048;
            else if (L.Kp == 512) tls_limit = 6144;
            else tls_limit = Rlo ? 3072 : 4096;
        }
        if (tls_on && L.wf && M <= tls_limit && (pro == 0 || pro == 2 || (L.fd && L.fc))) {
            TlArgs b = a;
            b.W = L.wf;
            if (pro == 1 || pro == 3) { b.bias = L.fd; b.row_const = L.fc; }
            if (tls_linear_supported(b, pro)) { a = b; small = true; }
        }
        // round 5: the two residual-carrying launches on hi / lo planes take the rolling LDS-DMA loop as well (tl2_linear_kernel<..., ROLL, HL>)
        // at whole-chip token counts (no N split: at least 128 token blocks); DSH_TL2_HL=0 keeps them on the first generation
        // (the rolling kernel's FiLM prologue stages at most 10 clips per 256-token block and its asm stores take 32-bit plane offsets:
        //  windows of 26 .. 28 frames at whole-chip batch, or planes of 4 GiB and more, stay on the first-generation kernel)
        const bool hl2_fits = (pro != 2 || std::min(255 / (fr > 0 ? fr : 1) + 2, bmod > 0 ? bmod : 1) <= 10) && (size_t)M * L.N * 2 < ((size_t)1 << 32);
        const bool hl2 = tl2_hl && Rlo && L.wf && !small && hl2_fits && ((pro == 2 && L.Kp == 512 && M >= 128 * 256) || (pro == 0 && L.Kp == 1024 && M >= 128 * 128));
        const bool use2 = !small && tl2_on && L.wf && (((!R || tl2_all) && !Rlo) || hl2);
        if (use2) {
            a.W = L.wf;
            if (pro == 1 || pro == 3) {
                DSH_REQUIRE(L.fd && L.fc, "token-per-lane Linear behind a LayerNorm needs the folded weight vectors");
                a.bias = L.fd; a.row_const = L.fc;
            }
        }
        if (prof) prof->begin(cls);
        const int rc = small ? launch_tls_linear(a, pro, st) : (use2 ? launch_tl2_linear(a, pro, st) : launch_tl_linear(a, pro, st));
        if (prof) prof->end(fl, by);
        if (notify_ev && ++tl_launches == notify_at) DSH_HIP_CHECK(hipEventRecord(notify_ev, st));
        return rc;
    }
    const T* hT() const { return sizeof(T) == 4 ? reinterpret_cast<const T*>(h) : h16; }
    T* h16_out() const { return sizeof(T) == 4 ? nullptr : h16; }

    int ensure_workspace(int B, int T_);
    int run_block_tail(const Layer& L, int M, int D, int nbatch, int frames, const float* film, int film_ld, int film_off0,
                       int bmod, float* hres, T* h16o, const T* hA_after_sty1, const float* res_in = nullptr, const T* y_in = nullptr);
    int prep_audio(const int64_t* t);
    static bool aud_hoist() { static const bool on = [] { const char* e = getenv("DSH_AUD_HOIST"); return !(e && atoi(e) == 0); }(); return on; }
    int aud_front() {
        const int DA = cfg.audio_dim, Mc_ = batch * frames;
        if (int e = launch_pack_cols<T>(audio_f, DA, Mc_, 0, DA, DA, 2.0f, (T*)nullptr, 0, aud_x2, DA, st)) return e;
        if (int e = launch_ln_rows<T>(aud_x2, DA, Mc_, DA, nullptr, 0, aud.sa_ln.g, aud.sa_ln.b, n, DA, st)) return e;
        if (int e = gemm(aud.qkv, n, DA, Mc_, ACT_NONE, false, nullptr, 0, 0, nullptr, 0, qkv, 3 * DA)) return e;
        return launch_linear_attention<T>(qkv, 3 * DA, batch, frames, DA, DA / cfg.num_heads, aud_y, DA, st);
    }
    int prep_encoder(Encoder& E);
    int run_encoder(Encoder& E, const float* x, int c0, int w, const float* expr, int expr_w, const float* c1,
                    const float* c2, float* eps, bool want_x0);
    // timestep cache: n slots of [film_tab(exp) | film_tab(ges) | aproj(exp) | aproj(ges)] for the current condition
    char* lvl_slots = nullptr; size_t lvl_stride = 0, lvl_cap = 0; int lvl_n = 0;
    char* lvl_borrowed = nullptr; size_t lvl_borrowed_stride = 0; int lvl_borrowed_n = 0;   // prefetch instance: the main instance's slots
    bool light_cond = false;         // set_condition_light(): no hubert features -> only mode 3 may run
    int part = 0;                    // 0 whole evaluation / 1 expression encoder only / 2 gesture encoder only (pipelined small-batch loop, denoiser.h)
    int level_copy(const int64_t* level, int restore);
};

// ------------------------------------------------------------------------------------------------
static void host_layernorm(const float* x, const float* g, const float* b, int P, std::vector<double>& out) {
    double mean = 0; for (int i = 0; i < P; ++i) mean += x[i]; mean /= P;
    double var = 0; for (int i = 0; i < P; ++i) var += (x[i] - mean) * (x[i] - mean); var /= P;
    const double rstd = 1.0 / sqrt(var + 1e-5);
    out.resize(P);
    for (int i = 0; i < P; ++i) out[i] = (x[i] - mean) * rstd * g[i] + b[i];
}

template <typename T>
int Denoiser<T>::layer_from(const std::map<std::string, HostTensor>& w, const std::string& p, Layer& L, int D, int P,
                            const float* null_emb) {
    const int F = cfg.ff_size;
    L.has_feat = P > 0;
    L.tl = std::is_same<T, bf16>::value && D == 512 && F == 1024;
    if (L.has_feat) {
        L.P = P; L.Pp = L.tl ? 1024 : kpad(P);
        DSH_REQUIRE(P <= 1024, "concat width exceeds the K = 1024 token-per-lane kernel");
        if (L.tl) {   // gamma/beta zero-padded to the 1024-wide token-per-lane row: padded columns normalise to exactly 0
            const HostTensor* G = find(w, p + ".feat_proj.0.weight"); const HostTensor* Bz = find(w, p + ".feat_proj.0.bias");
            if (!G || !Bz) return -1;
            DSH_REQUIRE((int)G->numel() == P && (int)Bz->numel() == P, ("shape mismatch for " + p + ".feat_proj.0").c_str());
            std::vector<float> gp(1024, 0.f), bp(1024, 0.f);
            std::copy(G->data.begin(), G->data.end(), gp.begin());
            std::copy(Bz->data.begin(), Bz->data.end(), bp.begin());
            L.ln0.D = P;
            if (int e = upload_f32(&L.ln0.g, gp.data(), 1024)) return e;
            if (int e = upload_f32(&L.ln0.b, bp.data(), 1024)) return e;
        } else if (int e = ln_from(w, p + ".feat_proj.0", L.ln0, P)) return e;
        {
            const HostTensor *g0 = find(w, p + ".feat_proj.0.weight"), *b0 = find(w, p + ".feat_proj.0.bias");
            if (!g0 || !b0) return -1;
            const bool fold = L.tl || (f32_fuse && D == 512);
            if (int e = lin_from(w, p + ".feat_proj.1", L.f1, 2 * D, P, L.tl, L.tl ? 1024 : 0, false, fold ? g0->data.data() : nullptr,
                                 fold ? b0->data.data() : nullptr)) return e;
        }
        if (int e = lin_from(w, p + ".feat_proj.3", L.f3, D, 2 * D, L.tl)) return e;
        if (null_emb) {
            // feat_proj(null_cond_emb): one constant vector per layer (transformer.py:326-338), fp64 on host
            const HostTensor *g0 = find(w, p + ".feat_proj.0.weight"), *b0 = find(w, p + ".feat_proj.0.bias"),
                             *w1 = find(w, p + ".feat_proj.1.weight"), *b1 = find(w, p + ".feat_proj.1.bias"),
                             *w3 = find(w, p + ".feat_proj.3.weight"), *b3 = find(w, p + ".feat_proj.3.bias");
            std::vector<double> u;
            host_layernorm(null_emb, g0->data.data(), b0->data.data(), P, u);
            std::vector<double> hdn(2 * D);
            for (int o_ = 0; o_ < 2 * D; ++o_) {
                double a = b1->data[o_];
                const float* wr = &w1->data[(size_t)o_ * P];
                for (int k = 0; k < P; ++k) a += (double)wr[k] * u[k];
                hdn[o_] = a / (1.0 + exp(-a));
            }
            std::vector<float> c(D);
            for (int o_ = 0; o_ < D; ++o_) {
                double a = b3->data[o_];
                const float* wr = &w3->data[(size_t)o_ * 2 * D];
                for (int k = 0; k < 2 * D; ++k) a += (double)wr[k] * hdn[k];
                c[o_] = (float)a;
            }
            if (int e = upload_f32(&L.null_const, c.data(), D)) return e;
        }
    }
    if (int e = ln_from(w, p + ".sa_block.norm", L.sa_ln, D)) return e;
    {   // fused q|k|v  [3D, D]
        const HostTensor *wq = find(w, p + ".sa_block.query.weight"), *wk = find(w, p + ".sa_block.key.weight"),
                         *wv = find(w, p + ".sa_block.value.weight"), *bq = find(w, p + ".sa_block.query.bias"),
                         *bk = find(w, p + ".sa_block.key.bias"), *bv = find(w, p + ".sa_block.value.bias");
        if (!wq || !wk || !wv || !bq || !bk || !bv) return -1;
        DSH_REQUIRE((int64_t)wq->numel() == (int64_t)D * D, ("shape mismatch for " + p + ".sa_block.query").c_str());
        std::vector<float> W3((size_t)3 * D * D), B3((size_t)3 * D);
        std::memcpy(&W3[0], wq->data.data(), sizeof(float) * D * D);
        std::memcpy(&W3[(size_t)D * D], wk->data.data(), sizeof(float) * D * D);
        std::memcpy(&W3[(size_t)2 * D * D], wv->data.data(), sizeof(float) * D * D);
        std::memcpy(&B3[0], bq->data.data(), sizeof(float) * D);
        std::memcpy(&B3[D], bk->data.data(), sizeof(float) * D);
        std::memcpy(&B3[2 * D], bv->data.data(), sizeof(float) * D);
        const HostTensor *lg = find(w, p + ".sa_block.norm.weight"), *lb = find(w, p + ".sa_block.norm.bias");
        if (!lg || !lb) return -1;
        const bool fold = L.tl || (f32_fuse && D == 512);
        if (int e = make_lin(L.qkv, W3.data(), B3.data(), 3 * D, D, L.tl, 0, false, fold ? lg->data.data() : nullptr,
                             fold ? lb->data.data() : nullptr)) return e;
    }
    if (int e = sty_from(w, p + ".sa_block.proj_out", L.sty1, D, L.tl, L.tl)) return e;
    if (int e = lin_from(w, p + ".ffn.linear1", L.ffn1, F, D, L.tl, 0, L.tl)) return e;
    if (int e = lin_from(w, p + ".ffn.linear2", L.ffn2, D, F, L.tl, 0, L.tl)) return e;
    if (int e = sty_from(w, p + ".ffn.proj_out", L.sty2, D, L.tl, L.tl)) return e;
    if (L.tl) {
        // weight stream of the fused FFN kernel, 32 KB chunks in the order its phases consume them (tl_pack_ffn_stream, tl3_ffn.hip):
        //   W1 tile j (GEMM1) at chunk c1(j) = j ? 2 j - 1 : 0 | K chunk j of W2 (GEMM2) as fragments (output tile ot, k step ks) at
        //   (2 ot + ks) KB, at chunk c2(j) = j < 31 ? 2 j + 2 : 63 | W3 from chunk 64 in the order of the kernel generation (ffn_ver)
        //   (round 5) in FRONT of them the 16 weight tiles of the attention branch's StylizationBlock Linear, which the fused launch runs as its
        //   first stage (tl3_ffn_kernel<..., STY>); the plain FFN launch starts FFN_STREAM_OFF elements into the stream
        constexpr size_t CH = 16384;                       // bf16 elements per 32 KB chunk
        std::vector<T> st((size_t)(16 + 64 + 16) * CH);
        static_assert(sizeof(T) == 2 || sizeof(T) == 4, "element type");
        if (sizeof(T) == 2) {
            // (the 16 chunks in front are only filled when the default-off fused attention-branch stage is enabled, DSH_FFN_STY=1)
            if (ffn_sty) tl_pack_sty_tiles(reinterpret_cast<const uint16_t*>(L.sty1.out.hperm.data()), reinterpret_cast<uint16_t*>(st.data()));
            tl_pack_ffn_stream(ffn_ver, reinterpret_cast<const uint16_t*>(L.ffn1.hperm.data()), reinterpret_cast<const uint16_t*>(L.ffn2.hperm.data()),
                               reinterpret_cast<const uint16_t*>(L.sty2.out.hperm.data()), reinterpret_cast<uint16_t*>(st.data()) + FFN_STREAM_OFF);
        }
        if (int e = dalloc(&L.ffn_stream, st.size(), allocs)) return e;
        DSH_HIP_CHECK(hipMemcpy(L.ffn_stream, st.data(), st.size() * sizeof(T), hipMemcpyHostToDevice));
        wbytes += st.size() * sizeof(T);
        L.ffn1.hperm = std::vector<T>(); L.ffn2.hperm = std::vector<T>(); L.sty2.out.hperm = std::vector<T>(); L.sty1.out.hperm = std::vector<T>();
    }
    return 0;
}

// stack the FiLM Linears (StylizationBlock.emb_layers.1, [2D, E]) of several blocks into one weight
template <typename T>
int Denoiser<T>::film_from(const std::map<std::string, HostTensor>& w, const std::vector<std::string>& prefixes, Lin& L,
                           int D) {
    const int E = cfg.time_embed_dim();
    std::vector<float> W((size_t)prefixes.size() * 2 * D * E), B((size_t)prefixes.size() * 2 * D);
    for (size_t i = 0; i < prefixes.size(); ++i) {
        const HostTensor* wi = find(w, prefixes[i] + ".emb_layers.1.weight");
        const HostTensor* bi = find(w, prefixes[i] + ".emb_layers.1.bias");
        if (!wi || !bi) return -1;
        DSH_REQUIRE((int64_t)wi->numel() == (int64_t)2 * D * E, ("shape mismatch for " + prefixes[i] + ".emb_layers.1").c_str());
        std::memcpy(&W[i * 2 * D * (size_t)E], wi->data.data(), sizeof(float) * 2 * D * E);
        std::memcpy(&B[i * 2 * D], bi->data.data(), sizeof(float) * 2 * D);
    }
    return make_lin(L, W.data(), B.data(), (int)prefixes.size() * 2 * D, E);
}

template <typename T>
int Denoiser<T>::encoder_from(const std::map<std::string, HostTensor>& w, const std::string& p0, Encoder& E, int cin, int P, int audio_k) {
    const std::string p = p0.empty() ? std::string() : p0 + ".";      // the single-MotionTransformer state dict has no sub-module prefix
    const int D = cfg.latent_dim, TE = cfg.time_embed_dim(), HE = cfg.hubert_enc_dim, HD_ = cfg.hubert_dim;
    E.cin = cin; E.cin_p = kpad(cin);
    if (int e = lin_from(w, p + "joint_embed", E.joint, D, cin)) return e;
    if (int e = lin_from(w, p + "audio_proj", E.aproj, cfg.aud_latent_dim, audio_k)) return e;
    if (int e = lin_from(w, p + "time_embed.0", E.te0, TE, D)) return e;
    if (int e = lin_from(w, p + "time_embed.2", E.te2, TE, TE)) return e;
    if (int e = lin_from(w, p + "pid_embed.0", E.pe0, TE, cfg.style_dim)) return e;
    if (int e = lin_from(w, p + "pid_embed.2", E.pe2, TE, TE)) return e;
    if (int e = lin_from(w, p + "out", E.out, cin, D)) return e;
    if (std::is_same<T, bf16>::value && D == 512 && cfg.ff_size == 1024) {
        if (int e = lin_from(w, p + "out", E.out_tl, cin, D, true)) return e;
        DSH_REQUIRE(E.out_tl.N <= E.cin_p, "padded `out` head wider than the output scratch");
        // joint_embed as the operand of the fused layer-0 seed (tl_embed.hip): rows pi-permuted per 32-row tile, K padded to whole fragments
        const HostTensor* jw = find(w, p + "joint_embed.weight"); if (!jw) return -1;
        const int nf = ceil_div(cin, 16);
        if (nf == 7 || nf == 9) {
            const int Kj = nf * 16;
            std::vector<T> fr((size_t)D * Kj);
            for (int r = 0; r < D; ++r) {
                const int sr = (r & ~31) + tl_weight_src_row(r & 31);
                for (int k = 0; k < Kj; ++k) fr[tl2_frag_index(Kj, r >> 5, r & 31, k)] = from_f32<T>(k < cin ? jw->data[(size_t)sr * cin + k] : 0.f);
            }
            if (int e = dalloc(&E.joint_wf, fr.size(), allocs)) return e;
            DSH_HIP_CHECK(hipMemcpy(E.joint_wf, fr.data(), fr.size() * sizeof(T), hipMemcpyHostToDevice));
            wbytes += fr.size() * sizeof(T);
            E.joint_nf = nf;
        }
    }
    {   // hubert_encoder: Conv1d(1024,128,3) + BN(eval) folded, GELU, Conv1d(128,128,3)  (transformer.py:437-442)
        const HostTensor *c1w = find(w, p + "hubert_encoder.0.weight"), *c2w = find(w, p + "hubert_encoder.3.weight"),
                         *bg = find(w, p + "hubert_encoder.1.weight"), *bb = find(w, p + "hubert_encoder.1.bias"),
                         *bm = find(w, p + "hubert_encoder.1.running_mean"), *bvv = find(w, p + "hubert_encoder.1.running_var");
        if (!c1w || !c2w || !bg || !bb || !bm || !bvv) return -1;
        DSH_REQUIRE((int64_t)c1w->numel() == (int64_t)HE * HD_ * 3 && (int64_t)c2w->numel() == (int64_t)HE * HE * 3,
                    "hubert_encoder conv shape mismatch");
        std::vector<float> W1((size_t)HE * 3 * HD_), B1(HE), W2((size_t)HE * 3 * HE);
        for (int o_ = 0; o_ < HE; ++o_) {
            const double sc = (double)bg->data[o_] / sqrt((double)bvv->data[o_] + 1e-5);
            B1[o_] = (float)((double)bb->data[o_] - (double)bm->data[o_] * sc);
            for (int c = 0; c < HD_; ++c)
                for (int tap = 0; tap < 3; ++tap)
                    W1[(size_t)o_ * 3 * HD_ + (size_t)tap * HD_ + c] = (float)(c1w->data[((size_t)o_ * HD_ + c) * 3 + tap] * sc);
            for (int c = 0; c < HE; ++c)
                for (int tap = 0; tap < 3; ++tap)
                    W2[(size_t)o_ * 3 * HE + (size_t)tap * HE + c] = c2w->data[((size_t)o_ * HE + c) * 3 + tap];
        }
        if (int e = make_lin(E.conv1, W1.data(), B1.data(), HE, 3 * HD_)) return e;
        if (int e = make_lin(E.conv2, W2.data(), nullptr, HE, 3 * HE)) return e;
    }
    {   // positional table: checkpoint buffer PE.pe [1,1200,512] (transformer.py:19-31,391)
        const HostTensor* pe = find(w, p + "PE.pe"); if (!pe) return -1;
        DSH_REQUIRE(pe->numel() % D == 0, "PE.pe shape mismatch");
        if (int e = upload_f32(&E.pe, pe->data.data(), pe->numel())) return e;
    }
    const float* null_emb = nullptr;
    if (cfg.cfg_active()) {
        const HostTensor* ne = find(w, p + "null_cond_emb"); if (!ne) return -1;
        DSH_REQUIRE((int)ne->numel() == P, "null_cond_emb shape mismatch");
        null_emb = ne->data.data();
    }
    E.layers.resize(cfg.num_layers);
    std::vector<std::string> film_p;
    for (int l = 0; l < cfg.num_layers; ++l) {
        const std::string lp = p + "temporal_decoder_blocks." + std::to_string(l);
        if (int e = layer_from(w, lp, E.layers[l], D, P, null_emb)) return e;
        film_p.push_back(lp + ".sa_block.proj_out");
        film_p.push_back(lp + ".ffn.proj_out");
    }
    if (E.layers[0].tl || f32_fuse) {
        if (int e = dalloc(&E.film_g, (size_t)2 * cfg.num_layers * D, allocs)) return e;
        if (int e = dalloc(&E.film_b, (size_t)2 * cfg.num_layers * D, allocs)) return e;
        for (int l = 0; l < cfg.num_layers; ++l)
            for (int j = 0; j < 2; ++j) {
                const LNp& ln = j ? E.layers[l].sty2.ln : E.layers[l].sty1.ln;
                DSH_HIP_CHECK(hipMemcpy(E.film_g + (size_t)(2 * l + j) * D, ln.g, sizeof(float) * D, hipMemcpyDeviceToDevice));
                DSH_HIP_CHECK(hipMemcpy(E.film_b + (size_t)(2 * l + j) * D, ln.b, sizeof(float) * D, hipMemcpyDeviceToDevice));
            }
    }
    return film_from(w, film_p, E.film, D);
}

template <typename T>
int Denoiser<T>::finalize(const std::map<std::string, HostTensor>& w) {
    DSH_REQUIRE(!finalized, "weights already finalized");
    const int D = cfg.latent_dim, TE = cfg.time_embed_dim(), DA = cfg.audio_dim;
    DSH_REQUIRE(D == 512 && cfg.num_heads == 8 && DA == 128, "kernels are specialised for latent 512 / 8 heads / audio 128");
    if (cfg.single_transformer) {
        // MotionTransformer alone (transformer.py:349-587 with opt.unidiffuser = False): the `gesture` slot holds the one encoder
        if (int e = encoder_from(w, "", ges_, cfg.channels(), D + cfg.aud_latent_dim + cfg.hubert_enc_dim, DA)) return e;
        finalized = true;
        return 0;
    }
    if (int e = lin_from(w, "time_embed.0", aud_te0, TE, D)) return e;
    if (int e = lin_from(w, "time_embed.2", aud_te2, TE, TE)) return e;
    if (int e = layer_from(w, "encoder_aud", aud, DA, 0, nullptr)) return e;
    if (int e = film_from(w, {"encoder_aud.sa_block.proj_out", "encoder_aud.ffn.proj_out"}, aud_film, DA)) return e;
    if (std::is_same<T, bf16>::value && cfg.ff_size == 1024) {
        // operands of the fused encoder_aud tail (tl_aud.hip): 18-chunk weight stream, stacked biases, stacked LayerNorm affines of its two StylizationBlocks
        const HostTensor *ws1 = find(w, "encoder_aud.sa_block.proj_out.out_layers.2.weight"), *bs1 = find(w, "encoder_aud.sa_block.proj_out.out_layers.2.bias"),
                         *w1 = find(w, "encoder_aud.ffn.linear1.weight"), *b1 = find(w, "encoder_aud.ffn.linear1.bias"),
                         *w2 = find(w, "encoder_aud.ffn.linear2.weight"), *b2 = find(w, "encoder_aud.ffn.linear2.bias"),
                         *ws2 = find(w, "encoder_aud.ffn.proj_out.out_layers.2.weight"), *bs2 = find(w, "encoder_aud.ffn.proj_out.out_layers.2.bias"),
                         *g1 = find(w, "encoder_aud.sa_block.proj_out.norm.weight"), *be1 = find(w, "encoder_aud.sa_block.proj_out.norm.bias"),
                         *g2 = find(w, "encoder_aud.ffn.proj_out.norm.weight"), *be2 = find(w, "encoder_aud.ffn.proj_out.norm.bias");
        if (!ws1 || !bs1 || !w1 || !b1 || !w2 || !b2 || !ws2 || !bs2 || !g1 || !be1 || !g2 || !be2) return -1;
        DSH_REQUIRE((int64_t)w1->numel() == (int64_t)1024 * DA && (int64_t)w2->numel() == (int64_t)DA * 1024 && (int64_t)ws1->numel() == (int64_t)DA * DA, "encoder_aud weight shapes");
        std::vector<uint16_t> st((size_t)18 * 16384);
        tl_aud_pack_stream(ws1->data.data(), w1->data.data(), w2->data.data(), ws2->data.data(), st.data());
        // audio_proj of the two motion encoders rides behind the tail as 4 more chunks each (K = [mel | aud_feat] = 256 -> 256)
        const HostTensor *ape = find(w, "encoder_exp.audio_proj.weight"), *apg = find(w, "encoder_ges.audio_proj.weight"),
                         *bpe = find(w, "encoder_exp.audio_proj.bias"), *bpg = find(w, "encoder_ges.audio_proj.bias");
        if (!ape || !apg || !bpe || !bpg) return -1;
        if (cfg.aud_latent_dim == 256 && (int64_t)ape->numel() == 256 * 256 && (int64_t)apg->numel() == 256 * 256) {
            st.resize((size_t)26 * 16384);
            tl_aud_pack_audio_proj(ape->data.data(), st.data() + (size_t)18 * 16384);
            tl_aud_pack_audio_proj(apg->data.data(), st.data() + (size_t)22 * 16384);
            std::vector<float> bap(bpe->data); bap.insert(bap.end(), bpg->data.begin(), bpg->data.end());
            if (int e = upload_f32(&aud_ap_bias, bap.data(), bap.size())) return e;
        }
        if (int e = dalloc(&aud_stream, st.size(), allocs)) return e;
        DSH_HIP_CHECK(hipMemcpy(aud_stream, st.data(), st.size() * 2, hipMemcpyHostToDevice));
        wbytes += st.size() * 2;
        std::vector<float> bb; bb.insert(bb.end(), bs1->data.begin(), bs1->data.end()); bb.insert(bb.end(), b1->data.begin(), b1->data.end());
        bb.insert(bb.end(), b2->data.begin(), b2->data.end()); bb.insert(bb.end(), bs2->data.begin(), bs2->data.end());
        DSH_REQUIRE(bb.size() == 1408, "encoder_aud bias shapes");
        if (int e = upload_f32(&aud_bias, bb.data(), bb.size())) return e;
        std::vector<float> gg(g1->data); gg.insert(gg.end(), g2->data.begin(), g2->data.end());
        std::vector<float> be(be1->data); be.insert(be.end(), be2->data.begin(), be2->data.end());
        if (int e = upload_f32(&aud_film_g, gg.data(), gg.size())) return e;
        if (int e = upload_f32(&aud_film_b, be.data(), be.size())) return e;
    }
    const int Pexp = D + cfg.aud_latent_dim + cfg.hubert_enc_dim;
    if (int e = encoder_from(w, "encoder_exp", exp_, cfg.expression_dim, Pexp, 2 * DA)) return e;
    if (int e = encoder_from(w, "encoder_ges", ges_, cfg.dim_pose, Pexp + cfg.expression_dim, 2 * DA)) return e;
    finalized = true;
    return 0;
}

// ------------------------------------------------------------------------------------------------
template <typename T>
int Denoiser<T>::ensure_workspace(int B, int T_) {
    if (B <= capB && T_ <= capT) return 0;
    DSH_HIP_CHECK(hipStreamSynchronize(st));
    for (void* p : ws_allocs) (void)hipFree(p);
    ws_allocs.clear();
    capB = std::max(B, capB); capT = std::max(T_, capT);
    // rows padded to the 128-token block of tl_linear (it does not bounds-check rows)
    // (+128: a cond-half launch starts at row r0 = B*T, which is not block aligned)
    // (the token-per-lane path keeps the two CFG halves in separately block-aligned row ranges: rows [0, Mc) and
    //  [round_up(Mc, 128), +Mc), so that a half-only launch never touches the other half)
    // (256: the K = 512 LDS-DMA kernels own 256 tokens per block)
    const size_t Bc = capB, Mc = (size_t)round_up(capB * capT, 256) + 256, M = (size_t)round_up(capB * capT, 256) * (cfg.cfg_active() ? 2 : 1) + 256;
    const int D = cfg.latent_dim, TE = cfg.time_embed_dim(), F = cfg.ff_size, L = cfg.num_layers;
    const int cinp = std::max(exp_.cin_p, ges_.cin_p);      // (exp_ is empty in single-transformer mode)
    const int Ppmax = ges_.layers[0].Pp;
    auto& P = ws_allocs;
#define WS(ptr, nelem) if (int e = dalloc(&ptr, (nelem), P)) return e
    WS(audio_f, Mc * cfg.audio_dim);
    WS(h, M * D);
    WS(o, M * cinp);
    WS(expr_x0, Mc * expr_ld());
    DSH_HIP_CHECK(hipMemsetAsync(expr_x0, 0, Mc * expr_ld() * sizeof(float), st));   // (the pad columns are never written)
    WS(expr16, Mc * 128);
    WS(film_aud_tab, Bc * aud_film.N);
    WS(aud_feat_f, Mc * cfg.audio_dim);
    WS(aud_x2, Mc * cfg.audio_dim);
    WS(aud_y, Mc * cfg.audio_dim);
    WS(temb, Bc * D);
    WS(hid, Bc * TE);
    WS(semb, Bc * TE);
    WS(pid_in, Bc * kpad(cfg.style_dim));
    WS(spk_idx, Bc);
    WS(pid_rep, Bc * cfg.style_dim);
    WS(film_small, Bc * (size_t)ges_.film.N);
    WS(audio256, Mc * 2 * cfg.audio_dim);
    if (tl_path()) { WS(aproj_rm, Mc * cfg.aud_latent_dim); WS(hub_rm, Mc * cfg.hubert_enc_dim); WS(h0, Mc * D); }
    if (tl_path() && capT > 96) { WS(qkv_rm, M * 3 * D); WS(y_rm, M * D); }
    WS(x_in, Mc * cinp);
    if (sizeof(T) != 4) { WS(h16, M * D); WS(hlo, M * D); }
    WS(n, M * D);
    WS(y, M * D);
    WS(s, M * D);
    WS(qkv, M * 3 * D);
    WS(U, Mc * Ppmax);
    WS(g, M * F);
    WS(y2, M * D);
    WS(col, Mc * 3 * cfg.hubert_dim);
    WS(z, Mc * cfg.hubert_enc_dim);
    for (Encoder* E : encs()) {
        WS(E->pid_part, Bc * TE);
        WS(E->pid_part_s, Bc * TE);
        WS(E->hub, Mc * cfg.hubert_enc_dim);
        WS(E->film_tab, Bc * (size_t)(L * 2 * 2 * D));
        WS(E->aproj_buf, Mc * cfg.aud_latent_dim);
    }
    lvl_n = 0;                                   // the timestep cache is laid out for one (B, T)
#undef WS
    return 0;
}

// the part of set_condition() that the x-independent head of an evaluation needs: mel features and the speaker embedding
template <typename T>
int Denoiser<T>::set_condition_light(int B, int T_, const float* audio, const float* person_id) {
    DSH_REQUIRE(finalized, "weights not finalized");
    DSH_REQUIRE(B > 0 && T_ > 0, "batch and frames must be positive");
    DSH_REQUIRE(audio && person_id, "null conditioning pointer");
    if (int e = ensure_workspace(B, T_)) return e;
    batch = B; frames = T_;
    lvl_n = 0;                               // cached x-independent results belong to the previous condition
    lvl_borrowed = nullptr; lvl_borrowed_n = 0;
    const int Mc = B * T_, DA = cfg.audio_dim, TE = cfg.time_embed_dim();
    // mel features: fp32 copy (encoder_aud residual stream) + left half of the [audio | aud_feat] operand
    if (int e = launch_pack_cols<T>(audio, DA, Mc, 0, DA, DA, 1.0f, audio256, 2 * DA, audio_f, DA, st)) return e;
    // speaker embedding pid_embed(person_id)  (transformer.py:453-457,559): step invariant
    // encoder_aud up to its first StylizationBlock does not depend on the timestep: once per condition instead of once per evaluation
    // (950 clips: 280 us and four launches per evaluation; DSH_AUD_HOIST=0: per evaluation, the A/B switch)
    if (!cfg.single_transformer && aud_hoist()) { if (int e = aud_front()) return e; }
    // distinct speakers: the style rows come to the host once per condition (B x style floats; the one host sync of set_condition) and are
    // compared exactly; pid_embed then runs on the distinct rows and is gathered per clip
    {
        const int S = cfg.style_dim;
        pid_host.resize((size_t)B * S);
        DSH_HIP_CHECK(hipMemcpyAsync(pid_host.data(), person_id, pid_host.size() * sizeof(float), hipMemcpyDeviceToHost, st));
        DSH_HIP_CHECK(hipStreamSynchronize(st));
        spk_idx_host.assign(B, 0);
        std::vector<int> rep;                                   // first clip of every distinct row
        for (int b = 0; b < B; ++b) {
            int j = 0;
            for (; j < (int)rep.size(); ++j)
                if (std::memcmp(&pid_host[(size_t)b * S], &pid_host[(size_t)rep[j] * S], S * sizeof(float)) == 0) break;
            if (j == (int)rep.size()) rep.push_back(b);
            spk_idx_host[b] = j;
        }
        n_spk = (int)rep.size();
        std::vector<float> reps((size_t)n_spk * S);
        for (int j = 0; j < n_spk; ++j) std::memcpy(&reps[(size_t)j * S], &pid_host[(size_t)rep[j] * S], S * sizeof(float));
        // (pageable host sources: hipMemcpyAsync stages them before it returns, the vectors may be reused by the next call)
        DSH_HIP_CHECK(hipMemcpyAsync(spk_idx, spk_idx_host.data(), (size_t)B * sizeof(int), hipMemcpyHostToDevice, st));
        DSH_HIP_CHECK(hipMemcpyAsync(pid_rep, reps.data(), reps.size() * sizeof(float), hipMemcpyHostToDevice, st));
        DSH_HIP_CHECK(hipStreamSynchronize(st));
    }
    if (int e = launch_pack_cols<T>(pid_rep, cfg.style_dim, n_spk, 0, cfg.style_dim, kpad(cfg.style_dim), 1.0f, pid_in,
                                    kpad(cfg.style_dim), nullptr, 0, st)) return e;
    for (Encoder* E : encs()) {
        if (int e = gemm(E->pe0, pid_in, kpad(cfg.style_dim), n_spk, ACT_SILU, false, nullptr, 0, 0, nullptr, 0, hid, TE)) return e;
        if (int e = gemm(E->pe2, hid, TE, n_spk, ACT_NONE, false, nullptr, 0, 0, E->pid_part_s, TE, nullptr, 0)) return e;
        if (int e = launch_gather_rows_f32(E->pid_part_s, TE, spk_idx, E->pid_part, TE, B, TE, st)) return e;
    }
    conditioned = true;
    light_cond = true;
    return 0;
}

template <typename T>
int Denoiser<T>::set_condition(int B, int T_, const float* audio, const float* person_id, const float* hubert) {
    DSH_REQUIRE(hubert, "null conditioning pointer");
    if (int e = set_condition_light(B, T_, audio, person_id)) return e;
    conditioned = false;
    const int Mc = B * T_, HE = cfg.hubert_enc_dim, HD_ = cfg.hubert_dim;
    for (Encoder* E : encs()) {
        // hubert_encoder over time, zero padded per window
        if (int e = launch_im2col3_rows<float, T>(hubert, HD_, B, T_, HD_, col, 3 * HD_, st)) return e;
        if (int e = gemm(E->conv1, col, 3 * HD_, Mc, ACT_GELU, false, nullptr, 0, 0, nullptr, 0, z, HE)) return e;
        if (int e = launch_im2col3_rows<T, T>(z, HE, B, T_, HE, col, 3 * HE, st)) return e;
        if (tl_path()) {
            if (int e = gemm(E->conv2, col, 3 * HE, Mc, ACT_NONE, false, nullptr, 0, 0, nullptr, 0, hub_rm, HE)) return e;
            if (int e = launch_tile_rows_bf16<T>(hub_rm, HE, Mc, HE, E->hub, HE, st)) return e;
        } else if (int e = gemm(E->conv2, col, 3 * HE, Mc, ACT_NONE, false, nullptr, 0, 0, nullptr, 0, E->hub, HE)) return e;
    }
    conditioned = true;
    light_cond = false;
    return 0;
}

// sa_block (after its LayerNorm input is known) + ffn, shared by encoder_aud (D=128) and the main layers
template <typename T>
int Denoiser<T>::run_block_tail(const Layer& L, int M, int D, int nbatch, int fr, const float* film, int film_ld,
                                int film_off0, int bmod, float* hres, T* h16o, const T* hA, const float* res_in, const T* y_in) {
    // n (LayerNorm output) is already in `n`; y_in != null: the attention output is given (encoder_aud: computed at set_condition),
    // and the block's residual input is res_in instead of hres (which then only receives the result)
    const T* yy = y_in ? y_in : y;
    if (!y_in) {
        if (int e = gemm(L.qkv, n, D, M, ACT_NONE, false, nullptr, 0, 0, nullptr, 0, qkv, 3 * D)) return e;
        if (prof) prof->begin(PROF_ATTN);
        if (int e = launch_linear_attention<T>(qkv, 3 * D, nbatch, fr, D, D / cfg.num_heads, y, D, st)) return e;
        if (prof) prof->end(4.0 * M * (double)D * (D / cfg.num_heads));
        flops_acc += 4.0 * M * (double)D * (D / cfg.num_heads);
    }
    if (int e = launch_ln_film_silu_rows<T, T>(yy, D, M, D, L.sty1.ln.g, L.sty1.ln.b, film, film_ld, film_off0, fr, bmod, s, D, st)) return e;
    if (int e = gemm(L.sty1.out, s, D, M, ACT_NONE, false, res_in ? res_in : hres, D, 0, hres, D, h16o, D)) return e;
    if (int e = gemm(L.ffn1, hA, D, M, ACT_GELU, false, nullptr, 0, 0, nullptr, 0, g, cfg.ff_size)) return e;
    if (int e = gemm(L.ffn2, g, cfg.ff_size, M, ACT_NONE, false, nullptr, 0, 0, nullptr, 0, y2, D)) return e;
    return launch_ln_film_silu_rows<T, T>(y2, D, M, D, L.sty2.ln.g, L.sty2.ln.b, film, film_ld, film_off0 + 2 * D, fr, bmod, s, D, st);
    // caller issues the final sty2.out GEMM (its destination differs between encoder_aud and the main layers)
}

// fp32 path, round 6: the same branch of a main layer with the LayerNorm / StylizationBlock fronts inside the GEMM launches (gemm_f32_pro.hip):
// q|k|v(LN(h)) -> attention -> h += Linear(sty(y)) -> ffn.linear1 / GELU -> ffn.linear2 -> h += Linear(sty(y2)); seven launches instead of ten
template <typename T>
int Denoiser<T>::run_block_tail_fused(const Layer& L, int M, int D, int nbatch, int fr, const float* film, int film_ld, int film_off0, int bmod,
                                      int has_null, int r0, bool const_done, const float* next_const) {
    const int fb = f32_now();
    const float* yf = reinterpret_cast<const float*>(y);
    const float* y2f = reinterpret_cast<const float*>(y2);
    if (has_null && !const_done) {
        // classifier-free guidance: feat_proj(null_cond_emb), a constant per layer, is added to the unconditional rows of h in place (transformer.py:
        // 326-338).  The previous layer's last launch has done it (next_const below) — except in front of layer 0 and without the fused
        // StylizationBlock launch, where the LayerNorm row kernel still does
        if (int e = launch_ln_rows<T>(h, D, M, D, L.null_const, r0, L.sa_ln.g, L.sa_ln.b, n, D, st)) return e;
        if (int e = gemm(L.qkv, n, D, M, ACT_NONE, false, nullptr, 0, 0, nullptr, 0, qkv, 3 * D)) return e;
    } else if (!(fb & 1)) {
        if (int e = launch_ln_rows<T>(h, D, M, D, nullptr, 0, L.sa_ln.g, L.sa_ln.b, n, D, st)) return e;
        if (int e = gemm(L.qkv, n, D, M, ACT_NONE, false, nullptr, 0, 0, nullptr, 0, qkv, 3 * D)) return e;
    } else if (int e = gemm_pro(L.qkv, 1, one_seg(h, D, D), D, M, ACT_NONE, nullptr, 0, 0, fr, bmod, nullptr, reinterpret_cast<float*>(qkv), 3 * D)) return e;
    // the attention branch's StylizationBlock front applied to the attention launch's output registers, once per element (a block = the eight
    // heads of a sample owns whole rows), instead of by each of the eight N tiles of the Linear's row block: the Linear then runs front-less
    const bool attn_sty = (fb & 10) == 10 && linear_attention_sty_f32_supported(fr, D, D / cfg.num_heads, 3 * D, D);
    if (prof) prof->begin(PROF_ATTN);
    if (attn_sty) {
        if (int e = launch_linear_attention_sty_f32(reinterpret_cast<const float*>(qkv), 3 * D, nbatch, fr, D, reinterpret_cast<float*>(s), D, film, film_ld, film_off0, bmod, st)) return e;
    } else if (int e = launch_linear_attention<T>(qkv, 3 * D, nbatch, fr, D, D / cfg.num_heads, y, D, st)) return e;
    if (prof) prof->end(4.0 * M * (double)D * (D / cfg.num_heads));
    flops_acc += 4.0 * M * (double)D * (D / cfg.num_heads);
    if (attn_sty) {
        if (int e = gemm(L.sty1.out, s, D, M, ACT_NONE, false, h, D, 0, h, D, nullptr, D)) return e;
    } else if (fb & 2) {
        if (int e = gemm_pro(L.sty1.out, 2, one_seg(yf, D, D), D, M, ACT_NONE, film, film_ld, film_off0, fr, bmod, h, h, D)) return e;
    } else {
        if (int e = launch_ln_film_silu_rows<T, T>(y, D, M, D, L.sty1.ln.g, L.sty1.ln.b, film, film_ld, film_off0, fr, bmod, s, D, st)) return e;
        if (int e = gemm(L.sty1.out, s, D, M, ACT_NONE, false, h, D, 0, h, D, nullptr, D)) return e;
    }
    if (int e = gemm(L.ffn1, hT(), D, M, ACT_GELU, false, nullptr, 0, 0, nullptr, 0, g, cfg.ff_size)) return e;
    if (int e = gemm(L.ffn2, g, cfg.ff_size, M, ACT_NONE, false, nullptr, 0, 0, nullptr, 0, y2, D)) return e;
    if (fb & 2) return gemm_pro(L.sty2.out, 2, one_seg(y2f, D, D), D, M, ACT_NONE, film, film_ld, film_off0 + 2 * D, fr, bmod, h, h, D, next_const, r0);
    if (int e = launch_ln_film_silu_rows<T, T>(y2, D, M, D, L.sty2.ln.g, L.sty2.ln.b, film, film_ld, film_off0 + 2 * D, fr, bmod, s, D, st)) return e;
    return gemm(L.sty2.out, s, D, M, ACT_NONE, false, h, D, 0, h, D, nullptr, D);
}

// x-independent part of one motion encoder's evaluation: emb = time_embed(temb(t)) + pid_embed(pid) -> SiLU -> the stacked
// FiLM Linears (transformer.py:555-559, :77), and audio_proj([audio | aud_feat]) (:574).  Inputs: temb, audio256.
template <typename T>
int Denoiser<T>::prep_encoder(Encoder& E) {
    const int B = batch, D = cfg.latent_dim, TE = cfg.time_embed_dim(), Mc = B * frames;
    const int film_ld = E.film.N;
    // emb = time_embed(temb(t)) + pid_embed(pid); only SiLU(emb) is ever consumed (StylizationBlock.emb_layers)
    // (round 6) on the distinct (timestep, speaker) rows: with one timestep for the whole batch those are the distinct speakers (row j =
    // speaker j, temb rows 0 .. n_spk - 1 all hold that timestep); otherwise every clip is its own row
    const int R = emb_rows();
    if (int e = gemm(E.te0, temb, D, R, ACT_SILU, false, nullptr, 0, 0, nullptr, 0, hid, TE)) return e;
    if (int e = gemm(E.te2, hid, TE, R, ACT_SILU, true, t_uniform ? E.pid_part_s : E.pid_part, TE, 0, nullptr, 0, semb, TE)) return e;
    if (int e = gemm(E.film, semb, TE, R, ACT_NONE, false, nullptr, 0, 0, film_small, film_ld, nullptr, 0)) return e;
    if (int e = launch_film_expand(film_small, film_ld, t_uniform ? spk_idx : nullptr, E.film_tab, B, 2 * cfg.num_layers, D, E.film_g, E.film_b,
                                   (E.layers[0].tl || (f32_now() & 2)) ? 1 : 0, st)) return e;
    if (E.layers[0].tl && aproj_in_tail) return 0;        // audio_proj was a stage of the encoder_aud launch (tl_aud.hip)
    if (E.layers[0].tl) {
        // (K = E.aproj.K: [audio | aud_feat] under UniDiffuser, the 128 mel features of the left half for a single transformer)
        if (int e = gemm(E.aproj, audio256, 2 * cfg.audio_dim, Mc, ACT_NONE, false, nullptr, 0, 0, nullptr, 0, aproj_rm, cfg.aud_latent_dim)) return e;
        return launch_tile_rows_bf16<T>(aproj_rm, cfg.aud_latent_dim, Mc, cfg.aud_latent_dim, E.aproj_buf, cfg.aud_latent_dim, st);
    }
    return gemm(E.aproj, audio256, 2 * cfg.audio_dim, Mc, ACT_NONE, false, nullptr, 0, 0, nullptr, 0, E.aproj_buf, cfg.aud_latent_dim);
}

template <typename T>
int Denoiser<T>::run_encoder(Encoder& E, const float* x, int c0, int w, const float* expr, int expr_w, const float* c1,
                             const float* c2, float* eps, bool want_x0) {
    const int B = batch, fr = frames, D = cfg.latent_dim, C = cfg.channels();
    const bool tlp = E.layers[0].tl;
    // token-per-lane path: tiled activations; the conditional half starts at the next 256-row block after the null half
    const int Mc = B * fr, has_null = cfg.cfg_active() ? 1 : 0, r0 = has_null ? (tlp ? round_up(Mc, 256) : Mc) : 0;
    const int M = r0 + Mc;
    const int film_ld = E.film.N;
    T* const aproj = E.aproj_buf;
    // h = joint_embed(x) + PE[:T]; the CFG halves start identical
    float* hc = h + (size_t)r0 * D;           // (r0 is a multiple of 32 on the tiled path: same offset arithmetic)
    T* hc16 = sizeof(T) == 4 ? nullptr : h16 + (size_t)r0 * D;
    const char* jfe = getenv("DSH_JOINT_FUSE");          // (read per evaluation: the tests flip it inside one process)
    const bool joint_fuse = !(jfe && atoi(jfe) == 0);
    if (tlp && hilo && joint_fuse && E.joint_wf) {
        // round 6: joint_embed + bias + PE + CFG-null constant + plane split in ONE launch from the tiled bf16 channels of x (tl_embed.hip)
        if (int e = launch_tile_rows_bf16<float>(x + c0, C, Mc, w, x_in, E.joint_nf * 16, st)) return e;
        if (int e = launch_tl_joint(x_in, E.joint_nf, E.joint_wf, E.joint.b, E.pe, fr, has_null ? E.layers[0].null_const : nullptr, Mc, r0, h16, hlo, st)) return e;
        flops_acc += 2.0 * Mc * (double)D * E.cin;
    } else if (tlp) {
        if (int e = launch_pack_cols<T>(x, C, Mc, c0, w, E.cin_p, 1.0f, x_in, E.cin_p, nullptr, 0, st)) return e;
        // null half = cond half + feat_proj_0(null_cond_emb); one pass seeds the tiled fp32 stream and its bf16 shadow
        if (int e = gemm(E.joint, x_in, E.cin_p, Mc, ACT_NONE, false, E.pe, D, fr, h0, D, nullptr, 0)) return e;
        if (int e = launch_seed_stream(h0, Mc, D, E.layers[0].null_const, has_null, r0, h, h16, st, hilo ? hlo : nullptr)) return e;
    } else {
        if (int e = launch_pack_cols<T>(x, C, Mc, c0, w, E.cin_p, 1.0f, x_in, E.cin_p, nullptr, 0, st)) return e;
        if (int e = gemm(E.joint, x_in, E.cin_p, Mc, ACT_NONE, false, E.pe, D, fr, hc, D, nullptr, D)) return e;
        if (has_null) DSH_HIP_CHECK(hipMemcpyAsync(h, hc, (size_t)Mc * D * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    for (int l = 0; l < cfg.num_layers; ++l) {
        const Layer& L = E.layers[l];
        ConcatSegs sg;
        sg.p0 = hc; sg.ld0 = D; sg.w0 = D;
        sg.p1 = aproj; sg.ld1 = cfg.aud_latent_dim; sg.w1 = cfg.aud_latent_dim;
        sg.p2 = E.hub; sg.ld2 = cfg.hubert_enc_dim; sg.w2 = cfg.hubert_enc_dim;
        sg.p3 = expr; sg.ld3 = expr_ld(); sg.w3 = expr ? expr_w : 0;
        if (!L.tl && !(f32_now() & 1)) { if (int e = launch_concat_ln_rows<T>(sg, Mc, L.ln0.g, L.ln0.b, U, L.Pp, L.Pp, st)) return e; }
        if (L.tl) {
            // feat_proj.0 LayerNorm over the un-materialised concat is the register prologue of feat_proj.1
            if (!(dbg_skip & 1)) if (int e = tl(L.f1, 3, hc16, Mc, ACT_SILU, &L.ln0, nullptr, 0, 0, fr, B, nullptr, nullptr, g, nullptr, 0,
                           aproj, E.hub, expr ? expr16 : nullptr, L.P)) return e;
            if (hilo) {
                T* hlc = hlo + (size_t)r0 * D;
                if (!(dbg_skip & 2)) if (int e = tl(L.f3, 0, g, Mc, ACT_NONE, nullptr, nullptr, 0, 0, fr, B, reinterpret_cast<const float*>(hc16), nullptr, hc16, nullptr, 0,
                               nullptr, nullptr, nullptr, 0, 0x7fffffff, 0, hlc, hlc)) return e;
            } else if (int e = tl(L.f3, 0, g, Mc, ACT_NONE, nullptr, nullptr, 0, 0, fr, B, hc, hc, hc16, nullptr, 0)) return e;
        } else if (f32_now() & 1) {
            // feat_proj.0 LayerNorm over the un-materialised concat: folded into feat_proj.1, moments taken in its staging (gemm_f32_pro.hip)
            GemmProArgs cs{};
            cs.seg[0] = hc; cs.seg_ld[0] = D; cs.seg_end[0] = D / 32;
            cs.seg[1] = reinterpret_cast<const float*>(aproj); cs.seg_ld[1] = cfg.aud_latent_dim; cs.seg_end[1] = cs.seg_end[0] + cfg.aud_latent_dim / 32;
            cs.seg[2] = reinterpret_cast<const float*>(E.hub); cs.seg_ld[2] = cfg.hubert_enc_dim; cs.seg_end[2] = cs.seg_end[1] + cfg.hubert_enc_dim / 32;
            cs.seg[3] = expr; cs.seg_ld[3] = expr_ld(); cs.seg_end[3] = L.Pp / 32;
            DSH_REQUIRE(cfg.aud_latent_dim % 32 == 0 && cfg.hubert_enc_dim % 32 == 0 && L.Pp == 32 * cs.seg_end[2] + (expr ? expr_ld() : 0), "concat segments must be whole K tiles");
            if (int e = gemm_pro(L.f1, 1, cs, L.P, Mc, ACT_SILU, nullptr, 0, 0, fr, B, nullptr, reinterpret_cast<float*>(g), 2 * D)) return e;
            if (int e = gemm(L.f3, g, 2 * D, Mc, ACT_NONE, false, hc, D, 0, hc, D, nullptr, 0)) return e;
        } else {
            if (int e = gemm(L.f1, U, L.Pp, Mc, ACT_SILU, false, nullptr, 0, 0, nullptr, 0, g, 2 * D)) return e;
            if (int e = gemm(L.f3, g, 2 * D, Mc, ACT_NONE, false, hc, D, 0, hc, D, nullptr, 0)) return e;
        }
        if (L.tl) {
            // bf16 path: LayerNorm / FiLM / SiLU live in the register prologue of the token-per-lane Linear;
            // the CFG-null constant of the NEXT layer is folded into this layer's last epilogue (layer 0:
            // seed_stream above), so no row kernel touches h between the GEMMs.
            const int nb = B * (1 + has_null), hr0 = has_null ? r0 : 0x7fffffff;
            if (!(dbg_skip & 4)) if (int e = tl(L.qkv, 1, h16, M, ACT_NONE, &L.sa_ln, nullptr, 0, 0, fr, B, nullptr, nullptr, qkv, nullptr, 0)) return e;
            if (prof) prof->begin(PROF_ATTN);
            if (dbg_skip & 8) {
            } else if (fr <= 96) {
                if (int e = launch_linear_attention_tiled(qkv, nb, B, r0, fr, D, y, st, M >= 4096 ? next_rev() : 0)) return e;
            } else {
                // windows longer than the MFMA kernel's 96-frame tile (non-default n_poses): row-major VALU kernel
                // between two layout conversions per CFG half
                for (int hf = 0; hf <= has_null; ++hf) {
                    const size_t ro = hf ? (size_t)r0 : 0, rm = (size_t)hf * Mc;
                    if (int e = launch_untile_rows_bf16(qkv + ro * 3 * D, 3 * D, Mc, 3 * D, qkv_rm + rm * 3 * D, 3 * D, st)) return e;
                }
                if (int e = launch_linear_attention<T>(qkv_rm, 3 * D, nb, fr, D, D / cfg.num_heads, y_rm, D, st)) return e;
                for (int hf = 0; hf <= has_null; ++hf) {
                    const size_t ro = hf ? (size_t)r0 : 0, rm = (size_t)hf * Mc;
                    if (int e = launch_tile_rows_bf16<T>(y_rm + rm * D, D, Mc, D, y + ro * D, D, st)) return e;
                }
            }
            const double afl = 4.0 * Mc * (1 + has_null) * (double)D * (D / cfg.num_heads);
            if (prof) prof->end(afl);
            flops_acc += afl;
            // round 5: at whole-chip token counts the attention branch's StylizationBlock is the first stage of the fused FFN launch
            // (tl3_ffn_kernel<..., STY>, DSH_FFN_STY=0: separate launch as before)
            const bool sty_fused = ffn_sty && hilo && ffn_fuse && ffn_ver == 3 && L.ffn_stream && tl3_ffn_supported(M, fr, B, true);
            if (sty_fused) {
            } else if (hilo) {
                if (!(dbg_skip & 16)) if (int e = tl(L.sty1.out, 2, y, M, ACT_NONE, &L.sty1.ln, E.film_tab, film_ld, l * 4 * D, fr, B, reinterpret_cast<const float*>(h16), nullptr, h16,
                               nullptr, 0, nullptr, nullptr, nullptr, 0, hr0, 0, hlo, hlo)) return e;
            } else if (int e = tl(L.sty1.out, 2, y, M, ACT_NONE, &L.sty1.ln, E.film_tab, film_ld, l * 4 * D, fr, B, h, h, h16, nullptr, 0,
                                  nullptr, nullptr, nullptr, 0, hr0)) return e;
            const float* next_const = (has_null && l + 1 < cfg.num_layers) ? E.layers[l + 1].null_const : nullptr;
            if (ffn_fuse && L.ffn_stream && (ffn_ver == 3 ? tl3_ffn_supported(M, fr, B, hilo) : tl2_ffn_supported(M, fr, B))) {
                // ffn.linear1 -> GELU -> ffn.linear2 -> StylizationBlock -> + h in ONE kernel: hidden and y2 stay in registers
                Tl2FfnArgs c;
                c.X = h16; c.Wffn = L.ffn_stream + FFN_STREAM_OFF; c.b1 = L.ffn1.b; c.b2 = L.ffn2.b; c.b3 = L.sty2.out.b;
                c.Y = nullptr; c.bs1 = nullptr; c.film_off1 = 0;
                c.film = E.film_tab; c.film_ld = film_ld; c.film_off = l * 4 * D + 2 * D; c.frames = fr; c.bmod = B; c.half_row0 = hr0;
                c.R = h; c.Cf = h; c.Ct = h16; c.row_const = next_const; c.n_const_rows = Mc; c.M = M; c.trace = nullptr; c.clk = nullptr;
                c.Rhi = nullptr; c.Rlo = nullptr; c.Clo = nullptr;
                if (hilo) { c.R = nullptr; c.Cf = nullptr; c.Rhi = h16; c.Rlo = hlo; c.Clo = hlo; }
                if (sty_fused) { c.X = nullptr; c.Y = y; c.bs1 = L.sty1.out.b; c.film_off1 = l * 4 * D; c.Wffn = L.ffn_stream; }
                c.rev = next_rev();
                const double fl = 2.0 * M * (double)(2.0 * D * cfg.ff_size + (double)D * D) + (sty_fused ? 2.0 * M * (double)D * D : 0.0);
                // (hi / lo planes: the input IS the hi plane of the residual; tl3_ffn_kernel re-reads it for 6 of the 16 Linear3 tiles only)
                static const bool ffn_keep_hi = [] {
                    const char* e = getenv("DSH_FFN_PB"); const char* pc = getenv("DSH_FFN_PC");
                    return (!e || (atoi(e) & 1)) && (!pc || atoi(pc) == 1);
                }();
                const double by = (double)M * (hilo ? (D * 2 + D * 2 + D * 2 * (ffn_keep_hi ? 6.0 / 16 : 1.0) + D * 4) : (D * 2 + D * 4 * 2 + D * 2)) + (double)(2.0 * D * cfg.ff_size + (double)D * D) * 2;
                flops_acc += fl;
                if (prof) prof->begin(PROF_TL_FFN);
                const int rc = (dbg_skip & 32) ? 0 : (ffn_ver == 3 ? launch_tl3_ffn(c, st) : launch_tl2_ffn(c, st));
                if (prof) prof->end(fl, by);
                if (notify_ev && ++tl_launches == notify_at) DSH_HIP_CHECK(hipEventRecord(notify_ev, st));
                if (rc) return rc;
                continue;
            }
            if (int e = tl(L.ffn1, 0, h16, M, ACT_GELU, nullptr, nullptr, 0, 0, fr, B, nullptr, nullptr, g, nullptr, 0)) return e;
            if (int e = tl(L.ffn2, 0, g, M, ACT_NONE, nullptr, nullptr, 0, 0, fr, B, nullptr, nullptr, y2, nullptr, 0)) return e;
            if (hilo) {
                if (int e = tl(L.sty2.out, 2, y2, M, ACT_NONE, &L.sty2.ln, E.film_tab, film_ld, l * 4 * D + 2 * D, fr, B, reinterpret_cast<const float*>(h16), nullptr,
                               h16, next_const, Mc, nullptr, nullptr, nullptr, 0, hr0, 0, hlo, hlo)) return e;
            } else if (int e = tl(L.sty2.out, 2, y2, M, ACT_NONE, &L.sty2.ln, E.film_tab, film_ld, l * 4 * D + 2 * D, fr, B, h, h, h16,
                                  next_const, Mc, nullptr, nullptr, nullptr, 0, hr0)) return e;
        } else if (f32_now()) {
            // (the CFG-null constant of layer l + 1 rides in this layer's last epilogue when that is the fused StylizationBlock launch)
            const bool hand_over = has_null && (f32_now() & 3) == 3;
            const float* nc = (hand_over && l + 1 < cfg.num_layers) ? E.layers[l + 1].null_const : nullptr;
            if (int e = run_block_tail_fused(L, M, D, B * (1 + has_null), fr, E.film_tab, film_ld, l * 4 * D, B, has_null, r0, hand_over && l > 0, nc)) return e;
        } else {
            if (int e = launch_ln_rows<T>(h, D, M, D, has_null ? L.null_const : nullptr, r0, L.sa_ln.g, L.sa_ln.b, n, D, st)) return e;
            if (int e = run_block_tail(L, M, D, B * (1 + has_null), fr, E.film_tab, film_ld, l * 4 * D, B, h, h16_out(), hT())) return e;
            if (int e = gemm(L.sty2.out, s, D, M, ACT_NONE, false, h, D, 0, h, D, h16_out(), D)) return e;
        }
    }
    const char* ofe = getenv("DSH_OUT_FUSE");           // (read per evaluation: the tests flip it inside one process)
    if (tlp && hilo && E.out_tl.wf && (E.out_tl.N == 128 || E.out_tl.N == 160) && (ofe && atoi(ofe) != 0)) {
        // round 6: out head of both halves + CFG mix + expression x0 (+ its tiled copy) in ONE launch (tl_out.hip).  Built, bit-identical
        // (test_fused_output_head_is_bit_identical) and measured SLOWER in every regime — 530.3 vs 528.8 ms per 950-clip step, 11.5 k vs
        // 11.96 k frames/s on the 32-chain stream, 91.7 k vs 92.4 k at 100 clips (profiles/r06_r_*, r06_s_*): one wave per 32 tokens runs
        // both halves and every output tile back to back (one wave per SIMD, 320 registers), where the three launches it replaces spread the
        // same work over the chip.  Off unless DSH_OUT_FUSE=1.
        flops_acc += 2.0 * M * (double)E.out_tl.N * D;
        return launch_tl_out_mix(h16, E.out_tl.wf, E.out_tl.b, E.out_tl.N, Mc, r0, has_null, fr, w, c0, C, cfg.cond_scale, eps, x, c1, c2,
                                 want_x0 ? expr_x0 : nullptr, want_x0 ? expr16 : nullptr, st);
    }
    if (tlp) {
        if (int e = tl(E.out_tl, 0, h16, M, ACT_NONE, nullptr, nullptr, 0, 0, fr, B, nullptr, o, nullptr, nullptr, 0,
                       nullptr, nullptr, nullptr, 0, 0x7fffffff, E.cin_p)) return e;
    } else if (int e = gemm(E.out, hT(), D, M, ACT_NONE, false, nullptr, 0, 0, o, E.cin_p, nullptr, 0)) return e;
    if (int e = launch_cfg_mix(o, E.cin_p, Mc, r0, fr, w, has_null, cfg.cond_scale, eps, C, c0, x, C, c1, c2,
                               want_x0 ? expr_x0 : nullptr, expr_ld(), st)) return e;
    // tiled bf16 copy of the expression x0, zero padded to 128 columns: last segment of the gesture encoder's concat rows
    if (want_x0 && tlp) return launch_tile_rows_bf16<float>(expr_x0, expr_ld(), Mc, w, expr16, 128, st);
    return 0;
}

// x-independent head of an evaluation: timestep embedding and encoder_aud — one D=128 layer on 2*audio with
// UniDiffuser.time_embed (transformer.py:730-739); writes the right half of audio256 = [audio | aud_feat]
template <typename T>
int Denoiser<T>::prep_audio(const int64_t* t) {
    const int B = batch, fr = frames, D = cfg.latent_dim, DA = cfg.audio_dim, TE = cfg.time_embed_dim(), Mc = B * fr;
    if (int e = launch_temb_rows<T>(t, emb_rows(), D, temb, D, st)) return e;
    if (cfg.single_transformer) return 0;             // no encoder_aud: audio_proj reads the mel features (left half of audio256)
    // encoder_aud's embedding has no speaker term (transformer.py:730): with one timestep for the batch it is ONE row, which the
    // FiLM consumers address as (clip % 1)
    const int Ra = t_uniform ? 1 : B;
    if (int e = gemm(aud_te0, temb, D, Ra, ACT_SILU, false, nullptr, 0, 0, nullptr, 0, hid, TE)) return e;
    if (int e = gemm(aud_te2, hid, TE, Ra, ACT_SILU, false, nullptr, 0, 0, nullptr, 0, semb, TE)) return e;
    if (int e = gemm(aud_film, semb, TE, Ra, ACT_NONE, false, nullptr, 0, 0, film_aud_tab, aud_film.N, nullptr, 0)) return e;
    const char* afe = getenv("DSH_AUD_FUSE");          // (read per evaluation: the tests flip it inside one process)
    const bool aud_fuse = !(afe && atoi(afe) == 0);
    if (aud_stream && aud_fuse) {
        // round 6: everything behind the attention in ONE launch (tl_aud.hip) from the per-condition x = 2 audio and attention output
        if (!aud_hoist()) { if (int e = aud_front()) return e; }
        if (int e = launch_film_fold(film_aud_tab, aud_film.N, Ra, 2, DA, aud_film_g, aud_film_b, st)) return e;
        flops_acc += 2.0 * Mc * (2.0 * DA * DA + 2.0 * DA * cfg.ff_size);
        // ... and, on the token-per-lane path, audio_proj([mel | aud_feat]) of both motion encoders straight into their tiled concat operands
        const char* ape = getenv("DSH_APROJ_FUSE");
        // (built and measured: 533.5 / 539.3 vs 533.6 / 535.0 ms per 950-clip step, profiles/r06_n_ab_aproj_fuse.txt — the two small GEMMs hide
        //  under the other sub-batch streams, the longer launch does not; DSH_APROJ_FUSE=1 turns it on)
        aproj_in_tail = aud_ap_bias && tl_path() && (ape && atoi(ape) != 0);
        if (aproj_in_tail) flops_acc += 2.0 * 2.0 * Mc * 256.0 * 256.0;
        if (int e = launch_tl_aud_tail(aud_y, aud_x2, aud_stream, aud_bias, film_aud_tab, aud_film.N, Ra, fr, Mc, aud_feat_f, audio256 + DA, 2 * DA, st,
                                       aproj_in_tail ? 2 : 0, aud_ap_bias, exp_.aproj_buf, ges_.aproj_buf)) return e;
        // audio_proj of both motion encoders as ONE token-per-lane launch writing the tiled operands (instead of 2 x (GEMM + tile_rows))
        const char* a2 = getenv("DSH_APROJ_TL");
        if (!aproj_in_tail && aud_ap_bias && tl_path() && !(a2 && atoi(a2) == 0)) {
            if (int e = launch_tl_aproj(audio256, aud_stream + (size_t)18 * 16384, aud_ap_bias, 2, exp_.aproj_buf, ges_.aproj_buf, Mc, st)) return e;
            flops_acc += 2.0 * 2.0 * Mc * 256.0 * 256.0;
            aproj_in_tail = true;                          // (prep_encoder skips its own audio_proj)
        }
        return 0;
    }
    aproj_in_tail = false;
    float* ha = h;                       // [Mc,128] fp32 residual stream of encoder_aud (reuses h)
    T* ha16 = sizeof(T) == 4 ? nullptr : h16;
    const T* haA = sizeof(T) == 4 ? reinterpret_cast<const T*>(ha) : ha16;
    // (x = 2 audio and the attention output y come from set_condition(): run_block_tail starts behind the attention)
    if (!aud_hoist()) { if (int e = aud_front()) return e; }
    if (int e = run_block_tail(aud, Mc, DA, B, fr, film_aud_tab, aud_film.N, 0, Ra, ha, ha16, haA, aud_x2, aud_y)) return e;
    // audio_emb <- cat(audio_emb, aud_feat): right half of the audio_proj operand (+ fp32 tap for tests)
    return gemm(aud.sty2.out, s, DA, Mc, ACT_NONE, false, ha, DA, 0, aud_feat_f, DA, audio256 + DA, 2 * DA);
}

template <typename T>
int Denoiser<T>::level_cache_prepare(int n_levels) {
    DSH_REQUIRE(conditioned, "set_condition() must precede level_cache_prepare()");
    if (n_levels <= 0) return -1;
    const size_t film = (size_t)batch * ges_.film.N * sizeof(float), ap = (size_t)round_up(batch * frames, 32) * cfg.aud_latent_dim * sizeof(T);
    const size_t stride = (cfg.single_transformer ? 1 : 2) * (film + ap);
    if (stride * (size_t)n_levels > ((size_t)8 << 30)) return -1;          // <= 8 GiB of slots (a 317-clip sub-batch: 1.75 GiB; the part has 288 GB)
    if (stride * (size_t)n_levels > lvl_cap) {
        // the prefetch instance writes these slots from its own stream (adopt_level_slots): a run that aborted mid-schedule may
        // still have side-stream work in flight, so the whole device is drained before the slots are released (rare: grow-only)
        DSH_HIP_CHECK(hipDeviceSynchronize());
        if (lvl_slots) (void)hipFree(lvl_slots);
        lvl_slots = nullptr; lvl_cap = 0;
        DSH_HIP_CHECK(hipMalloc((void**)&lvl_slots, stride * (size_t)n_levels));
        lvl_cap = stride * (size_t)n_levels;
    }
    lvl_stride = stride; lvl_n = n_levels;
    return 0;
}

template <typename T>
int Denoiser<T>::level_copy(const int64_t* level, int restore) {
    const size_t film = (size_t)batch * ges_.film.N * sizeof(float), ap = (size_t)round_up(batch * frames, 32) * cfg.aud_latent_dim * sizeof(T);
    const std::vector<Encoder*> es = encs();
    char* slots = lvl_borrowed ? lvl_borrowed : lvl_slots;
    const size_t stride = lvl_borrowed ? lvl_borrowed_stride : lvl_stride;
    DSH_REQUIRE(level && (lvl_borrowed ? lvl_borrowed_n : lvl_n) > 0 && stride == es.size() * (film + ap), "timestep cache not prepared for this condition");
    LevelCopyArgs a;
    a.nseg = 0;
    for (size_t i = 0; i < es.size(); ++i) {
        if (part != 0 && (int)i != part - 1) continue;      // (a partial evaluation touches its own encoder's share only; es = {exp, ges})
        a.work[a.nseg] = reinterpret_cast<char*>(es[i]->film_tab); a.bytes[a.nseg] = film; a.off[a.nseg] = i * (film + ap); ++a.nseg;
        a.work[a.nseg] = reinterpret_cast<char*>(es[i]->aproj_buf); a.bytes[a.nseg] = ap; a.off[a.nseg] = i * (film + ap) + film; ++a.nseg;
    }
    a.slots = slots; a.stride = stride; a.level = level; a.restore = restore;
    return launch_level_copy(a, st);
}

template <typename T>
int Denoiser<T>::eval_level(const float* x, const int64_t* t, const float* c1, const float* c2, float* eps, int mode, const int64_t* level) {
    DSH_REQUIRE(conditioned, "set_condition() must precede eval()");
    DSH_REQUIRE(mode >= 0 && mode <= 3, "eval_level: unknown mode");
    DSH_REQUIRE(t && (mode == 3 || (x && c1 && c2 && eps)), "null pointer");
    DSH_REQUIRE(mode == 3 || !light_cond, "this instance only holds the x-independent conditioning (prefetch instance)");
    DSH_REQUIRE(part == 0 || ((mode == 2 || mode == 0) && !cfg.single_transformer), "a partial evaluation computes (mode 0) or restores (mode 2) its own head");
    flops_acc = 0;
    tl_launches = 0;
    if (mode == 3) {
        if (int e = prep_audio(t)) return e;
        for (Encoder* E : encs()) { if (int e = prep_encoder(*E)) return e; }
        return level_copy(level, 0);
    }
    if (mode == 2) {
        if (int e = level_copy(level, 1)) return e;
    } else {
        // (a partial evaluation computes the shared head — timestep embedding, encoder_aud — and its own encoder's share)
        if (int e = prep_audio(t)) return e;
        for (Encoder* E : encs()) { if (part == 0 || E == (part == 1 ? &exp_ : &ges_)) { if (int e = prep_encoder(*E)) return e; } }
        if (mode == 1) { if (int e = level_copy(level, 0)) return e; }
    }
    // ---- expression, then gesture conditioned on the expression x0 estimate (transformer.py:741-768)
    const int E_ = cfg.expression_dim, G_ = cfg.dim_pose;
    if (cfg.single_transformer) {
        if (int e = run_encoder(ges_, x, 0, cfg.channels(), nullptr, 0, c1, c2, eps, false)) return e;
    } else {
        if (part != 2) { if (int e = run_encoder(exp_, x, G_, E_, nullptr, 0, c1, c2, eps, true)) return e; }
        if (part != 1) { if (int e = run_encoder(ges_, x, 0, G_, expr_x0, E_, c1, c2, eps, false)) return e; }
    }
    flops_last_eval = flops_acc;
    return 0;
}

template <typename T>
int Denoiser<T>::debug_copy(const std::string& what, float* out) {
    DSH_REQUIRE(conditioned && out, "debug_copy before eval");
    DSH_REQUIRE(!cfg.single_transformer, "debug taps (aud_feat / expr_x0) exist only in the UniDiffuser model");
    const size_t Mc = (size_t)batch * frames;
    if (what == "aud_feat") {
        DSH_HIP_CHECK(hipMemcpyAsync(out, aud_feat_f, Mc * cfg.audio_dim * sizeof(float), hipMemcpyDeviceToDevice, st));
    } else if (what == "expr_x0") {
        DSH_HIP_CHECK(hipMemcpy2DAsync(out, cfg.expression_dim * sizeof(float), expr_x0, expr_ld() * sizeof(float), cfg.expression_dim * sizeof(float), Mc,
                                       hipMemcpyDeviceToDevice, st));
    } else {
        set_last_error("unknown debug tap '" + what + "'");
        return -1;
    }
    return 0;
}

// Large batches are evaluated as two or three (DSH_DUAL=n: at most n) independent sub-batches on as many streams (the extra instances
// share the weights).  Clips never interact inside the denoiser, so this changes no result; the second stream starts a few
// launches late, which keeps the two kernel sequences out of phase: an HBM-bound StylizationBlock launch of one
// sub-batch then shares the chip with an MFMA-bound q|k|v / FFN launch of the other, and one launch's load prologue
// and tail run under the other's main loop (sty + ffn.linear2 pair: 566 -> 492 us, scripts/bench_tl_overlap.py).
class DualDenoiser final : public DenoiserBase {
  public:
    DualDenoiser(DenoiserBase* primary, const ModelConfig& c, hipStream_t s) : cfg_(c), st_(s) {
        inst_.emplace_back(primary);
        const char* e = getenv("DSH_DUAL");
        nsplit_ = e ? std::max(1, std::min(8, atoi(e) == 1 ? 2 : atoi(e))) : 3;      // 0 / 1-> off is "0"; n >= 2: at most n streams
        if (e && atoi(e) == 0) nsplit_ = 1;
        const char* l = getenv("DSH_DUAL_LAG");
        lag_ = l ? atoi(l) : 3;
        // fp32 parity path: one GEMM launch of the config-2 batch (8704 rows) is only 1 - 3 rounds of co-resident tiles, so the
        // second stream's launches fill the partial last rounds (+1.5 % measured); bf16: from 12288 rows (want_split)
        // fp32 path: two streams from 4096 rows, three from 8700 (the 256-clip BEAT batch of configs[1]: 8704 rows = 17 x 512, every
        // Linear a 1.06 / 2.1 / 3.2-round launch of 64 x 64 tiles whose tail another sub-batch fills; measured 24.8 k -> 25.1 k frames/s)
        if (c.precision == 0) { min_rows_ = 4096; rows_per_stream_ = 2900; }
        const char* rs = getenv("DSH_DUAL_ROWS");
        if (rs && atoi(rs) > 0) rows_per_stream_ = (size_t)atoi(rs);
        const char* mr = getenv("DSH_DUAL_MIN_ROWS");
        if (mr && atoi(mr) > 0) min_rows_ = (size_t)atoi(mr);
        const char* pr = getenv("DSH_PIPE_ROWS");
        if (pr) pipe_rows_ = (size_t)atol(pr);
    }
    ~DualDenoiser() override {
        (void)hipDeviceSynchronize();                        // side streams may still be running a level ahead of an aborted loop
        twin_.reset();
        if (twin_stream_) (void)hipStreamDestroy(twin_stream_);
        if (twin_ev_) (void)hipEventDestroy(twin_ev_);
        for (Prefetch& f : pf_) f.prep.reset();              // the side instances borrow the sub-batch instances' slots and weights
        while (inst_.size() > 1) inst_.pop_back();
        for (hipStream_t st : streams_) (void)hipStreamDestroy(st);
        for (hipEvent_t ev : events_) (void)hipEventDestroy(ev);
        if (cond_buf_) (void)hipFree(cond_buf_);
        for (Prefetch& f : pf_) {
            f.prep.reset();
            if (f.stream) (void)hipStreamDestroy(f.stream);
            for (hipEvent_t ev : f.lvl_ev) (void)hipEventDestroy(ev);
            if (f.ev_fork) (void)hipEventDestroy(f.ev_fork);
            if (f.ev_done) (void)hipEventDestroy(f.ev_done);
            if (f.t_dev) (void)hipFree(f.t_dev);
        }
    }
    int finalize(const std::map<std::string, HostTensor>& w) override { return inst_[0]->finalize(w); }
    int set_condition(int B, int T, const float* audio, const float* person_id, const float* hubert) override {
        DSH_REQUIRE(B > 0 && T > 0 && audio && person_id && hubert, "set_condition: null conditioning pointer / empty batch");
        // The split (one stream vs sub-batches on several) may change between evals (profiler on/off), which re-runs the
        // per-instance set_condition: the conditioning is therefore copied into context-owned buffers, so the caller's
        // tensors only need to stay valid until this call's work on the context stream has been enqueued (stream order).
        const size_t na = (size_t)B * T * cfg_.audio_dim, np = (size_t)B * cfg_.style_dim, nh = (size_t)B * T * cfg_.hubert_dim;
        // the prefetch instance reads the previous conditioning on its own stream: order the overwrite behind it
        for (Prefetch& f : pf_)
            if (f.busy) { DSH_HIP_CHECK(hipStreamWaitEvent(st_, f.ev_done, 0)); f.busy = false; }
        if (twin_busy_) { DSH_HIP_CHECK(hipEventRecord(twin_ev_, twin_stream_)); DSH_HIP_CHECK(hipStreamWaitEvent(st_, twin_ev_, 0)); twin_busy_ = false; }
        twin_cond_ok_ = false;
        if (!inst_.empty()) (void)inst_[0]->set_part(0);
        if (na + np + nh > cond_cap_) {
            DSH_HIP_CHECK(hipStreamSynchronize(st_));
            if (cond_buf_) (void)hipFree(cond_buf_);
            cond_buf_ = nullptr; cond_cap_ = 0;
            DSH_HIP_CHECK(hipMalloc(&cond_buf_, (na + np + nh) * sizeof(float)));
            cond_cap_ = na + np + nh;
        }
        float* a = cond_buf_; float* p = a + na; float* h = p + np;
        DSH_HIP_CHECK(hipMemcpyAsync(a, audio, na * sizeof(float), hipMemcpyDeviceToDevice, st_));
        DSH_HIP_CHECK(hipMemcpyAsync(p, person_id, np * sizeof(float), hipMemcpyDeviceToDevice, st_));
        DSH_HIP_CHECK(hipMemcpyAsync(h, hubert, nh * sizeof(float), hipMemcpyDeviceToDevice, st_));
        cond_ = {B, T, a, p, h};
        batch = B; frames = T;
        return apply_condition((B == sticky_B_ && T == sticky_T_ && pipe_possible()) ? 1 : want_split(B, T));
    }
    int eval(const float* x, const int64_t* t, const float* c1, const float* c2, float* eps) override {
        DSH_REQUIRE(cond_.B > 0, "set_condition() must precede eval()");
        inst_[0]->prof = prof;
        const int ns = (loop_unsplit_ && split_now_ == 1) ? 1 : want_split(cond_.B, cond_.T);
        if (ns != split_now_) { if (int e = apply_condition(ns)) return e; }   // e.g. the profiler was switched on in between
        for (auto& in : inst_) in->t_uniform = t_uniform;
        if (ns == 1) return inst_[0]->eval(x, t, c1, c2, eps);
        const int C = cfg_.channels();
        DSH_HIP_CHECK(hipEventRecord(ev_fork_, st_));
        for (int i = 0; i < ns; ++i) {
            const int b0 = first_clip(i, ns);
            const size_t off = (size_t)b0 * cond_.T * C;
            hipStream_t si = i == 0 ? st_ : streams_[i - 1];
            if (i > 0) {
                DSH_HIP_CHECK(hipStreamWaitEvent(si, ev_fork_, 0));
                DSH_HIP_CHECK(hipStreamWaitEvent(si, ev_lag_[i - 1], 0));      // a few launches behind sub-batch i - 1
            }
            if (i + 1 < ns) inst_[i]->notify_after_launches(ev_lag_[i], lag_);
            else inst_[i]->notify_after_launches(nullptr, 0);
            if (int e = inst_[i]->eval(x + off, t + b0, c1 + b0, c2 + b0, eps + off)) return e;
            if (i > 0) {
                DSH_HIP_CHECK(hipEventRecord(ev_join_[i - 1], si));
                DSH_HIP_CHECK(hipStreamWaitEvent(st_, ev_join_[i - 1], 0));
            }
        }
        return 0;
    }
    int sub_count() const override { return (cond_.B > 0 && (split_now_ == 1 || want_split(cond_.B, cond_.T) == split_now_)) ? split_now_ : 1; }
    int sub_get(int i, DenoiserBase** inst, hipStream_t* stream, int* first, int* n) override {
        DSH_REQUIRE(i >= 0 && i < split_now_ && split_now_ > 1 && inst && stream && first && n, "sub_get: no such sub-batch");
        *inst = inst_[i].get();
        *stream = i == 0 ? st_ : streams_[i - 1];
        *first = first_clip(i, split_now_);
        *n = first_clip(i + 1, split_now_) - *first;
        inst_[i]->prof = nullptr;
        return 0;
    }
    int level_cache_prepare(int n_levels) override {
        if (cond_.B <= 0 || split_now_ != 1) return -1;
        return inst_[0]->level_cache_prepare(n_levels);
    }
    // Prefetch: a second instance (shared weights, own workspace) computes the x-independent head of every scheduled level on a
    // side stream, ahead of the loop, into the main instance's cache slots.  At launch-bound batch sizes the main chain keeps a
    // handful of CUs busy, so the side stream runs beside it: the 27 launches (0.28 ms at B = 1) leave every evaluation's
    // critical path, also for schedules that visit each level once (the first window of a chain).  DSH_LEVEL_PREFETCH=0: off.
    int level_prefetch(const int64_t* t_values_host, int n_levels, const int* order, int n_order, int begin, int sub = -1) override {
        if (n_levels <= 0 || n_order < 0 || !t_values_host || (n_order > 0 && !order)) return -1;
        // whole batch on one stream (sub < 0) or sub-batch `sub` of a split batch: the instance that evaluates, its stream, its clips
        if (sub < 0 ? (split_now_ != 1) : (sub >= split_now_ || split_now_ < 2)) return -1;
        const int mi = sub < 0 ? 0 : sub;
        DenoiserBase* main = inst_[mi].get();
        hipStream_t main_st = mi == 0 ? st_ : streams_[mi - 1];
        const int b0 = sub < 0 ? 0 : first_clip(sub, split_now_), nb = sub < 0 ? cond_.B : first_clip(sub + 1, split_now_) - b0;
        if ((int)pf_.size() <= mi) pf_.resize(mi + 1);
        Prefetch& f = pf_[mi];
        if (begin) {
            f.active = false;
            const char* off = getenv("DSH_LEVEL_PREFETCH");
            if (off && atoi(off) == 0) return -1;
            if (sub < 0 ? (level_cache_prepare(n_levels) != 0) : (main->level_cache_prepare(n_levels) != 0)) return -1;
            char* slots = nullptr; size_t stride = 0; int nslots = 0;
            if (main->level_slots(&slots, &stride, &nslots) != 0 || nslots < n_levels) return -1;
            if (!f.prep) {
                DSH_HIP_CHECK(hipStreamCreateWithFlags(&f.stream, hipStreamNonBlocking));
                DSH_HIP_CHECK(hipEventCreateWithFlags(&f.ev_fork, hipEventDisableTiming));
                DSH_HIP_CHECK(hipEventCreateWithFlags(&f.ev_done, hipEventDisableTiming));
                DenoiserBase* c = inst_[0]->clone_shared(f.stream);
                DSH_REQUIRE(c != nullptr, "weights not finalized");
                f.prep.reset(c);
            }
            while ((int)f.lvl_ev.size() < n_levels) { hipEvent_t ev; DSH_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); f.lvl_ev.push_back(ev); }
            const size_t need = (size_t)n_levels * (nb + 1);
            if (need > f.t_cap) {
                DSH_HIP_CHECK(hipStreamSynchronize(f.stream));
                if (f.t_dev) (void)hipFree(f.t_dev);
                f.t_dev = nullptr; f.t_cap = 0;
                DSH_HIP_CHECK(hipMalloc((void**)&f.t_dev, need * sizeof(int64_t)));
                f.t_cap = need;
            }
            // everything already enqueued on the evaluating stream (the conditioning copies, the previous run's last restore from
            // the slots) precedes the side stream's work
            DSH_HIP_CHECK(hipEventRecord(f.ev_fork, main_st));
            DSH_HIP_CHECK(hipStreamWaitEvent(f.stream, f.ev_fork, 0));
            if (int e = f.prep->set_condition_light(nb, cond_.T, cond_.audio + (size_t)b0 * cond_.T * cfg_.audio_dim, cond_.pid + (size_t)b0 * cfg_.style_dim)) return e;
            if (int e = f.prep->adopt_level_slots(slots, stride, nslots)) return e;
            f.levels = n_levels; f.nb = nb;
            f.active = true;
        }
        DSH_REQUIRE(f.active && n_levels == f.levels && nb == f.nb, "level_prefetch: no run in progress");
        int64_t* idx = f.t_dev + (size_t)n_levels * nb;                     // [n_levels] slot indices 0 .. n-1
        for (int i = 0; i < n_order; ++i) {
            const int k = order[i];
            DSH_REQUIRE(k >= 0 && k < n_levels, "level_prefetch: level out of range");
            int64_t* tk = f.t_dev + (size_t)k * nb;
            if (int e = launch_fill_i64(tk, t_values_host[k], (size_t)nb, f.stream)) return e;
            if (int e = launch_fill_i64(idx + k, (int64_t)k, 1, f.stream)) return e;
            f.prep->t_uniform = emb_dedup_enabled();                        // (tk was just filled with one value)
            if (int e = f.prep->eval_level(nullptr, tk, nullptr, nullptr, nullptr, 3, idx + k)) return e;
            DSH_HIP_CHECK(hipEventRecord(f.lvl_ev[k], f.stream));
        }
        DSH_HIP_CHECK(hipEventRecord(f.ev_done, f.stream));
        f.busy = true;
        return 0;
    }
    int level_prefetch_cancel(int sub = -1) override {
        const int mi = sub < 0 ? 0 : sub;
        if (mi >= (int)pf_.size()) return 0;
        Prefetch& f = pf_[mi];
        // whatever the side stream has queued — including the part of a level_prefetch() call that failed midway, which never reached its
        // own ev_done record — is ordered in front of the evaluating stream's inline fallback
        if (f.stream && f.ev_done) {
            DSH_HIP_CHECK(hipEventRecord(f.ev_done, f.stream));
            DSH_HIP_CHECK(hipStreamWaitEvent(mi == 0 ? st_ : streams_[mi - 1], f.ev_done, 0));
        }
        f.busy = false;
        f.active = false;
        return 0;
    }
    // ---- pipelined small-batch loop (denoiser.h): the gesture-side twin of the whole-batch instance -----------------------------
    int pipe_begin(DenoiserBase** twin, hipStream_t* stream) override {
        const char* off = getenv("DSH_PIPE");
        if ((off && atoi(off) == 0) || cfg_.single_transformer || cond_.B <= 0 || split_now_ != 1 || !twin || !stream) return -1;
        // (DDIM loops: the twin restores its head from the slots the prefetch run fills; loops without a timestep cache — DDPM — compute it)
        char* slots = nullptr; size_t stride = 0; int nslots = 0;
        const bool have_slots = !pf_.empty() && pf_[0].active && inst_[0]->level_slots(&slots, &stride, &nslots) == 0;
        if (!twin_) {
            // (default priority: a lowest-priority gesture stream was measured — single clip 16.6 -> 77 ms, profiles/r06b_ar_ab_twin_stream_priority.txt)
            DSH_HIP_CHECK(hipStreamCreateWithFlags(&twin_stream_, hipStreamNonBlocking));
            DSH_HIP_CHECK(hipEventCreateWithFlags(&twin_ev_, hipEventDisableTiming));
            DenoiserBase* c = inst_[0]->clone_shared(twin_stream_);
            DSH_REQUIRE(c != nullptr, "weights not finalized");
            twin_.reset(c);
            twin_cond_ok_ = false;
        }
        // everything already enqueued on the evaluating stream (the conditioning copies) precedes the twin's work
        DSH_HIP_CHECK(hipEventRecord(twin_ev_, st_));
        DSH_HIP_CHECK(hipStreamWaitEvent(twin_stream_, twin_ev_, 0));
        if (!twin_cond_ok_) {
            if (int e = twin_->set_part(0)) return e;
            if (int e = twin_->set_condition(cond_.B, cond_.T, cond_.audio, cond_.pid, cond_.hubert)) return e;
            twin_cond_ok_ = true;
        }
        if (have_slots) { if (int e = twin_->adopt_level_slots(slots, stride, nslots)) return e; }
        if (int e = twin_->set_part(2)) return e;
        if (int e = inst_[0]->set_part(1)) return e;
        twin_->t_uniform = t_uniform; inst_[0]->t_uniform = t_uniform;
        twin_->prof = nullptr;
        twin_busy_ = true;
        *twin = twin_.get(); *stream = twin_stream_;
        return 0;
    }
    int pipe_end() override {
        if (twin_) (void)twin_->set_part(0);
        return inst_[0]->set_part(0);
    }
    int import_expr(DenoiserBase* src, hipStream_t s) override {
        // (called on the CONTEXT's denoiser with src = the twin: the twin takes the whole-batch instance's expression estimate)
        DSH_REQUIRE(twin_ && src == twin_.get(), "import_expr: no pipelined run in progress");
        return twin_->import_expr(inst_[0].get(), s);
    }
    int loop_begin(int kind) override {
        if (cond_.B <= 0) return 0;
        const bool unsplit = (kind == 0 || kind == 1) && pipe_possible() && (size_t)cond_.B * cond_.T <= pipe_rows_;
        if (!unsplit) { sticky_B_ = sticky_T_ = 0; return 0; }
        sticky_B_ = cond_.B; sticky_T_ = cond_.T; loop_unsplit_ = true;
        if (split_now_ != 1) return apply_condition(1);
        return 0;
    }
    int loop_end() override { loop_unsplit_ = false; return 0; }
    int gesture_channels() const override { return cfg_.single_transformer ? -1 : cfg_.dim_pose; }
    int level_wait_stream(int level, hipStream_t s) override {
        DSH_REQUIRE(!pf_.empty() && level >= 0 && level < (int)pf_[0].lvl_ev.size(), "level_wait_stream: level out of range");
        DSH_HIP_CHECK(hipStreamWaitEvent(s, pf_[0].lvl_ev[level], 0));
        return 0;
    }
    int level_wait(int level, int sub = -1) override {
        const int mi = sub < 0 ? 0 : sub;
        DSH_REQUIRE(mi < (int)pf_.size() && level >= 0 && level < (int)pf_[mi].lvl_ev.size(), "level_wait: level out of range");
        DSH_HIP_CHECK(hipStreamWaitEvent(mi == 0 ? st_ : streams_[mi - 1], pf_[mi].lvl_ev[level], 0));
        return 0;
    }
    int eval_level(const float* x, const int64_t* t, const float* c1, const float* c2, float* eps, int mode, const int64_t* level) override {
        if (mode == 0) return eval(x, t, c1, c2, eps);
        DSH_REQUIRE(cond_.B > 0 && split_now_ == 1, "eval_level: the timestep cache is a single-stream feature");
        inst_[0]->prof = prof;
        inst_[0]->t_uniform = t_uniform;
        return inst_[0]->eval_level(x, t, c1, c2, eps, mode, level);
    }
    double issued_flops_per_eval() const override {
        double f = 0;
        for (int i = 0; i < split_now_; ++i) f += inst_[i]->issued_flops_per_eval();
        return f;
    }
    size_t weight_bytes() const override { return inst_[0]->weight_bytes(); }
    int debug_copy(const std::string& what, float* out) override {
        DSH_REQUIRE(split_now_ == 1, "debug taps are only available on single-stream (small-batch) evaluations");
        return inst_[0]->debug_copy(what, out);
    }

  private:
    struct Cond { int B = 0, T = 0; const float* audio = nullptr; const float* pid = nullptr; const float* hubert = nullptr; };
    int want_split(int B, int T) const {
        if (nsplit_ < 2 || (prof && prof->on) || (size_t)B * T < min_rows_) return 1;
        // Two streams from min_rows_ token rows, three from 3 * rows_per_stream_ = 64 500 (the 950-clip batch of configs[2]: 83 600).
        // Measured on MI355X with the sampling loop's free-running sub-batch streams (sampler.hip; round 3), frames/s at
        // 1 / 2 / 3 streams: 100 clips (8.8k rows) 83.0k / 81.6k / -; 200 clips 87.6k / 112.4k / 107.4k; 475 clips 111.6k / 129.6k /
        // 125.8k; 650 clips - / 134.8k / 133.2k; 800 clips - / 129.9k / 134.9k; 950 clips - / 133.0k / 136.2k (four: 121k; a disjoint
        // CU partition per stream through hipExtStreamCreateWithCUMask:
        // 100 - 118k).  With one fork / join per EVALUATION (dsh_eval) the same three streams give 130k at 950 clips.
        const int by_rows = std::max(2, (int)((size_t)B * T / rows_per_stream_));
        return std::min(std::min(nsplit_, by_rows), B);
    }
    // sub-batch i covers clips [first_clip(i), first_clip(i + 1)): equal shares, or (bench experiment) the cumulative boundaries of
    // DSH_SPLIT_AT="c1,c2,..." when they fit the batch and the stream count
    int first_clip(int i, int ns) const {
        static const std::vector<int> at = [] {
            std::vector<int> v;
            if (const char* e = getenv("DSH_SPLIT_AT")) { for (const char* p = e; *p;) { v.push_back(atoi(p)); while (*p && *p != ',') ++p; if (*p == ',') ++p; } }
            return v;
        }();
        if ((int)at.size() == ns - 1 && i > 0 && i < ns) {
            bool ok = at[0] > 0 && at.back() < cond_.B;
            for (size_t k = 1; k < at.size(); ++k) ok = ok && at[k] > at[k - 1];
            if (ok) return at[i - 1];
        }
        return (int)((int64_t)cond_.B * i / ns);
    }
    int apply_condition(int ns) {
        while ((int)inst_.size() < ns) {
            hipStream_t st; hipEvent_t lag, join;
            DSH_HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
            streams_.push_back(st);
            DSH_HIP_CHECK(hipEventCreateWithFlags(&lag, hipEventDisableTiming)); events_.push_back(lag); ev_lag_.push_back(lag);
            DSH_HIP_CHECK(hipEventCreateWithFlags(&join, hipEventDisableTiming)); events_.push_back(join); ev_join_.push_back(join);
            if (!ev_fork_) { DSH_HIP_CHECK(hipEventCreateWithFlags(&ev_fork_, hipEventDisableTiming)); events_.push_back(ev_fork_); }
            DenoiserBase* c = inst_[0]->clone_shared(st);
            DSH_REQUIRE(c != nullptr, "weights not finalized");
            inst_.emplace_back(c);
        }
        split_now_ = ns;
        inst_[0]->prof = prof;
        if (ns == 1) return inst_[0]->set_condition(cond_.B, cond_.T, cond_.audio, cond_.pid, cond_.hubert);
        DSH_HIP_CHECK(hipEventRecord(ev_fork_, st_));
        for (int i = 0; i < ns; ++i) {
            const int b0 = first_clip(i, ns), nb = first_clip(i + 1, ns) - b0;
            const size_t ft = (size_t)b0 * cond_.T;
            hipStream_t si = i == 0 ? st_ : streams_[i - 1];
            if (i > 0) DSH_HIP_CHECK(hipStreamWaitEvent(si, ev_fork_, 0));
            if (int e = inst_[i]->set_condition(nb, cond_.T, cond_.audio + ft * cfg_.audio_dim, cond_.pid + (size_t)b0 * cfg_.style_dim,
                                                cond_.hubert + ft * cfg_.hubert_dim)) return e;
            if (i > 0) {
                DSH_HIP_CHECK(hipEventRecord(ev_join_[i - 1], si));
                DSH_HIP_CHECK(hipStreamWaitEvent(st_, ev_join_[i - 1], 0));
            }
        }
        return 0;
    }
    std::vector<std::unique_ptr<DenoiserBase>> inst_;      // [0] owns the weights; the others share them
    ModelConfig cfg_;
    hipStream_t st_;
    std::vector<hipStream_t> streams_;
    std::vector<hipEvent_t> events_, ev_lag_, ev_join_;
    hipEvent_t ev_fork_ = nullptr;
    Cond cond_;
    float* cond_buf_ = nullptr;                            // context-owned copy of [audio | person_id | hubert]
    size_t cond_cap_ = 0;
    // side-stream producers of the x-independent head (level_prefetch): [0] the whole batch / sub-batch 0, [i] sub-batch i
    struct Prefetch {
        std::unique_ptr<DenoiserBase> prep;                // shared weights, own workspace, conditioned with mel features + speaker only
        hipStream_t stream = nullptr;
        std::vector<hipEvent_t> lvl_ev;
        hipEvent_t ev_fork = nullptr, ev_done = nullptr;
        int64_t* t_dev = nullptr; size_t t_cap = 0;
        bool busy = false, active = false;
        int levels = 0, nb = 0;
    };
    std::vector<Prefetch> pf_;
    std::unique_ptr<DenoiserBase> twin_;                   // gesture-side twin of inst_[0] for the pipelined small-batch loop (pipe_begin)
    hipStream_t twin_stream_ = nullptr; hipEvent_t twin_ev_ = nullptr; bool twin_cond_ok_ = false, twin_busy_ = false;
    size_t pipe_rows_ = 64499;                             // DDIM loops below this many token rows: one batch, two encoder streams (loop_begin)
    int sticky_B_ = 0, sticky_T_ = 0;                      // shape whose last loop ran unsplit: set_condition conditions it as one batch
    bool loop_unsplit_ = false;                            // a sampling loop is running this batch unsplit on purpose (its evaluations must not re-split it)
    bool pipe_possible() const {
        // (the pipelined loop restores every head from the slots a side-stream prefetch run fills: all three switches must be on)
        for (const char* k : {"DSH_PIPE", "DSH_LEVEL_PREFETCH", "DSH_LEVEL_CACHE"}) { const char* v = getenv(k); if (v && atoi(v) == 0) return false; }
        return !cfg_.single_transformer && !(prof && prof->on);
    }
    int nsplit_ = 2, split_now_ = 1, lag_ = 3;
    size_t min_rows_ = 12288;                              // batches below this many token rows run on one stream
    size_t rows_per_stream_ = 21500;                       // streams = rows / this (at least two, at most DSH_DUAL): three from 64 500 rows
};

}  // namespace

DenoiserBase* make_denoiser(const ModelConfig& cfg, hipStream_t stream) {
    DenoiserBase* d = nullptr;
    if (cfg.precision == 0) d = new Denoiser<float>(cfg, stream);
    else if (cfg.precision == 1) d = new Denoiser<bf16>(cfg, stream);
    if (!d) return nullptr;
    return new DualDenoiser(d, cfg, stream);
}

}  // namespace dsh
