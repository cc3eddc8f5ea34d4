// Host side of the sampling loops: coefficient tables, respacing, the RePaint jump schedule and
// the DDIM / harmonize / DDPM drivers.  Everything is enqueued on the context stream; there are
// no host<->device syncs inside a loop (the reference syncs several times per step:
// `True in mask`, `noise_weight[0,0,0] < 0.2`, per-step H2D table copies — SURVEY.md §3.1).
//
//   tables        GaussianDiffusion.__init__      models/gaussian_diffusion.py:334-390 (fp64)
//   respacing     space_timesteps/SpacedDiffusion models/respace.py:7-34,68-82
//   schedule      get_schedule_jump_cjm_ddim      models/scheduler.py:178-208
//   loops         ddim_sample_loop(+harmonize)    models/gaussian_diffusion.py:1106-1278
//                 p_sample_loop_progressive       models/gaussian_diffusion.py:923-974
#include "sampler.h"

#include <math.h>
#include <stdlib.h>

namespace dsh {

void build_tables(const std::vector<double>& betas, DiffusionTables& t) {
    const size_t n = betas.size();
    t.betas = betas;
    t.ac.resize(n); t.ac_prev.resize(n); t.c1.resize(n); t.c2.resize(n); t.post_var.resize(n);
    t.post_logvar.resize(n); t.coef1.resize(n); t.coef2.resize(n);
    double cp = 1.0;
    for (size_t i = 0; i < n; ++i) {
        t.ac_prev[i] = cp;
        cp *= (1.0 - betas[i]);
        t.ac[i] = cp;
    }
    for (size_t i = 0; i < n; ++i) {
        t.c1[i] = sqrt(1.0 / t.ac[i]);
        t.c2[i] = sqrt(1.0 / t.ac[i] - 1.0);
        t.post_var[i] = betas[i] * (1.0 - t.ac_prev[i]) / (1.0 - t.ac[i]);
        t.coef1[i] = betas[i] * sqrt(t.ac_prev[i]) / (1.0 - t.ac[i]);
        t.coef2[i] = (1.0 - t.ac_prev[i]) * sqrt(1.0 - betas[i]) / (1.0 - t.ac[i]);
    }
    for (size_t i = 0; i < n; ++i) t.post_logvar[i] = log(t.post_var[i == 0 ? (n > 1 ? 1 : 0) : i]);
}

std::vector<double> linear_betas(int n) {
    // np.linspace(scale*1e-4, scale*0.02, n): start + i*step with step = (stop-start)/(n-1), last = stop
    const double scale = 1000.0 / n, b0 = scale * 1e-4, b1 = scale * 0.02;
    std::vector<double> b(n);
    const double step = n > 1 ? (b1 - b0) / (n - 1) : 0.0;
    for (int i = 0; i < n; ++i) b[i] = b0 + i * step;
    if (n > 1) b[n - 1] = b1;
    return b;
}

int make_tables(int steps, int respacing, DiffusionTables& out, std::string& err) {
    if (steps < 2) { err = "diffusion_steps must be >= 2"; return -1; }
    DiffusionTables base;
    build_tables(linear_betas(steps), base);
    if (respacing <= 0) {
        out = base;
        out.tmap.resize(steps);
        for (int i = 0; i < steps; ++i) out.tmap[i] = i;
        return 0;
    }
    // 'ddimK': first integer stride whose range(0, steps, stride) has exactly K entries
    int stride = 0;
    for (int s = 1; s < steps; ++s)
        if ((steps + s - 1) / s == respacing) { stride = s; break; }
    if (!stride) { err = "cannot create exactly " + std::to_string(respacing) + " steps with an integer stride"; return -1; }
    std::vector<int> tmap;
    std::vector<double> nb;
    double last = 1.0;
    for (int i = 0; i < steps; i += stride) {
        nb.push_back(1.0 - base.ac[i] / last);
        last = base.ac[i];
        tmap.push_back(i);
    }
    build_tables(nb, out);
    out.tmap = tmap;
    return 0;
}

std::vector<int> jump_schedule(int respacing, int jump_length, int jump_n_sample) {
    const int t_T = respacing == 25 ? 15 : (int)(respacing * 0.6);
    std::vector<int> jumps(t_T > 0 ? t_T : 1, 0);
    if (jump_length > 0)
        for (int j = 0; j < t_T - jump_length; j += jump_length) jumps[j] = jump_n_sample - 1;
    std::vector<int> ts;
    int t = t_T;
    while (t >= 1) {
        t -= 1;
        ts.push_back(t);
        if (t < (int)jumps.size() && jumps[t] > 0) {
            jumps[t] -= 1;
            for (int i = 0; i < jump_length; ++i) { t += 1; ts.push_back(t); }
        }
    }
    ts.push_back(-1);
    return ts;
}

// ------------------------------------------------------------------------------------------------
static int plan_steps(const SamplerOpts& o, bool masked, std::vector<SamplerStep>& steps, std::string& err) {
    steps.clear();
    if (o.kind == 1) {  // DDPM ancestral, full chain
        for (int t = o.diffusion_steps - 1; t >= 0; --t) steps.push_back({STEP_DDPM, t});
        return 0;
    }
    if (o.kind != 0) { err = "unknown sampler kind"; return -1; }
    if (masked && !o.no_repaint) {
        const std::vector<int> times = o.no_resample ? jump_schedule(o.respacing, 1, 1)
                                                     : jump_schedule(o.respacing, o.jump_length, o.jump_n_sample);
        for (size_t i = 0; i + 1 < times.size(); ++i) {
            const int t_last = times[i], t_cur = times[i + 1];
            if (t_last < 0 || t_last >= o.respacing) { err = "jump schedule leaves the spaced range"; return -1; }
            steps.push_back({t_cur < t_last ? STEP_DDIM : STEP_UNDO, t_last});
        }
    } else {
        for (int k = o.respacing - 1; k >= 0; --k) steps.push_back({STEP_DDIM, k});
    }
    return 0;
}

int64_t sampler_num_draws(const SamplerOpts& o, bool masked, bool init_from_x) {
    std::vector<SamplerStep> steps; std::string err;
    if (plan_steps(o, masked, steps, err)) { set_last_error(err); return -1; }
    int64_t n = init_from_x ? 0 : 1;
    const bool tail_gt = masked && o.same_overlap_noisy && o.clip_idx > 0;     // that branch draws no gt noise
    for (const auto& s : steps) n += (s.kind == STEP_DDIM) ? ((masked && !tail_gt) ? 2 : 1) : 1;
    return n;
}
int64_t sampler_num_steps(const SamplerOpts& o, bool masked) {
    std::vector<SamplerStep> steps; std::string err;
    if (plan_steps(o, masked, steps, err)) { set_last_error(err); return -1; }
    return (int64_t)steps.size();
}

Sampler::~Sampler() {
    drop_graph();
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    for (hipEvent_t e : {ev_pE, ev_pC, ev_pG}) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : ev_sub) (void)hipEventDestroy(e);
    for (void* p : bufs) (void)hipFree(p);
    if (row_keys) (void)hipFree(row_keys);
    if (tails) (void)hipFree(tails);
    if (tail_tmp) (void)hipFree(tail_tmp);
    if (nz_eta) (void)hipFree(nz_eta);
}

int Sampler::set_row_keys(const uint64_t* keys_host, int n) {
    DSH_REQUIRE(n >= 0 && (n == 0 || keys_host), "set_row_keys: null key array");
    n_row_keys = 0;
    if (n == 0) return 0;
    if (n > cap_row_keys) {
        DSH_HIP_CHECK(hipStreamSynchronize(st));
        if (row_keys) (void)hipFree(row_keys);
        row_keys = nullptr; cap_row_keys = 0;
        DSH_HIP_CHECK(hipMalloc((void**)&row_keys, (size_t)n * sizeof(uint64_t)));
        cap_row_keys = n;
    }
    DSH_HIP_CHECK(hipMemcpyAsync(row_keys, keys_host, (size_t)n * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    DSH_HIP_CHECK(hipStreamSynchronize(st));       // keys_host is pageable caller memory
    n_row_keys = n;
    return 0;
}

void Sampler::drop_graph() {
    if (graph_execG) { (void)hipGraphExecDestroy(graph_execG); graph_execG = nullptr; }
    if (graphG) { (void)hipGraphDestroy(graphG); graphG = nullptr; }
    for (int m = 0; m < 3; ++m) {
        if (graph_exec[m]) { (void)hipGraphExecDestroy(graph_exec[m]); graph_exec[m] = nullptr; }
        if (graph[m]) { (void)hipGraphDestroy(graph[m]); graph[m] = nullptr; }
    }
}

// One denoiser evaluation eps = model(x, t, c1, c2).  At small batch an eval is ~170 launches of a few
// microseconds each, i.e. launch / dependency bound: the first eval of a run executes eagerly (also warms one-time
// kernel attribute setup), later ones are stream-captured into a hipGraph once per cache mode and replayed.
// All pointers (x, tbuf, c1buf, c2buf, lvlbuf, eps, the denoiser workspace and its timestep-cache slots) are fixed for
// the duration of a run; only the CONTENTS of tbuf/c1buf/c2buf/lvlbuf change between steps, so one graph per mode
// (0 plain, 1 compute + save level, 2 restore level) serves every step.
int Sampler::eval_step(DenoiserBase* den, float* x, int n_eval, bool use_graph, int mode) {
    if (!use_graph || n_eval == 0) return den->eval_level(x, tbuf, c1buf, c2buf, eps, mode, lvlbuf);
    if (!graph_exec[mode]) {
        if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) != hipSuccess) {
            (void)hipGetLastError();
            return den->eval_level(x, tbuf, c1buf, c2buf, eps, mode, lvlbuf);
        }
        const int rc = den->eval_level(x, tbuf, c1buf, c2buf, eps, mode, lvlbuf);
        hipError_t e = hipStreamEndCapture(st, &graph[mode]);
        if (rc != 0 || e != hipSuccess || graph[mode] == nullptr) {
            (void)hipGetLastError();
            drop_graph();
            return rc != 0 ? rc : den->eval_level(x, tbuf, c1buf, c2buf, eps, mode, lvlbuf);
        }
        if (hipGraphInstantiate(&graph_exec[mode], graph[mode], nullptr, nullptr, 0) != hipSuccess) {
            (void)hipGetLastError();
            drop_graph();
            return den->eval_level(x, tbuf, c1buf, c2buf, eps, mode, lvlbuf);
        }
    }
    DSH_HIP_CHECK(hipGraphLaunch(graph_exec[mode], st));
    return 0;
}

// the gesture-side evaluation of the pipelined loop (mode 2: head restored from the timestep cache), on the twin's stream
int Sampler::eval_step_twin(DenoiserBase* twin, hipStream_t s, float* x, int n_eval, bool use_graph, int mode) {
    if (!use_graph || n_eval == 0) return twin->eval_level(x, tbufG, c1bufG, c2bufG, eps, mode, lvlbufG);
    if (!graph_execG) {
        if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) {
            (void)hipGetLastError();
            return twin->eval_level(x, tbufG, c1bufG, c2bufG, eps, mode, lvlbufG);
        }
        const int rc = twin->eval_level(x, tbufG, c1bufG, c2bufG, eps, mode, lvlbufG);
        hipError_t e = hipStreamEndCapture(s, &graphG);
        if (rc != 0 || e != hipSuccess || graphG == nullptr) {
            (void)hipGetLastError();
            if (graphG) { (void)hipGraphDestroy(graphG); graphG = nullptr; }
            return rc != 0 ? rc : twin->eval_level(x, tbufG, c1bufG, c2bufG, eps, mode, lvlbufG);
        }
        if (hipGraphInstantiate(&graph_execG, graphG, nullptr, nullptr, 0) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipGraphDestroy(graphG); graphG = nullptr; graph_execG = nullptr;
            return twin->eval_level(x, tbufG, c1bufG, c2bufG, eps, mode, lvlbufG);
        }
    }
    DSH_HIP_CHECK(hipGraphLaunch(graph_execG, s));
    return 0;
}

int Sampler::ensure(size_t n, int B) {
    if (n <= cap_n && B <= cap_b) return 0;
    DSH_HIP_CHECK(hipStreamSynchronize(st));
    for (void* p : bufs) (void)hipFree(p);
    bufs.clear();
    cap_n = std::max(n, cap_n); cap_b = std::max(B, cap_b);
    auto alloc = [&](void** p, size_t bytes) -> int {
        DSH_HIP_CHECK(hipMalloc(p, bytes)); bufs.push_back(*p); return 0; };
    if (int e = alloc((void**)&eps, cap_n * sizeof(float))) return e;
    if (int e = alloc((void**)&nz1, cap_n * sizeof(float))) return e;
    if (int e = alloc((void**)&tbuf, cap_b * sizeof(int64_t))) return e;
    if (int e = alloc((void**)&c1buf, cap_b * sizeof(float))) return e;
    if (int e = alloc((void**)&c2buf, cap_b * sizeof(float))) return e;
    if (int e = alloc((void**)&lvlbuf, 8 * sizeof(int64_t))) return e;          // one per sub-batch stream
    // (pipelined loop: the gesture chain's own scalars and noise scratch, sized for the batches that loop serves — below the sub-batch split's
    //  three-stream range — whatever larger batch this context has also sampled)
    capG_n = std::min(cap_n, (size_t)100000 * channels);
    if (int e = alloc((void**)&nz1G, capG_n * sizeof(float))) return e;
    if (int e = alloc((void**)&nz_etaG, capG_n * sizeof(float))) return e;
    if (int e = alloc((void**)&tbufG, cap_b * sizeof(int64_t))) return e;
    if (int e = alloc((void**)&c1bufG, cap_b * sizeof(float))) return e;
    if (int e = alloc((void**)&c2bufG, cap_b * sizeof(float))) return e;
    if (int e = alloc((void**)&lvlbufG, 8 * sizeof(int64_t))) return e;
    return 0;
}

int Sampler::run(DenoiserBase* den, const SamplerOpts& o, float* x, bool init_from_x, const float* gt,
                 const uint8_t* mask, bool masked, const float* noise_stack, int64_t n_draws, float* trace) {
    DSH_REQUIRE(den && den->batch > 0, "set_condition() must precede sample()");
    DSH_REQUIRE(x != nullptr, "null sample buffer");
    DSH_REQUIRE(!masked || (gt && mask), "masked sampling needs gt and mask");
    DSH_REQUIRE(o.noise_mode == 0 || o.noise_mode == 1, "unknown noise mode");
    DSH_REQUIRE(!(masked && o.kind == 1), "mask-present DDPM (p_sample_loop_progressive_harmonize) is not supported");
    const int B = den->batch;
    if (int e = den->loop_begin(o.kind)) return e;          // (may re-condition a mid-size batch as one batch: the two encoder chains replace the sub-batch streams)
    den->t_uniform = emb_dedup_enabled();   // every evaluation of a sampling loop runs the whole batch at ONE timestep (launch_fill_step below)
    const size_t n = (size_t)B * den->frames * channels;
    std::vector<SamplerStep> steps; std::string err;
    if (plan_steps(o, masked, steps, err)) { set_last_error(err); return -1; }
    const int64_t need = sampler_num_draws(o, masked, init_from_x);
    if (o.noise_mode == 0) {
        DSH_REQUIRE(noise_stack != nullptr && n_draws >= need, "noise stack shorter than the loop's draw count");
    }
    // tables are cached per (steps, respacing)
    const int resp = o.kind == 1 ? 0 : o.respacing;
    if (tb_steps != o.diffusion_steps || tb_resp != resp) {
        if (make_tables(o.diffusion_steps, resp, tb, err)) { set_last_error(err); return -1; }
        tb_steps = o.diffusion_steps; tb_resp = resp;
    }
    if (int e = ensure(n, B)) return e;
    if (o.kind == 0 && o.eta != 0.f && o.noise_mode == 1 && cap_eta < n) {
        DSH_HIP_CHECK(hipStreamSynchronize(st));
        if (nz_eta) (void)hipFree(nz_eta);
        nz_eta = nullptr; cap_eta = 0;
        DSH_HIP_CHECK(hipMalloc((void**)&nz_eta, n * sizeof(float)));
        cap_eta = n;
    }

    DSH_REQUIRE(n_row_keys == 0 || o.noise_mode != 1 || (n_row_keys == B && (n / B) % 4 == 0),
                "row keys were set for a different batch size (or frames*channels is not a multiple of 4)");
    const bool per_row = o.noise_mode == 1 && n_row_keys == B;
    // --same_overlap_noisy state
    const size_t blc = (size_t)B * o.overlap_len * channels;
    const bool son = o.same_overlap_noisy != 0 && o.kind == 0;
    if (son) {
        DSH_REQUIRE(o.overlap_len > 0 && o.overlap_len <= den->frames, "same_overlap_noisy needs 0 < overlap_len <= frames");
        if (tails_blc != blc || tails_levels != o.respacing) {
            DSH_REQUIRE(o.clip_idx == 0 || !masked, "same_overlap_noisy: the saved noisy tails belong to a different batch / overlap shape");
            DSH_HIP_CHECK(hipStreamSynchronize(st));
            if (tails) (void)hipFree(tails);
            if (tail_tmp) (void)hipFree(tail_tmp);
            tails = nullptr; tail_tmp = nullptr;
            DSH_HIP_CHECK(hipMalloc((void**)&tails, blc * o.respacing * sizeof(float)));
            DSH_HIP_CHECK(hipMalloc((void**)&tail_tmp, blc * sizeof(float)));
            DSH_HIP_CHECK(hipMemsetAsync(tails, 0, blc * o.respacing * sizeof(float), st));
            tails_blc = blc; tails_levels = o.respacing;
        }
    }
    const bool tail_gt = son && masked && o.clip_idx > 0;
    int64_t draw = 0;
    const uint64_t quads = per_row ? (n / B) / 4 : (n + 3) / 4;
    const size_t row_n = n / B;                        // values per batch row
    // ---- sub-batch streams (large batches): every sub-batch runs the WHOLE loop on its own stream; one fork, one join ----------
    struct Sub { DenoiserBase* d; hipStream_t s; int b0, nb; size_t off, cnt; };
    std::vector<Sub> subs;
    {
        const int ns = (prof && prof->on) ? 1 : den->sub_count();
        if (ns > 1 && row_n % 4 == 0) {
            for (int i = 0; i < ns; ++i) {
                Sub u;
                if (int e = den->sub_get(i, &u.d, &u.s, &u.b0, &u.nb)) return e;
                u.off = (size_t)u.b0 * row_n; u.cnt = (size_t)u.nb * row_n;
                subs.push_back(u);
            }
            if (!ev_fork) DSH_HIP_CHECK(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
            while (ev_sub.size() < 2 * subs.size()) { hipEvent_t e; DSH_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); ev_sub.push_back(e); }
            DSH_HIP_CHECK(hipEventRecord(ev_fork, st));
            for (size_t i = 1; i < subs.size(); ++i) DSH_HIP_CHECK(hipStreamWaitEvent(subs[i].s, ev_fork, 0));
        } else {
            subs.push_back(Sub{den, st, 0, B, 0, n});
        }
    }
    const bool split = subs.size() > 1;
    // the next N(0,1) tensor of the loop: draw index (advanced once per draw, whatever the split) ...
    auto next_draw = [&]() -> int64_t { return draw++; };
    // ... and its values for one sub-batch (device pointer; scratch = a full-size buffer the sub-batch owns its slice of)
    auto noise_for = [&](int64_t idx, const Sub& u, float* scratch, const float** out) -> int {
        if (o.noise_mode == 0) { *out = noise_stack + (size_t)idx * n + u.off; return 0; }
        if (per_row) { if (int e = launch_philox_randn_rows(scratch + u.off, u.nb, row_n, o.seed, (uint64_t)idx * quads, row_keys + u.b0, u.s)) return e; }
        else if (int e = launch_philox_randn(scratch + u.off, u.cnt, o.seed, (uint64_t)idx * quads + u.off / 4, u.s)) return e;
        *out = scratch + u.off;
        return 0;
    };

    // pipelined small-batch loop (denoiser.h): expression encoder on the context stream, gesture encoder one step behind on the twin's
    DenoiserBase* twin = nullptr; hipStream_t sG = nullptr; bool pipe = false; int gch = 0;
    // everything between the fork above and the join below: any error leaves through `return` of this lambda, so that the sub-batch
    // streams are joined into the context stream on EVERY exit path (they write x, trace and the noisy tails the caller may free)
    auto loop = [&]() -> int {
    if (!init_from_x) {
        const int64_t idx = next_draw();
        for (const Sub& u : subs) {
            const float* z;
            if (int e = noise_for(idx, u, x, &z)) return e;
            if (z != x + u.off) DSH_HIP_CHECK(hipMemcpyAsync(x + u.off, z, u.cnt * sizeof(float), hipMemcpyDeviceToDevice, u.s));
        }
    }
    const bool do_mask = masked;
    int64_t step_idx = 0;
    // graphs only where launches dominate (a few thousand token rows), never while profiling events are recorded,
    // and never on the legacy NULL stream (it cannot be captured)
    static const size_t graph_rows = [] { const char* e = getenv("DSH_GRAPH_ROWS"); return e ? (size_t)atol(e) : (size_t)4096; }();
    const bool small = (size_t)B * den->frames <= graph_rows;
    const bool use_graph = st != nullptr && small && !(prof && prof->on) && getenv("DSH_NO_GRAPH") == nullptr;
    int n_eval = 0;
    drop_graph();
    // timestep cache (denoiser.h): worth it when the schedule revisits levels (out-painting jump schedule: 63 evaluations
    // over 16 levels); small (launch-bound) batches only.  DSH_LEVEL_CACHE=0 disables it.
    std::vector<char> level_seen;
    bool prefetched = false;
    std::vector<int> order;              // levels in first-use order
    std::vector<int64_t> tv;             // level -> model timestep
    size_t pf_next = 0;                  // order[0 .. pf_next) have been handed to the prefetch stream
    // (the side-stream head and the two-stream encoder pipeline also pay above the graph range, up to where batches are split over sub-batch
    //  streams: DSH_PIPE_ROWS, default below)
    static const size_t pipe_rows = [] { const char* e = getenv("DSH_PIPE_ROWS"); return e ? (size_t)atol(e) : (size_t)64499; }();
    const bool small_pf = (size_t)B * den->frames <= pipe_rows;
    if ((small || small_pf) && o.kind == 0 && !split) {
        std::vector<int> cnt(o.respacing, 0);
        int evals = 0, distinct = 0;
        for (const SamplerStep& sp : steps) if (sp.kind != STEP_UNDO) { ++evals; if (cnt[sp.level]++ == 0) { ++distinct; order.push_back(sp.level); } }
        const char* lc = getenv("DSH_LEVEL_CACHE");
        const bool cache_on = !(lc && atoi(lc) == 0);
        // side-stream prefetch of every scheduled level (also pays for schedules without repeats); else the inline cache
        // (one level is queued now, the others one evaluation ahead of their first use: the host never runs far in front of
        //  the main chain, and the main chain never waits for the host to finish queueing 25 levels)
        if (cache_on && st != nullptr && !(prof && prof->on) && !order.empty()) {
            tv.resize(o.respacing);
            for (int k = 0; k < o.respacing; ++k) tv[k] = (int64_t)tb.tmap[k];
            prefetched = den->level_prefetch(tv.data(), o.respacing, order.data(), 1, 1) == 0;
            if (prefetched) pf_next = 1;
        }
        if (prefetched) level_seen.assign(o.respacing, 0);
        else if (evals > distinct && cache_on && den->level_cache_prepare(o.respacing) == 0) level_seen.assign(o.respacing, 0);
        // the two encoders' chains on two streams: every evaluation restores its head from the slots the prefetch run fills (mode 2), no
        // per-step trace of the whole sample, no saved noisy tails (both need all channels of a step at once)
        gch = den->gesture_channels();
        if (prefetched && !trace && !son && nz1G && n <= capG_n && gch > 0 && gch < channels && den->pipe_begin(&twin, &sG) == 0) {
            pipe = true;
            for (hipEvent_t* e : {&ev_pE, &ev_pC, &ev_pG}) if (!*e) DSH_HIP_CHECK(hipEventCreateWithFlags(e, hipEventDisableTiming));
        }
    }
    // DDPM loops have no timestep cache (every level is visited once, 1000 of them): each chain computes its own head
    if (!pipe && o.kind == 1 && !split && small_pf && !trace && nz1G && n <= capG_n) {
        gch = den->gesture_channels();
        if (gch > 0 && gch < channels && den->pipe_begin(&twin, &sG) == 0) {
            pipe = true;
            for (hipEvent_t* e : {&ev_pE, &ev_pC, &ev_pG}) if (!*e) DSH_HIP_CHECK(hipEventCreateWithFlags(e, hipEventDisableTiming));
        }
    }
    const int twin_mode = o.kind == 0 ? 2 : 0;
    // sub-batch streams: the x-independent head of an evaluation (time / speaker / FiLM embeddings, encoder_aud, audio_proj: 23 small
    // dependent launches, each of which waits ~50 - 90 us for a free CU beside the other sub-batches' 100-us blocks) is computed by a
    // side instance per sub-batch, one evaluation ahead of its first use (round 4; the window-chain regime has done this since
    // round 2).  Falls back to the inline timestep cache for schedules that revisit levels, or to plain evaluations.
    bool split_cache = false, split_pf = false;
    if (split && o.kind == 0) {
        std::vector<int> cnt(o.respacing, 0);
        int evals = 0, distinct = 0;
        for (const SamplerStep& sp : steps) if (sp.kind != STEP_UNDO) { ++evals; if (cnt[sp.level]++ == 0) { ++distinct; order.push_back(sp.level); } }
        const char* lc = getenv("DSH_LEVEL_CACHE");
        const bool cache_on = !(lc && atoi(lc) == 0);
        // (measured on the 950-clip batch, round 4: 628.7 / 629.5 ms per step with the side streams against 624.8 / 624.6 without — at
        //  this size the chip is throughput-bound, three more streams only add contention — so it is opt-in: DSH_SPLIT_PREFETCH=1)
        const char* sp_e = getenv("DSH_SPLIT_PREFETCH");
        if (sp_e && atoi(sp_e) != 0 && cache_on && st != nullptr && !order.empty()) {
            tv.resize(o.respacing);
            for (int k = 0; k < o.respacing; ++k) tv[k] = (int64_t)tb.tmap[k];
            split_pf = true;
            size_t started = 0;
            for (; started < subs.size() && split_pf; ++started) split_pf = den->level_prefetch(tv.data(), o.respacing, order.data(), 1, 1, (int)started) == 0;
            if (split_pf) { pf_next = 1; level_seen.assign(o.respacing, 0); }
            else {
                // a later sub-batch could not start its side stream: the ones already started have a level evaluation queued that
                // writes the very cache slot the inline fallback below fills — order the sub-batch streams behind them and end those runs
                for (size_t i = 0; i < started; ++i) if (int e = den->level_prefetch_cancel((int)i)) return e;
            }
        }
        if (!split_pf && evals > distinct && cache_on) {
            split_cache = true;
            for (const Sub& u : subs) split_cache = split_cache && u.d->level_cache_prepare(o.respacing) == 0;
            if (split_cache) level_seen.assign(o.respacing, 0);
        }
    }
    const int lag = [] { const char* l = getenv("DSH_DUAL_LAG"); return l ? atoi(l) : 3; }();
    bool first_eval = true;
    for (const SamplerStep& sp : steps) {
        const int k = sp.level;
        if (sp.kind == STEP_UNDO) {
            const float beta = (float)tb.betas[k];
            const int64_t idx = next_draw();
            if (pipe) {
                // each chain undoes its own channels on its own stream (same draw, same values: the noise depends on the position only)
                const Sub uE{den, st, 0, B, 0, n}, uG{twin, sG, 0, B, 0, n};
                const float *zE, *zG;
                if (int e = noise_for(idx, uE, nz1, &zE)) return e;
                if (int e = launch_undo_step(x, zE, sqrtf(1.0f - beta), sqrtf(beta), n, st, channels, gch, channels)) return e;
                if (int e = noise_for(idx, uG, nz1G, &zG)) return e;
                if (int e = launch_undo_step(x, zG, sqrtf(1.0f - beta), sqrtf(beta), n, sG, channels, 0, gch)) return e;
            } else
            for (const Sub& u : subs) {
                const float* z;
                if (int e = noise_for(idx, u, nz1, &z)) return e;
                if (int e = launch_undo_step(x + u.off, z, sqrtf(1.0f - beta), sqrtf(beta), u.cnt, u.s)) return e;
            }
        } else {
            const float c1 = (float)tb.c1[k], c2 = (float)tb.c2[k];
            int mode = 0;
            if (!split) {
                // (pipelined: E_k overwrites the expression estimate the twin copied behind E_{k-1})
                if (pipe && n_eval > 0) DSH_HIP_CHECK(hipStreamWaitEvent(st, ev_pC, 0));
                if (int e = launch_fill_step(tbuf, c1buf, c2buf, lvlbuf, (int64_t)tb.tmap[k], c1, c2, (int64_t)k, B, st)) return e;
                if (prefetched) {
                    mode = 2;
                    if (!level_seen[k]) {
                        // first use: this level was queued one evaluation ago (or just above); queue the next new one now
                        size_t pos = 0;
                        while (pos < order.size() && order[pos] != k) ++pos;
                        const size_t want = std::min(order.size(), pos + 2);
                        if (want > pf_next) {
                            if (int e = den->level_prefetch(tv.data(), o.respacing, order.data() + pf_next, (int)(want - pf_next), 0)) return e;
                            pf_next = want;
                        }
                        if (int e = den->level_wait(k)) return e;
                        level_seen[k] = 1;
                    }
                } else if (!level_seen.empty()) { mode = level_seen[k] ? 2 : 1; level_seen[k] = 1; }
                if (int e = eval_step(den, x, n_eval++, use_graph, mode)) return e;
            } else {
                // every sub-batch on its own stream; at the very first evaluation sub-batch i + 1 starts a few launches behind
                // sub-batch i (so that the kernel sequences are out of phase from the start); afterwards the streams run free
                size_t want = pf_next;
                if (split_pf && !level_seen[k]) {                       // first use: queue the next new level on every side stream
                    size_t pos = 0;
                    while (pos < order.size() && order[pos] != k) ++pos;
                    want = std::max(pf_next, std::min(order.size(), pos + 2));
                }
                for (size_t i = 0; i < subs.size(); ++i) {
                    const Sub& u = subs[i];
                    if (int e = launch_fill_step(tbuf + u.b0, c1buf + u.b0, c2buf + u.b0, lvlbuf + i, (int64_t)tb.tmap[k], c1, c2, (int64_t)k, u.nb, u.s)) return e;
                    if (first_eval && i > 0) DSH_HIP_CHECK(hipStreamWaitEvent(u.s, ev_sub[2 * (i - 1)], 0));
                    u.d->notify_after_launches((first_eval && i + 1 < subs.size()) ? ev_sub[2 * i] : nullptr, lag);
                    if (split_pf && !level_seen[k]) {
                        if (want > pf_next) { if (int e = den->level_prefetch(tv.data(), o.respacing, order.data() + pf_next, (int)(want - pf_next), 0, (int)i)) return e; }
                        if (int e = den->level_wait(k, (int)i)) return e;
                    }
                    const int smode = split_pf ? 2 : split_cache ? (level_seen[k] ? 2 : 1) : 0;
                    u.d->t_uniform = den->t_uniform;
                    if (int e = u.d->eval_level(x + u.off, tbuf + u.b0, c1buf + u.b0, c2buf + u.b0, eps + u.off, smode, lvlbuf + i)) return e;
                    u.d->notify_after_launches(nullptr, 0);
                }
                if (split_cache || split_pf) level_seen[k] = 1;
                pf_next = want;
                first_eval = false;
                ++n_eval;
            }
            if (sp.kind == STEP_DDIM) {
                const int64_t idx1 = next_draw();                               // randn_like of the step: times sigma (= 0 at eta = 0)
                // sigma = eta sqrt((1 - abar_prev) / (1 - abar)) sqrt(1 - abar / abar_prev), fp32 like the reference's tensors
                float sigma = 0.f, coef_eps = sqrtf(1.0f - (float)tb.ac_prev[k]);
                if (o.eta != 0.f) {
                    const float ab = (float)tb.ac[k], abp1 = (float)tb.ac_prev[k];
                    sigma = (o.eta * sqrtf((1.0f - abp1) / (1.0f - ab))) * sqrtf(1.0f - ab / abp1);
                    coef_eps = sqrtf((1.0f - abp1) - sigma * sigma);
                    if (k == 0) sigma = 0.f;                                    // nonzero_mask: no noise at (spaced) t == 0; the mean keeps coef_eps
                }
                const int64_t idx2 = (do_mask && !tail_gt) ? next_draw() : -1;   // N(0,1) of the noised gt (RePaint blend)
                auto ddim_update = [&](const Sub& u, float* sc1, float* sc_eta, int c_lo, int c_hi) -> int {
                    DdimStepArgs a;
                    a.c_lo = c_lo; a.c_hi = c_hi;
                    a.x = x + u.off; a.eps = eps + u.off; a.x0_out = nullptr; a.c1 = c1; a.c2 = c2;
                    const float abp = (float)tb.ac_prev[k];
                    a.sqrt_ab_prev = sqrtf(abp);
                    a.sqrt_1m_ab_prev = sqrtf(1.0f - abp);
                    a.coef_eps = coef_eps; a.sigma = sigma; a.noise1 = nullptr;
                    if (o.eta != 0.f) {
                        // (nz1 is the scratch of this draw AND of the RePaint draw below: a second buffer only exists for eta != 0)
                        const float* z1 = nullptr;
                        if (int e = noise_for(idx1, u, sc_eta, &z1)) return e;
                        a.noise1 = z1;
                    }
                    a.mask = nullptr; a.gt = nullptr; a.noise2 = nullptr; a.blend = 0; a.clip = o.clip_denoised;
                    a.overlap_len = o.overlap_len; a.frames = den->frames; a.channels = channels; a.n = u.cnt;
                    const size_t toff = (size_t)u.b0 * o.overlap_len * channels;
                    a.tail_in = nullptr; a.tail_out = son ? tail_tmp + toff : nullptr;
                    if (do_mask) {
                        const float* z2 = nullptr;
                        if (tail_gt) a.tail_in = tails + (size_t)k * blc + toff;
                        else if (int e = noise_for(idx2, u, sc1, &z2)) return e;
                        a.mask = mask + u.off; a.gt = gt + u.off; a.noise2 = z2;
                        a.blend = (a.sqrt_1m_ab_prev < 0.2f && o.add_blend) ? 1 : 0;
                    }
                    if (int e = launch_ddim_step(a, u.s)) return e;
                    if (son) DSH_HIP_CHECK(hipMemcpyAsync(tails + (size_t)k * blc + toff, tail_tmp + toff, (size_t)u.nb * o.overlap_len * channels * sizeof(float),
                                                         hipMemcpyDeviceToDevice, u.s));
                    return 0;
                };
                if (pipe) {
                    // E_k: the expression channels advance on the context stream ...
                    if (int e = ddim_update(Sub{den, st, 0, B, 0, n}, nz1, nz_eta, gch, channels)) return e;
                    DSH_HIP_CHECK(hipEventRecord(ev_pE, st));
                    // ... G_k on the twin's: takes E_k's x0 estimate, evaluates the gesture encoder at the same level, advances the gesture channels
                    DSH_HIP_CHECK(hipStreamWaitEvent(sG, ev_pE, 0));
                    if (int e = den->import_expr(twin, sG)) return e;
                    DSH_HIP_CHECK(hipEventRecord(ev_pC, sG));
                    if (int e = launch_fill_step(tbufG, c1bufG, c2bufG, lvlbufG, (int64_t)tb.tmap[k], c1, c2, (int64_t)k, B, sG)) return e;
                    if (twin_mode == 2) { if (int e = den->level_wait_stream(k, sG)) return e; }
                    if (int e = eval_step_twin(twin, sG, x, n_eval - 1, use_graph, twin_mode)) return e;
                    if (int e = ddim_update(Sub{twin, sG, 0, B, 0, n}, nz1G, nz_etaG, 0, gch)) return e;
                } else
                for (const Sub& u : subs) { if (int e = ddim_update(u, nz1, nz_eta, 0, 0)) return e; }
            } else {
                const int64_t idx = next_draw();
                auto ddpm_update = [&](const Sub& u, float* sc, int c_lo, int c_hi) -> int {
                    const float* z;
                    if (int e = noise_for(idx, u, sc, &z)) return e;
                    DdpmStepArgs a;
                    a.x = x + u.off; a.eps = eps + u.off; a.noise = z; a.x0_out = nullptr; a.c1 = c1; a.c2 = c2;
                    a.coef1 = (float)tb.coef1[k]; a.coef2 = (float)tb.coef2[k];
                    a.sigma = k == 0 ? 0.0f : expf(0.5f * (float)tb.post_logvar[k]);
                    a.n = u.cnt; a.clip = o.clip_denoised; a.channels = channels; a.c_lo = c_lo; a.c_hi = c_hi;
                    return launch_ddpm_step(a, u.s);
                };
                if (pipe) {
                    if (int e = ddpm_update(Sub{den, st, 0, B, 0, n}, nz1, gch, channels)) return e;
                    DSH_HIP_CHECK(hipEventRecord(ev_pE, st));
                    DSH_HIP_CHECK(hipStreamWaitEvent(sG, ev_pE, 0));
                    if (int e = den->import_expr(twin, sG)) return e;
                    DSH_HIP_CHECK(hipEventRecord(ev_pC, sG));
                    if (int e = launch_fill_step(tbufG, c1bufG, c2bufG, lvlbufG, (int64_t)tb.tmap[k], c1, c2, (int64_t)k, B, sG)) return e;
                    if (int e = eval_step_twin(twin, sG, x, n_eval - 1, use_graph, twin_mode)) return e;
                    if (int e = ddpm_update(Sub{twin, sG, 0, B, 0, n}, nz1G, 0, gch)) return e;
                } else
                for (const Sub& u : subs) { if (int e = ddpm_update(u, nz1, 0, 0)) return e; }
            }
        }
        if (trace)
            for (const Sub& u : subs)
                DSH_HIP_CHECK(hipMemcpyAsync(trace + (size_t)step_idx * n + u.off, x + u.off, u.cnt * sizeof(float), hipMemcpyDeviceToDevice, u.s));
        ++step_idx;
    }
    return 0;
    };
    const int rc = loop();
    if (pipe) {
        // the gesture chain joins the context stream on every exit path; both instances go back to whole evaluations
        DSH_HIP_CHECK(hipEventRecord(ev_pG, sG));
        DSH_HIP_CHECK(hipStreamWaitEvent(st, ev_pG, 0));
        (void)den->pipe_end();
    }
    (void)den->loop_end();
    if (split)
        for (size_t i = 1; i < subs.size(); ++i) {
            DSH_HIP_CHECK(hipEventRecord(ev_sub[2 * i + 1], subs[i].s));
            DSH_HIP_CHECK(hipStreamWaitEvent(st, ev_sub[2 * i + 1], 0));
        }
    if (graph_exec[0] || graph_exec[1] || graph_exec[2]) { DSH_HIP_CHECK(hipStreamSynchronize(st)); drop_graph(); }
    return rc;
}

}  // namespace dsh
