// MFMA GEMM for the DiffSHEG denoiser:  C[M,N] = epi(A[M,K] · W[N,K]^T)
//
// Every Linear of the reference (models/transformer.py: feat_proj :284-289, query/key/value
// :106-108, StylizationBlock.out_layers :83, FFN :172-173, emb_layers :77, joint_embed/audio_proj/
// out :428-476) is an "NT" product: activations [tokens, K] and torch Linear weights [N, K] are
// both K-contiguous, so one tile row is 128 bytes for either element type (32 fp32 / 64 bf16).
//
// gfx950 mapping: 128x128 block tile, 4 waves (2x2), each wave a 64x64 sub-tile = 2x2 MFMA 32x32
// accumulators (64 acc VGPRs).  fp32 uses v_mfma_f32_32x32x2_f32 (exact fp32, 157 TF peak), bf16
// uses v_mfma_f32_32x32x16_bf16 (fp32 accumulate).  Tiles are staged global -> VGPR -> LDS with
// 16-byte accesses; LDS rows are padded 128 -> 144 bytes, which makes the ds_read_b128 fragment
// reads bank-conflict free (row stride 36 dwords: 16 rows of a lane group hit 16 distinct 4-bank
// slots).  Global loads of tile k+1 are issued before the MFMAs of tile k and written to the other
// LDS buffer afterwards (one barrier per K tile).
//
// fp32 fragment trick: v_mfma_f32_32x32x2_f32 wants A[i][k] with k = lane>>5.  Each lane reads 4
// consecutive k (one ds_read_b128) at byte offset chunk*32 + (lane>>5)*16 and issues 4 MFMAs; the
// k-slots of A and B are permuted identically, which leaves the dot product unchanged.
#include <stdlib.h>

#include "dsh_common.h"

namespace dsh {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int ROW_BYTES = GEMM_BK_BYTES;     // 128
constexpr int LDS_ROW = 144;                 // padded
// Block tile = (64 MI) x (64 NJ), MI, NJ in {1, 2}: 4 waves (2 x 2), each a (32 MI) x (32 NJ) sub-tile of MI x NJ accumulators.
// 128 x 128 is the efficient shape; the smaller ones exist for tile QUANTISATION: the fp32 parity configuration has
// M = 8704 token rows, i.e. 68 x 4 = 272 tiles of 128 x 128 for an N = 512 Linear on 256 CUs — 16 CUs get two tiles and the
// launch takes two tile times for 1.06 tiles of work per CU (53 %); 64 x 64 tiles give 1088 = 4.25 per CU -> 5 (85 %).
// launch_gemm_t picks the shape with the smallest (tiles per CU, rounded up) x tile area.
// WN = 1 halves the block along N (two waves, 128 threads): the 64 x 32 tile keeps 6 - 8 blocks resident per CU, i.e. the
// round a launch is quantised to gets finer still (a partially filled LAST round costs a full one: the dispatcher packs the
// leftover blocks onto few CUs; measured 57 % efficiency for 1.06 rounds of 64 x 64 tiles).
constexpr int gemm_lds_bytes(int MI, int NJ, int WN = 2) { return 2 * 32 * (2 * MI + WN * NJ) * LDS_ROW; }   // A+B stages, double buffered (128x128: 73,728 B)

template <typename T>
__device__ __forceinline__ void mfma_chunk(const u32x4& a, const u32x4& b, f32x16& acc);

template <>
__device__ __forceinline__ void mfma_chunk<float>(const u32x4& a, const u32x4& b, f32x16& acc) {
    const f32x4 af = __builtin_bit_cast(f32x4, a);
    const f32x4 bf = __builtin_bit_cast(f32x4, b);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, bf.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, bf.w, acc, 0, 0, 0);
}

template <>
__device__ __forceinline__ void mfma_chunk<bf16>(const u32x4& a, const u32x4& b, f32x16& acc) {
    bf16x8 av = __builtin_bit_cast(bf16x8, a);
    bf16x8 bv = __builtin_bit_cast(bf16x8, b);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
}

template <typename T, int VAR, int MI = 2, int NJ = 2, int WN = 2>
__global__ __launch_bounds__(128 * WN) void gemm_nt_kernel(GemmArgs p) {
    constexpr int NTHR = 128 * WN;                       // 2 x WN waves
    constexpr int BM = 64 * MI, BN = 32 * WN * NJ, A_LDS = BM * LDS_ROW, W_LDS = BN * LDS_ROW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = WN == 2 ? wave >> 1 : wave, wn = WN == 2 ? wave & 1 : 0;
    int bm, bn;
    if (VAR == 0) {
        bm = blockIdx.y; bn = blockIdx.x;
    } else {
        // XCD-aware mapping (block b runs on XCD b % 8, each XCD has its own L2): the N-tiles of one
        // M-tile get the same b % 8 and adjacent dispatch slots, so the A panel is fetched from HBM once.
        // With fewer than 8 M-tiles that would leave XCDs idle (M = 256 FiLM GEMM: 2 of 8 XCDs did all the work, 563 us
        // instead of ~130): there the blocks are simply dealt round-robin over the XCDs, N fastest.
        const int NT = p.nt_n, MT = p.nt_m;
        const int bid = blockIdx.x;
        if (MT >= 8) {
            const int group = bid / (8 * NT), rem = bid % (8 * NT);
            bm = group * 8 + (rem % 8);
            bn = rem / 8;
        } else {
            bm = bid / NT;
            bn = bid % NT;
        }
        if (bm >= MT) return;
    }
    const int m0 = bm * BM;
    const int n0 = bn * BN;

    const char* Ab = reinterpret_cast<const char*>(p.A);
    const char* Wb = reinterpret_cast<const char*>(p.W);
    const size_t lda_b = (size_t)p.lda * sizeof(T);
    const size_t ldw_b = (size_t)p.ldw * sizeof(T);
    const int nk = (p.K * (int)sizeof(T)) / ROW_BYTES;

    // per-thread staging coordinates: 16-byte chunks of the A (W) tile, 8 per row
    constexpr int NA = BM * 8 / NTHR, NW = BN * 8 / NTHR, NMAX = NA > NW ? NA : NW;
    const char* a_src[NA];
    const char* w_src[NW];
    int lds_off[NMAX];
#pragma unroll
    for (int i = 0; i < NMAX; ++i) {
        const int id = tid + i * NTHR;
        const int row = id >> 3, c16 = id & 7;
        if (i < NA) { int ra = m0 + row; ra = ra < p.M ? ra : p.M - 1; a_src[i] = Ab + (size_t)ra * lda_b + c16 * 16; }
        if (i < NW) { int rw = n0 + row; rw = rw < p.N ? rw : p.N - 1; w_src[i] = Wb + (size_t)rw * ldw_b + c16 * 16; }
        lds_off[i] = row * LDS_ROW + c16 * 16;
    }

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    u32x4 ra[NA], rw[NW];
#pragma unroll
    for (int i = 0; i < NA; ++i) ra[i] = *reinterpret_cast<const u32x4*>(a_src[i]);
#pragma unroll
    for (int i = 0; i < NW; ++i) rw[i] = *reinterpret_cast<const u32x4*>(w_src[i]);
    char* sA = smem;
    char* sW = smem + 2 * A_LDS;
#pragma unroll
    for (int i = 0; i < NA; ++i) *reinterpret_cast<u32x4*>(sA + lds_off[i]) = ra[i];
#pragma unroll
    for (int i = 0; i < NW; ++i) *reinterpret_cast<u32x4*>(sW + lds_off[i]) = rw[i];
    __syncthreads();

    // fragment read offsets (bytes) inside one stage
    const int frag_row = lane & 31;
    const int frag_kb = (lane >> 5) * 16;
    const int a_frag0 = (wm * 32 * MI + frag_row) * LDS_ROW + frag_kb;
    const int w_frag0 = (wn * 32 * NJ + frag_row) * LDS_ROW + frag_kb;

    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = (kt + 1) < nk;
        if (more) {
            const size_t koff = (size_t)(kt + 1) * ROW_BYTES;
#pragma unroll
            for (int i = 0; i < NA; ++i) ra[i] = *reinterpret_cast<const u32x4*>(a_src[i] + koff);
#pragma unroll
            for (int i = 0; i < NW; ++i) rw[i] = *reinterpret_cast<const u32x4*>(w_src[i] + koff);
        }
        const char* cA = sA + cur * A_LDS;
        const char* cW = sW + cur * W_LDS;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            u32x4 fa[MI], fb[NJ];
#pragma unroll
            for (int i = 0; i < MI; ++i) fa[i] = *reinterpret_cast<const u32x4*>(cA + a_frag0 + i * 32 * LDS_ROW + c * 32);
#pragma unroll
            for (int j = 0; j < NJ; ++j) fb[j] = *reinterpret_cast<const u32x4*>(cW + w_frag0 + j * 32 * LDS_ROW + c * 32);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    if (VAR == 0) mfma_chunk<T>(fa[i], fb[j], acc[i][j]);
                    else mfma_chunk<T>(fb[j], fa[i], acc[i][j]);   // D[n][m]: each lane ends up with 4 consecutive n of one row m -> 16-byte epilogue I/O
                }
        }
        if (more) {
            char* nA = sA + (cur ^ 1) * A_LDS;
            char* nW = sW + (cur ^ 1) * W_LDS;
#pragma unroll
            for (int i = 0; i < NA; ++i) *reinterpret_cast<u32x4*>(nA + lds_off[i]) = ra[i];
#pragma unroll
            for (int i = 0; i < NW; ++i) *reinterpret_cast<u32x4*>(nW + lds_off[i]) = rw[i];
        }
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: bias -> activation -> (+residual) [-> activation] -> store fp32 and/or T ----
    // MFMA 32x32 C layout: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    T* Ct = reinterpret_cast<T*>(p.Ct);
    if (VAR == 0) {
    const int col_l = lane & 31;
    const int row_l = 4 * (lane >> 5);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int col = n0 + wn * 32 * NJ + j * 32 + col_l;
        if (col >= p.N) continue;
        const float bv = p.bias ? p.bias[col] : 0.0f;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 32 * MI + i * 32 + (r & 3) + 8 * (r >> 2) + row_l;
                if (row >= p.M) continue;
                float v = acc[i][j][r] + bv;
                if (!p.act_after_res) v = apply_act(v, p.act);
                if (p.R) {
                    const int rr = p.res_mod > 0 ? (row % p.res_mod) : row;
                    v += p.R[(size_t)rr * p.ldr + col];
                }
                if (p.act_after_res) v = apply_act(v, p.act);
                if (p.Cf) p.Cf[(size_t)row * p.ldcf + col] = v;
                if (Ct) Ct[(size_t)row * p.ldct + col] = from_f32<T>(v);
            }
        }
    }
    } else {
    // D[n][m] layout: m = lane & 31, n = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const bool vec_ok = (p.N % 4 == 0) && (!p.R || p.ldr % 4 == 0) && (!p.Cf || p.ldcf % 4 == 0) && (!Ct || p.ldct % 4 == 0);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int row = m0 + wm * 32 * MI + i * 32 + (lane & 31);
        if (row >= p.M) continue;
        const int rr = p.res_mod > 0 ? (row % p.res_mod) : row;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = n0 + wn * 32 * NJ + j * 32 + 8 * q + 4 * (lane >> 5);
                if (col >= p.N) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q + e];
                if (vec_ok) {
                    if (p.bias) { const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias + col); v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w; }
                    if (!p.act_after_res) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], p.act);
                    }
                    if (p.R) { const f32x4 r4 = *reinterpret_cast<const f32x4*>(p.R + (size_t)rr * p.ldr + col); v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w; }
                    if (p.act_after_res) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], p.act);
                    }
                    if (p.Cf) { f32x4 o4; o4.x = v[0]; o4.y = v[1]; o4.z = v[2]; o4.w = v[3]; *reinterpret_cast<f32x4*>(p.Cf + (size_t)row * p.ldcf + col) = o4; }
                    if (Ct) {
                        T o[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = from_f32<T>(v[e]);
                        if (sizeof(T) == 2) *reinterpret_cast<uint2*>(Ct + (size_t)row * p.ldct + col) = *reinterpret_cast<uint2*>(o);
                        else *reinterpret_cast<f32x4*>(Ct + (size_t)row * p.ldct + col) = *reinterpret_cast<f32x4*>(o);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int c = col + e;
                        if (c >= p.N) continue;
                        float x = v[e] + (p.bias ? p.bias[c] : 0.0f);
                        if (!p.act_after_res) x = apply_act(x, p.act);
                        if (p.R) x += p.R[(size_t)rr * p.ldr + c];
                        if (p.act_after_res) x = apply_act(x, p.act);
                        if (p.Cf) p.Cf[(size_t)row * p.ldcf + c] = x;
                        if (Ct) Ct[(size_t)row * p.ldct + c] = from_f32<T>(x);
                    }
                }
            }
        }
    }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Few-row product with K split over the waves of a block (fp32 parity path at window-chain batches: 34 .. a few hundred rows).
// At these sizes gemm_nt_kernel is one round of a few blocks whose waves each walk the WHOLE K serially — K / 2 exact-fp32 MFMAs of
// 64 cycles per 32 x 32 sub-tile, 16 us for K = 1024 — while 240 CUs idle: BASELINE configs[0] (BEAT, batch 1, 1000 steps) spent
// 2.6 of its 3.9 ms per step in such launches (rocprofv3, round 4).  Here a block is ONE 32 x 32 output tile and its eight waves take
// the 128-byte K tiles round-robin (wave w: tiles w, w + 8, ...), each through its own LDS slice (coalesced 16-byte global loads,
// the same padded rows and fragment reads as gemm_nt_kernel, no block barrier inside the K loop); the eight partial accumulators
// are added in a fixed order (((w0 + w1) + w2) + ...) by wave 0, which runs the epilogue.  An eighth of the MFMAs per wave and
// N / 32 x M / 32 blocks instead of N / 64 x M / 64.  Deterministic; the summation order differs from the large-tile kernel's
// (fp32 round-off, ~1e-7 relative).  (Four waves with double-buffered slices: 2.80 ms per configs[0] step; these eight: 2.68 ms.)
constexpr int KS_NW = 8;                                    // waves per block = K slices
constexpr int KS_WAVE_LDS = 2 * 32 * LDS_ROW;               // [A | W][32 rows x 144 B] = 9,216 B per wave, single buffer
constexpr int KS_LDS = KS_NW * KS_WAVE_LDS + (KS_NW - 1) * 16 * 64 * 4;   // + the partial accumulators of waves 1..7 = 102,400 B

template <typename T>
__global__ __launch_bounds__(64 * KS_NW) void gemm_nt_ksplit_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    const char* Ab = reinterpret_cast<const char*>(p.A);
    const char* Wb = reinterpret_cast<const char*>(p.W);
    const size_t lda_b = (size_t)p.lda * sizeof(T), ldw_b = (size_t)p.ldw * sizeof(T);
    const int nk = (p.K * (int)sizeof(T)) / ROW_BYTES;
    char* sA = smem + wave * KS_WAVE_LDS;                  // this wave's staging slice
    char* sW = sA + 32 * LDS_ROW;
    const char* a_src[4];
    const char* w_src[4];
    int lds_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {                          // one wave instruction = 8 rows x 128 B
        const int id = lane + 64 * i, row = id >> 3, c16 = id & 7;
        int ra = m0 + row; ra = ra < p.M ? ra : p.M - 1;
        int rw = n0 + row; rw = rw < p.N ? rw : p.N - 1;
        a_src[i] = Ab + (size_t)ra * lda_b + c16 * 16;
        w_src[i] = Wb + (size_t)rw * ldw_b + c16 * 16;
        lds_off[i] = row * LDS_ROW + c16 * 16;
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    u32x4 ra4[4], rw4[4];
    int kt = wave;
    if (kt < nk) {
        const size_t koff = (size_t)kt * ROW_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) { ra4[i] = *reinterpret_cast<const u32x4*>(a_src[i] + koff); rw4[i] = *reinterpret_cast<const u32x4*>(w_src[i] + koff); }
    }
    const int frag0 = (lane & 31) * LDS_ROW + (lane >> 5) * 16;
    for (; kt < nk; kt += KS_NW) {
        // (one wave, one LDS queue: the writes below cannot overtake the fragment reads of the previous tile, and the reads
        //  that follow see them — no barrier)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<u32x4*>(sA + lds_off[i]) = ra4[i];
            *reinterpret_cast<u32x4*>(sW + lds_off[i]) = rw4[i];
        }
        if (kt + KS_NW < nk) {                             // the next tile of this wave is in flight during the MFMAs
            const size_t koff = (size_t)(kt + KS_NW) * ROW_BYTES;
#pragma unroll
            for (int i = 0; i < 4; ++i) { ra4[i] = *reinterpret_cast<const u32x4*>(a_src[i] + koff); rw4[i] = *reinterpret_cast<const u32x4*>(w_src[i] + koff); }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const u32x4 fa = *reinterpret_cast<const u32x4*>(sA + frag0 + c * 32);
            const u32x4 fb = *reinterpret_cast<const u32x4*>(sW + frag0 + c * 32);
            mfma_chunk<T>(fb, fa, acc);                    // D[n][m]: each lane ends up with 4 consecutive n of one row m
        }
    }
    float* red = reinterpret_cast<float*>(smem + KS_NW * KS_WAVE_LDS);
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((wave - 1) * 16 + r) * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int w = 0; w < KS_NW - 1; ++w)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += red[(w * 16 + r) * 64 + lane];
    // ---- epilogue (the D[n][m] form of gemm_nt_kernel): bias -> activation -> (+residual) [-> activation] -> store
    T* Ct = reinterpret_cast<T*>(p.Ct);
    const int row = m0 + (lane & 31);
    if (row >= p.M) return;
    const int rr = p.res_mod > 0 ? (row % p.res_mod) : row;
    const bool vec_ok = (p.N % 4 == 0) && (!p.R || p.ldr % 4 == 0) && (!p.Cf || p.ldcf % 4 == 0) && (!Ct || p.ldct % 4 == 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int col = n0 + 8 * q + 4 * (lane >> 5);
        if (col >= p.N) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[4 * q + e];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = col + e;
            if (c >= p.N) { v[e] = 0.f; continue; }
            float x = v[e] + (p.bias ? p.bias[c] : 0.0f);
            if (!p.act_after_res) x = apply_act(x, p.act);
            if (p.R) x += p.R[(size_t)rr * p.ldr + c];
            if (p.act_after_res) x = apply_act(x, p.act);
            v[e] = x;
        }
        if (vec_ok) {
            if (p.Cf) { f32x4 o4; o4.x = v[0]; o4.y = v[1]; o4.z = v[2]; o4.w = v[3]; *reinterpret_cast<f32x4*>(p.Cf + (size_t)row * p.ldcf + col) = o4; }
            if (Ct) {
                T o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = from_f32<T>(v[e]);
                if (sizeof(T) == 2) *reinterpret_cast<uint2*>(Ct + (size_t)row * p.ldct + col) = *reinterpret_cast<uint2*>(o);
                else *reinterpret_cast<f32x4*>(Ct + (size_t)row * p.ldct + col) = *reinterpret_cast<f32x4*>(o);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = col + e;
                if (c >= p.N) continue;
                if (p.Cf) p.Cf[(size_t)row * p.ldcf + c] = v[e];
                if (Ct) Ct[(size_t)row * p.ldct + c] = from_f32<T>(v[e]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Skinny product for M <= 16 rows (time / speaker / FiLM embeddings at chain batch sizes: models/transformer.py:77, :446-457).
// Such a launch is pure weight streaming — the stacked FiLM Linear alone is 67 MB of bf16 for two rows — and the 128 x 128
// MFMA tile above does it with 128 blocks that each walk 512 KB serially (30 us per launch at B = 1).  Here one wave owns four
// output features: the 64 lanes split K in 16-byte pieces, keep up to eight such pieces per weight row in flight, multiply
// against the activation rows staged in LDS (fp32), and finish with a wavefront reduction.  fp32 accumulation over the
// same operands, different summation order than the tile kernel (both within fp32 round-off of the exact sum).
constexpr int GV_ROWS = 4;                       // output features per wave
constexpr int GV_WAVES = 4;                      // waves per block -> 16 features per block
constexpr int GV_MMAX = 16;

template <typename T, int MT>
__global__ __launch_bounds__(64 * GV_WAVES) void gemv_rows_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int EPL = 16 / (int)sizeof(T);     // elements per lane and k step (8 bf16 / 4 fp32)
    float* sA = reinterpret_cast<float*>(smem);  // [MT][K] fp32
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const T* A = reinterpret_cast<const T*>(p.A);
    for (int i = tid; i < MT * p.K; i += 64 * GV_WAVES) {
        const int m = i / p.K, k = i - m * p.K;
        sA[i] = m < p.M ? to_f32<T>(A[(size_t)m * p.lda + k]) : 0.f;
    }
    __syncthreads();
    const int n0 = (blockIdx.x * GV_WAVES + wave) * GV_ROWS;
    const T* W = reinterpret_cast<const T*>(p.W);
    float acc[MT][GV_ROWS];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < GV_ROWS; ++r) acc[m][r] = 0.f;
    const int nsteps = (p.K + 64 * EPL - 1) / (64 * EPL);
    for (int st0 = 0; st0 < nsteps; st0 += 2) {
        u32x4 w[2][GV_ROWS];
        int kk[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            kk[u] = ((st0 + u) * 64 + lane) * EPL;
#pragma unroll
            for (int r = 0; r < GV_ROWS; ++r) {
                const int n = n0 + r < p.N ? n0 + r : p.N - 1;
                const bool ok = kk[u] < p.K;
                w[u][r] = ok ? *reinterpret_cast<const u32x4*>(W + (size_t)n * p.ldw + kk[u]) : u32x4{0, 0, 0, 0};
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (kk[u] >= p.K) continue;
            float wf[GV_ROWS][EPL];
#pragma unroll
            for (int r = 0; r < GV_ROWS; ++r) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t bits = w[u][r][j];       // scalar copy: __builtin_bit_cast on a vector ELEMENT reads element 0 (hipcc 7.2)
                    if (sizeof(T) == 2) {
                        wf[r][(2 * j) % EPL] = __builtin_bit_cast(float, bits << 16);
                        wf[r][(2 * j + 1) % EPL] = __builtin_bit_cast(float, bits & 0xffff0000u);
                    } else {
                        wf[r][j % EPL] = __builtin_bit_cast(float, bits);
                    }
                }
            }
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const float* ar = sA + m * p.K + kk[u];
                float av[EPL];
#pragma unroll
                for (int j = 0; j < EPL; j += 4) {
                    const f32x4 a4 = *reinterpret_cast<const f32x4*>(ar + j);
                    av[j] = a4.x; av[j + 1] = a4.y; av[j + 2] = a4.z; av[j + 3] = a4.w;
                }
#pragma unroll
                for (int r = 0; r < GV_ROWS; ++r)
#pragma unroll
                    for (int j = 0; j < EPL; ++j) acc[m][r] = fmaf(av[j], wf[r][j], acc[m][r]);
            }
        }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < GV_ROWS; ++r) {
            float v = acc[m][r];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            acc[m][r] = v;
        }
    // lane (m * GV_ROWS + r) writes output (m, n0 + r)
    T* Ct = reinterpret_cast<T*>(p.Ct);
    float mine = 0.f;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < GV_ROWS; ++r)
            if (lane == m * GV_ROWS + r) mine = acc[m][r];
    const int m = lane / GV_ROWS, n = n0 + (lane % GV_ROWS);
    if (lane < MT * GV_ROWS && m < p.M && n < p.N) {
        float x = mine + (p.bias ? p.bias[n] : 0.0f);
        if (!p.act_after_res) x = apply_act(x, p.act);
        const int rr = p.res_mod > 0 ? (m % p.res_mod) : m;
        if (p.R) x += p.R[(size_t)rr * p.ldr + n];
        if (p.act_after_res) x = apply_act(x, p.act);
        if (p.Cf) p.Cf[(size_t)m * p.ldcf + n] = x;
        if (Ct) Ct[(size_t)m * p.ldct + n] = from_f32<T>(x);
    }
}

template <typename T>
static int launch_gemv_t(const GemmArgs& a, hipStream_t s) {
    const int blocks = ceil_div(a.N, GV_ROWS * GV_WAVES);
    const int mt = a.M <= 4 ? 4 : 16;
    const size_t lds = (size_t)mt * a.K * sizeof(float);
    if (mt == 4) hipLaunchKernelGGL((gemv_rows_kernel<T, 4>), dim3(blocks), dim3(64 * GV_WAVES), lds, s, a);
    else hipLaunchKernelGGL((gemv_rows_kernel<T, 16>), dim3(blocks), dim3(64 * GV_WAVES), lds, s, a);
    DSH_HIP_CHECK(hipGetLastError());
    return 0;
}

template <typename T>
static int launch_gemm_t(const GemmArgs& a, hipStream_t s) {
    DSH_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "gemm dims must be positive");
    DSH_REQUIRE(a.K % gemm_k_align<T>() == 0, "gemm K must be padded to the 128-byte K tile");
    DSH_REQUIRE(a.lda >= a.K && a.ldw >= a.K, "gemm leading dims smaller than K");
    DSH_REQUIRE((a.lda * sizeof(T)) % 16 == 0 && (a.ldw * sizeof(T)) % 16 == 0, "gemm leading dims must be 16-byte multiples");
    DSH_REQUIRE(((uintptr_t)a.A % 16) == 0 && ((uintptr_t)a.W % 16) == 0, "gemm operands must be 16-byte aligned");
    // M <= 16: weight streaming (activation rows staged in LDS as fp32: up to 16 x 2048 x 4 bytes)
    static int gemv_on = -1;
    if (gemv_on < 0) { const char* e = getenv("DSH_GEMV"); gemv_on = e ? atoi(e) : 1; }
    if (gemv_on && a.M <= GV_MMAX && (size_t)(a.M <= 4 ? 4 : 16) * a.K * sizeof(float) <= 128 * 1024) {
        static bool gv_attr = false;
        if (!gv_attr) {
            DSH_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemv_rows_kernel<T, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
            DSH_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemv_rows_kernel<T, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
            gv_attr = true;
        }
        return launch_gemv_t<T>(a, s);
    }
    // fp32, a few hundred rows at most: K split over the waves of a 32 x 32-tile block (gemm_nt_ksplit_kernel); DSH_GEMM_KSPLIT=n
    // sets the row limit (default 512; 0: off)
    if constexpr (sizeof(T) == 4) {
        static const int ks_rows = [] { const char* e = getenv("DSH_GEMM_KSPLIT"); return e ? atoi(e) : 512; }();
        if (a.M <= ks_rows) {
            static bool ks_attr = false;
            if (!ks_attr) {
                DSH_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_ksplit_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, KS_LDS));
                ks_attr = true;
            }
            hipLaunchKernelGGL((gemm_nt_ksplit_kernel<T>), dim3(ceil_div(a.N, 32), ceil_div(a.M, 32)), dim3(64 * KS_NW), KS_LDS, s, a);
            DSH_HIP_CHECK(hipGetLastError());
            return 0;
        }
    }
    static int variant = -1, tile_sel = 1;
    if (variant < 0) {
        const char* e = getenv("DSH_GEMM_VARIANT");
        variant = e ? atoi(e) : 1;
        const char* ts = getenv("DSH_GEMM_TILE");        // 0: always 128 x 128; 2 / 3 / 4: always 128 x 64 / 64 x 64 / 64 x 32 (measurement)
        tile_sel = ts ? atoi(ts) : 1;
        DSH_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_kernel<T, 0, 2, 2>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, gemm_lds_bytes(2, 2)));
        DSH_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_kernel<T, 1, 2, 2>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, gemm_lds_bytes(2, 2)));
        DSH_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_kernel<T, 1, 2, 1>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, gemm_lds_bytes(2, 1)));
        DSH_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_kernel<T, 1, 1, 1>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, gemm_lds_bytes(1, 1)));
        DSH_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_kernel<T, 1, 1, 1, 1>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, gemm_lds_bytes(1, 1, 1)));
    }
    // Tile shape: 64 x 64 unless overridden.  Measured: on the fp32 parity configuration (M = 8704) the smaller tile removes the
    // quantisation of 272 128 x 128 tiles on 256 CUs (16.8 k -> 22.8 k frames/s); on the bf16 path's skinny / deep-K launches (FiLM
    // table GEMM M = 475, K = 2048; K = 128 .. 256 projections) four co-resident blocks per CU hide the per-K-step latency that two
    // 128 x 128 blocks expose (35.1 -> 29.8 ms per bench step); at chain batches it simply yields more blocks.  A cost model
    // (tiles per CU x area) picked 128 x 128 for the bf16 launches and was slower.  DSH_GEMM_TILE = 0 / 2 / 3 / 4 force a shape.
    struct Shape { int mi, nj, wn, bm, bn, occ; double area, ovh; };
    static const Shape shapes[4] = {{2, 2, 2, 128, 128, 2, 4.0, 1.00}, {2, 1, 2, 128, 64, 2, 2.0, 1.04},
                                    {1, 1, 2, 64, 64, 4, 1.0, 1.10}, {1, 1, 1, 64, 32, 5, 0.5, 1.18}};
    int pick = 0;
    if (variant != 0 && tile_sel) {
        pick = 2;
        // the stacked FiLM Linear of an encoder (a few hundred rows x 16 384 features, K = 2048: 67 MB of weights per launch) is the one shape
        // that prefers 128 x 128 tiles — a quarter of the weight-panel re-reads: 158 -> 111 us per launch at 950 rows (round 4 measurement;
        // round 5: every launch costs the three-stream step 80 - 97 % of its own time, so the 2 ms per step are worth taking)
        if (sizeof(T) == 2 && tile_sel == 1 && a.M <= 2048 && a.N >= 8192 && a.K >= 1024) pick = 0;
        if (tile_sel >= 2 && tile_sel <= 4) pick = tile_sel - 1;
    }
    const Shape& sh = shapes[pick];
    GemmArgs b = a;
    b.nt_n = ceil_div(a.N, sh.bn);
    b.nt_m = ceil_div(a.M, sh.bm);
    if (variant == 0) {
        hipLaunchKernelGGL((gemm_nt_kernel<T, 0, 2, 2>), dim3(b.nt_n, b.nt_m), dim3(256), gemm_lds_bytes(2, 2), s, b);
    } else {
        const int groups = ceil_div(b.nt_m, 8);
        const dim3 grid(b.nt_m >= 8 ? groups * 8 * b.nt_n : b.nt_m * b.nt_n);
        if (pick == 0) hipLaunchKernelGGL((gemm_nt_kernel<T, 1, 2, 2>), grid, dim3(256), gemm_lds_bytes(2, 2), s, b);
        else if (pick == 1) hipLaunchKernelGGL((gemm_nt_kernel<T, 1, 2, 1>), grid, dim3(256), gemm_lds_bytes(2, 1), s, b);
        else if (pick == 2) hipLaunchKernelGGL((gemm_nt_kernel<T, 1, 1, 1>), grid, dim3(256), gemm_lds_bytes(1, 1), s, b);
        else hipLaunchKernelGGL((gemm_nt_kernel<T, 1, 1, 1, 1>), grid, dim3(128), gemm_lds_bytes(1, 1, 1), s, b);
    }
    DSH_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_gemm_f32(const GemmArgs& a, hipStream_t s) { return launch_gemm_t<float>(a, s); }
int launch_gemm_bf16(const GemmArgs& a, hipStream_t s) { return launch_gemm_t<bf16>(a, s); }

}  // namespace dsh
