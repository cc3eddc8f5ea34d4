// Token-per-lane kernels, second generation: the weight operand reaches LDS by LDS-DMA (global_load_lds_dwordx4) from a
// FRAGMENT-ORDERED copy of the weights, and the FFN branch of a decoder layer is one kernel.
//
// Why (round-1 profile of tl_linear.hip): the MFMA-bound instantiations ran at 24-33 % of the matrix peak.
// With the activation operand stationary in registers, the only thing the main loop has to do besides MFMAs is to move W,
// and the first generation moved it global -> VGPR -> ds_write_b128 -> LDS: 16 global loads, 16 wide LDS stores (13 cycles
// of issue each) and two vmcnt waits per wave and 32-feature tile, plus 64 staging VGPRs.  Here:
//
//   * W is stored in HBM exactly as the LDS image the MFMA A-operand reads want ("fragment order"): for the 32-row tile nt and
//     the 16-wide k step s, one contiguous KB = 64 lanes x 16 B, lane L = (n = L & 31, h = L >> 5) holding
//     W'[32 nt + n][16 s + 8 h .. + 7]  (W' = rows pi-permuted inside the tile, tl_weight_src_row).  A tile is a contiguous
//     K * 64 bytes.  global_load_lds writes LDS linearly (M0 base + lane * 16), so the copy is a plain linear stream and the
//     A-fragment read is ds_read_b128 at base + s * 1024 + lane * 16: conflict free by construction, no padding.
//   * per tile and wave: K / 64 DMA instructions, issued in one burst right after the tile barrier together with the
//     residual loads of this tile and the (deferred) stores of the previous one; one counted wait (vmcnt(0)) at the END of
//     the tile, when everything issued at its start has long landed.  Between them the wave issues only MFMAs and
//     ds_read_b128.  No staging registers: K = 512 kernels fit 2 blocks / CU with room to spare.
//   * tl2_ffn_kernel: ffn.linear1 -> GELU -> ffn.linear2 -> StylizationBlock(LN, FiLM, SiLU, Linear) -> + h for a wave's
//     32 tokens without leaving the register file: the 1024-wide hidden is produced 32 features at a time and immediately
//     consumed as one K chunk of linear2, whose 16 accumulators (the wave's 32 x 512 output) stay resident; LayerNorm
//     statistics are taken from the fp32 accumulators.  HBM traffic per token drops from 13 KB (three launches) to 6 KB
//     (h16 in, fp32 h in/out, h16 out); models/transformer.py:169-181, :86-97.
//
// Layouts of the activation tensors (tiled bf16, lane-native fp32) are those of tl_linear.hip.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "dsh_common.h"
#include "dsh_kernels.h"
#include "tl_common.h"

namespace dsh {

namespace {

// The FLAT-GLOBAL form (64-bit per-lane address), still used by tl2_linear_kernel: there — two waves per SIMD (K = 512), or a
// kernel whose prologue loads keep hipcc's vmcnt bookkeeping busy (K = 1024 concat) — the buffer form measured no better
// (q|k|v 288 vs 288 us) or worse (feat_proj.1 219 -> 266 us), round 3.
__device__ __forceinline__ void dma_sel(int k, const char* src_lane, char* lds_wave) {
    const char* s4 = src_lane + (k >> 2) * 4096;
    char* d4 = lds_wave + (k >> 2) * 4096;
    switch (k & 3) {
        case 0: __builtin_amdgcn_global_load_lds((gptr_t)s4, (lptr_t)d4, 16, 0, 0); break;
        case 1: __builtin_amdgcn_global_load_lds((gptr_t)s4, (lptr_t)d4, 16, 1024, 0); break;
        case 2: __builtin_amdgcn_global_load_lds((gptr_t)s4, (lptr_t)d4, 16, 2048, 0); break;
        default: __builtin_amdgcn_global_load_lds((gptr_t)s4, (lptr_t)d4, 16, 3072, 0); break;
    }
}
template <int N>
__device__ __forceinline__ void dma_kbs(const char* src_lane, char* lds_wave) {
#pragma unroll
    for (int i = 0; i < N; ++i) dma_sel(i, src_lane, lds_wave);
}

// In-loop phase probe (bench only, PROBE instantiations): shader-clock stamps at four points of every main-loop iteration,
// accumulated per phase by wave 0 of a few blocks.  The stamps of iteration i are only READ at the top of iteration i + 1,
// right after the barrier, where no LDS read is outstanding: the s_waitcnt lgkmcnt(0) the read implies costs nothing there.
struct PhaseProbe {
    unsigned long long t[4] = {0, 0, 0, 0}, prev[4] = {0, 0, 0, 0}, acc[4] = {0, 0, 0, 0};
    int n = 0;
    __device__ __forceinline__ void stamp(int i) { t[i] = __builtin_readcyclecounter(); }
    // call after stamp(0) of the new iteration: spans of the previous one = [barrier passed -> setup done, -> MFMA groups (with
    // their DMA / loads / stores) issued, -> epilogue done, -> next counted wait + barrier passed]
    __device__ __forceinline__ void fold() {
        if (n > 0) {
            acc[0] += prev[1] - prev[0]; acc[1] += prev[2] - prev[1]; acc[2] += prev[3] - prev[2]; acc[3] += t[0] - prev[3];
        }
        ++n;
    }
    __device__ __forceinline__ void roll() { prev[0] = t[0]; prev[1] = t[1]; prev[2] = t[2]; prev[3] = t[3]; }
    __device__ __forceinline__ void dump(unsigned long long* out, unsigned long long c0, unsigned long long w0) {
        if (out && (threadIdx.x & 63) == 0 && threadIdx.x < 64) {
            unsigned long long* r = out + (size_t)blockIdx.x * 8;
            r[0] = acc[0]; r[1] = acc[1]; r[2] = acc[2]; r[3] = acc[3]; r[4] = (unsigned long long)n;
            r[5] = __builtin_readcyclecounter() - c0; r[6] = wall_clock64() - w0; r[7] = 0;
        }
    }
};

// NF MFMAs of `acc` against fragments fr[0 .. NF): A fragments at lds + i KB, read 4 ahead (one group = 4 MFMAs = 128 cycles,
// about one ds_read_b128 latency); the issue order is pinned, hipcc otherwise re-serialises each read in front of its MFMA
template <int NF, int GS = 4>
__device__ __forceinline__ void mfma_run(f32x16& acc, const char* lds_lane, const u32x4* fr) {
    static_assert(NF % GS == 0, "whole fragment groups");
    u32x4 aw[2][GS];
#pragma unroll
    for (int i = 0; i < GS; ++i) aw[0][i] = *reinterpret_cast<const u32x4*>(lds_lane + i * 1024);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < NF / GS; ++g) {
        if (g + 1 < NF / GS) {
#pragma unroll
            for (int i = 0; i < GS; ++i) aw[(g + 1) & 1][i] = *reinterpret_cast<const u32x4*>(lds_lane + ((g + 1) * GS + i) * 1024);
        }
#pragma unroll
        for (int i = 0; i < GS; ++i)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aw[g & 1][i]), __builtin_bit_cast(bf16x8, fr[g * GS + i]), acc, 0, 0, 0);
    }
#pragma unroll
    for (int g = 0; g < NF / GS; ++g) {
        if (g + 1 < NF / GS) __builtin_amdgcn_sched_group_barrier(0x100, GS, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, GS, 0);
    }
}

}  // namespace

// =====================================================================================================================
// Fused Linear: out = epilogue(prologue(X) W^T); TlArgs as tl_linear_kernel, W in fragment order.
//
// Geometry.  K = 512: 8 waves / 256 tokens per block, one block per CU (two waves per SIMD, 128 fragment registers each).
// K = 1024: 4 waves / 128 tokens (256 fragment registers: one wave per SIMD).  Measured on MI355X (round 2, in-loop phase
// probe): with 128-token blocks that issued their weight DMA, residual loads and stores as one burst after the tile barrier,
// a wave spent 1400 - 2800 cycles per 32-feature tile just ISSUING those vector-memory instructions (the CU's address path
// moves 64 B/clk and every wave of the CU was in the same burst), against 1200 cycles of MFMAs.  Hence:
//   * W reuse per CU is doubled where registers allow it (256 tokens share one W stream);
//   * the weight stream runs through a ring of four 32 KB chunks (32 fragments: a whole K = 512 tile or half a K = 1024 one),
//     the DMA of chunk p + 3 is issued during phase p (32 MFMAs per wave on chunk p), one instruction per MFMA group, and
//     the top-of-phase wait is COUNTED (only chunk p must have landed);
//   * the stores of a finished tile are issued one tile later, also one per MFMA group (ordinary stores: inline-asm stores
//     produced sporadic garbage).  The counted waits stay valid whatever order stores complete in: "at most Y
//     operations outstanding", Y = the number of LOADS younger than the one waited for, implies that load has returned (if it
//     had not, those Y younger loads would be outstanding too).
//   * PRO 1 / 3 (LayerNorm before the Linear) do not normalise the operand any more: the affine is folded into the weights,
//     W'[n][k] = gamma[k] W[n][k], and   LN(x) W^T + b = rstd (x W'^T - mean c) + d,   c[n] = sum_k W'[n][k],
//     d[n] = b[n] + sum_k beta[k] W[n][k]  (a.row_const = c, a.bias = d; built by finalize()).  The prologue only takes the
//     row statistics of the raw bf16 fragments; the bf16 rounding of the normalised operand disappears.
// LDS: [4][32 KB] chunk ring | bias / d [N] | c [N]; the StylizationBlock prologue's folded FiLM rows overlay the two ring
// slots that are not in flight before the loop starts.
constexpr int T2_CHUNK = 32 * 1024;
constexpr int T2_MAXCLIP = 10;                     // clips a block may span in the FiLM prologue (256 tokens: clips of >= 29 frames)

// (amdgpu_waves_per_eu pins the occupancy the register allocator aims at to the one block per CU the 128 KB LDS ring allows: without
//  the upper bound hipcc squeezed two rolling K = 1024 instantiations into 126 VGPRs "for" four waves per SIMD and spilled 1 KB per
//  lane into scratch)
// NZ (rolling K = 1024 concat form only): trailing all-zero fragments of the row — the expression encoder's concat is 896 wide (8 zero
// fragments), the gesture encoder's 999 (1008 with the padded expression segment: 1) — whose MFMAs and fragment reads are skipped
template <int KD, int PRO, bool HAS_R, int OUT, int ACT, bool PROBE = false, bool ROLL = false, bool HL = false, int NZ = 0>
__global__ __attribute__((amdgpu_flat_work_group_size((KD == 512 ? 512 : 256), (KD == 512 ? 512 : 256)), amdgpu_waves_per_eu((KD == 512 ? 2 : 1), (KD == 512 ? 2 : 1))))
void tl2_linear_kernel(TlArgs p) {
    constexpr int NW = KD == 512 ? 8 : 4;            // waves per block
    constexpr int NTHR = NW * 64, TOK = NW * 32;
    constexpr int PH = KD / 512;                     // phases (32-fragment chunks) per 32-feature tile
    constexpr int ND = 32 / NW;                      // DMA instructions (1 KB each) per wave and chunk
    constexpr int NR = HAS_R ? 4 : 0;                // residual loads per tile
    constexpr int NSTORE = ((OUT & 5) ? 4 : 0) + ((OUT & 2) ? 2 : 0);
    constexpr bool FOLD = PRO == 1 || PRO == 3;      // LayerNorm folded into W: epilogue applies rstd / mean
    constexpr bool HAS_C = (PRO == 2 && HAS_R && ACT == ACT_NONE);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    PhaseProbe pp;
    const unsigned long long pc0 = PROBE ? __builtin_readcyclecounter() : 0, pw0 = PROBE ? wall_clock64() : 0;
    trace_mark(p.trace, 0);
    start_stagger(p.stag_groups, p.stag_sleep);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ml = lane & 31, h = lane >> 5;
    const int bx = tl_block_index(p.rev);
    const int tb = bx * NW + wave;                               // 32-token block owned by this wave (rows are not bounds-checked)
    const int row = tb * 32 + ml;
    const int lane_off = ml * 32 + h * 16;
    const int NT = p.N / 32;
    const int nt0 = blockIdx.y * p.tiles_per_block;
    const int nt1 = (nt0 + p.tiles_per_block) < NT ? (nt0 + p.tiles_per_block) : NT;
    const int p0 = nt0 * PH, p1 = nt1 * PH;                     // chunk range of this block
    // this wave's share of a chunk: DMA source (per lane) and LDS destination (wave uniform)
    const char* wsrc = reinterpret_cast<const char*>(p.W) + wave * (ND * 1024) + lane * 16;
    char* wdst = smem + wave * (ND * 1024);
    const bool whot = (p.dbg & 8) != 0;                          // bench ablation: the stream re-reads its first chunk (L2-hot)
    auto dma_src = [&](int q) -> const char* { const int c = whot ? p0 : (q < p1 ? q : p1 - 1); return wsrc + (size_t)c * T2_CHUNK; };
    auto dma_dst = [&](int q) -> char* { return wdst + (q & 3) * T2_CHUNK; };
    // ROLL: the MUBUF form of the LDS-DMA (dma_buf, tl_common.h) — hipcc keeps exact lgkmcnt counts next to it, which the rolling
    // main loop below depends on (next to the FLAT form it waits lgkmcnt(0) in front of every MFMA: the read 4 slots ahead would
    // be exposed in every slot)
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.W), 0, p.N * KD * 2, 0x00020000);
    const int wvoff = wave * (ND * 1024) + lane * 16;
    // round 6: every block of a launch streams the SAME weight matrix, and the blocks of a round run in lockstep — 32 CUs of an XCD ask
    // their L2 for the same lines at the same moment.  With p.rot the rolling loop walks the tiles in a rotated order, block b starting
    // at tile (b / 8) % tiles (b / 8 = the block's index inside its XCD): the stream is the same, the CUs are spread over it.  Tiles are
    // independent, so results do not change.  `nt` below stays the LOGICAL tile (ring slots, loop control), ptile(nt) the PHYSICAL one
    // (weight source, bias / c / d, residual and output addresses).
    const int ntl = nt1 - nt0;
    const int rot = (ROLL && p.rot) ? (int)((unsigned)(bx >> 3) % (unsigned)ntl) : 0;
    auto ptile = [&](int nt) -> int { const int t = nt + rot; return t >= nt1 ? t - ntl : t; };
    auto dma_soff = [&](int q) -> int {
        const int qc = q < p1 ? q : p1 - 1, lt = PH == 1 ? qc : (qc >> 1), kk = PH == 1 ? 0 : (qc & 1);
        return (ptile(lt) * PH + kk) * T2_CHUNK;
    };
    auto issue_chunk = [&](int q) {
        if (ROLL) {
#pragma unroll
            for (int k = 0; k < ND; ++k) dma_buf(k, wrsrc, wvoff, dma_soff(q), dma_dst(q));
        } else dma_kbs<ND>(dma_src(q), dma_dst(q));
    };
    issue_chunk(p0);
    issue_chunk(p0 + 1);

    // ---- prologue parameters: folded FiLM rows (A | B) of this block's clips (PRO 2 only) --------------------------------
    constexpr int NPRM = PRO == 2 ? T2_MAXCLIP * 1024 / (NTHR * 4) : 1;
    f32x4 prm[NPRM];
    int clip0 = 0;
    if (PRO == 2) {
        const int rb = bx * TOK, rrb = rb >= p.half_row0 ? rb - p.half_row0 : rb;
        clip0 = rrb / p.frames;
        const int nclip = (rrb + TOK - 1) / p.frames - clip0 + 1;
#pragma unroll
        for (int c = 0; c < NPRM; ++c) {                        // float index f of the staged table: clip f / 1024, offset f % 1024
            const int f = (c * NTHR + tid) * 4;
            const int ci = f >> 10, cc = ci < nclip ? ci : nclip - 1;
            prm[c] = *reinterpret_cast<const f32x4*>(p.film + (size_t)((clip0 + cc) % p.bmod) * p.film_ld + p.film_off + (f & 1023));
        }
    }
    // ---- activation rows -> B fragments ---------------------------------------------------------------------------------
    constexpr int NFRAG = KD / 16;
    u32x4 frag[NFRAG];
    if (PRO == 3) {
        const char* r0 = reinterpret_cast<const char*>(p.X) + (size_t)tb * (512 / 16) * 1024 + lane_off;
        const char* r1 = reinterpret_cast<const char*>(p.X1) + (size_t)tb * (256 / 16) * 1024 + lane_off;
        const char* r2 = reinterpret_cast<const char*>(p.X2) + (size_t)tb * (128 / 16) * 1024 + lane_off;
        const bool has3 = p.X3 != nullptr;
        const char* r3 = has3 ? reinterpret_cast<const char*>(p.X3) + (size_t)tb * (128 / 16) * 1024 + lane_off : r2;
#pragma unroll
        for (int s = 0; s < NFRAG; ++s) {
            const char* src = s < 32 ? r0 + s * 1024 : (s < 48 ? r1 + (s - 32) * 1024 : (s < 56 ? r2 + (s - 48) * 1024 : r3 + (s - 56) * 1024));
            u32x4 v = *reinterpret_cast<const u32x4*>(src);
            if (s >= 56 && !has3) { v[0] = 0; v[1] = 0; v[2] = 0; v[3] = 0; }
            frag[s] = v;
        }
    } else {
        const char* xr = reinterpret_cast<const char*>(p.X) + (size_t)tb * (p.ldx / 16) * 1024 + lane_off;
#pragma unroll
        for (int s = 0; s < NFRAG; ++s) frag[s] = *reinterpret_cast<const u32x4*>(xr + s * 1024);
    }
    float* sbias = reinterpret_cast<float*>(smem + 4 * T2_CHUNK);
    float* sconst = sbias + p.N;
    for (int i = tid; i < p.N; i += NTHR) {
        sbias[i] = p.bias ? p.bias[i] : 0.f;
        sconst[i] = p.row_const ? p.row_const[i] : 0.f;
    }
    float* sprm = reinterpret_cast<float*>(smem + ((p0 + 2) & 3) * T2_CHUNK);    // two ring slots not yet in flight (contiguous mod 4 ...)
    // (slots (p0 + 2) & 3 and (p0 + 3) & 3 are adjacent unless (p0 + 2) & 3 == 3: the table is staged clip by clip, each clip
    //  inside one slot, so adjacency is not needed: clip c lives in slot (p0 + 2 + (c >> 3)) & 3 at (c & 7) * 4 KB)
    auto clip_ptr = [&](int c) -> float* { return reinterpret_cast<float*>(smem + ((p0 + 2 + (c >> 3)) & 3) * T2_CHUNK) + (c & 7) * 1024; };
    (void)sprm;
    if (PRO == 2) {
#pragma unroll
        for (int c = 0; c < NPRM; ++c) {
            const int f = (c * NTHR + tid) * 4;
            *reinterpret_cast<f32x4*>(clip_ptr(f >> 10) + (f & 1023)) = prm[c];
        }
    }
    __syncthreads();                                  // tables visible to the whole block
    float rstd = 1.f, nmr = 0.f;                      // FOLD: per-row LayerNorm statistics, applied in the epilogue
    if (PRO == 2) {
        const int rr = row >= p.half_row0 ? row - p.half_row0 : row;
        int ci = rr / p.frames - clip0;               // rows past the last clip (block padding) may exceed the staged rows
        ci = ci < T2_MAXCLIP ? ci : T2_MAXCLIP - 1;
        const float* ca = clip_ptr(ci) + 8 * h;
        ln_frags<NFRAG, true>(frag, ca, ca + 512, (float)KD, (float)KD);
    } else if (FOLD) {
        // raw moments of the bf16 row by packed dot products (row_moments_bf16, tl_common.h)
        const float kn = PRO == 3 ? (float)p.kreal : (float)KD;
        float sum, sq;
        row_moments_bf16<NFRAG>(frag, sum, sq);
        const float mean = sum / kn;
        sq = fmaxf(sq - sum * mean, 0.f);                           // sum (x - mean)^2
        rstd = 1.0f / sqrtf(sq / kn + 1e-5f);
        nmr = -mean * rstd;
    }
    // K = 1024: 64 fragments = 256 registers do not fit the 256 architectural VGPRs next to everything else; hipcc then SPILLS
    // fragments to AGPRs and reloads each with four v_accvgpr_read in front of its MFMA (two extra instructions per MFMA of the
    // loop, round-3 disassembly).  An MFMA reads its B operand from an AGPR just as well: the upper half of the row is moved
    // there once, as values of the accumulator register class.  (128 instructions fewer per 64-MFMA tile; the launch time of
    // feat_proj.1 did not move: 222 us either way.)
#pragma unroll
    for (int s = 0; s < NFRAG; ++s) {
        if (KD == 1024 && s >= 32) asm volatile("" : "+a"(frag[s]));
        else asm volatile("" ::"v"(frag[s]));
    }
    // the residual of the first tile; then everything requested so far has landed (rows, first two chunks), and the third
    // chunk goes in flight: from here on the queue follows the steady-state pattern the counted waits assume
    f32x4 rres[4];
    const size_t fstride = (size_t)4 * 64 * 4;                   // floats per (token block, tile) of the lane-native fp32 layout
    const size_t fbase = ((size_t)tb * NT * 4 * 64 + lane) * 4;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                  // FiLM rows consumed by every wave: their ring slots may be overwritten
    issue_chunk(p0 + 2);
    trace_mark(p.trace, 1);

    // ---- main loop -------------------------------------------------------------------------------------------------------
    char* Ctb = reinterpret_cast<char*>(p.Ct);
    const float const_on = (p.row_const != nullptr && row < p.n_const_rows) ? 1.0f : 0.0f;
    if constexpr (ROLL) {
        // Rolling main loop (round 5; the structure of tl3_ffn_kernel's pipelined phase C): the loop below ends every tile with
        // "last MFMA -> accumulator read -> epilogue -> counted wait -> barrier -> bias -> first four fragment reads -> LDS latency ->
        // first MFMA", and both waves of a SIMD reach that sequence together (round-3 phase probe: 1892 cycles of MFMA groups, 713
        // of epilogue and 533 at the wait + barrier per q|k|v tile).  Here
        //   * a phase is 32 explicit issue slots (one MFMA, the fragment read 4 slots ahead, the phase's DMA pieces, one small piece
        //     of the previous tile's epilogue) fenced by sched_barrier(0);
        //   * the A fragments roll across the phase boundary (slots 28..31 read the next chunk's first four), so the counted wait +
        //     barrier that publish chunk q + 1 sit behind slot 27 of phase q, in front of them an lgkmcnt(0) that retires this wave's
        //     reads of chunk q (all issued by slot 23), whose ring slot the next phase's DMA refills;
        //   * the tile alternates between two accumulators: the epilogue of tile t - 1 (folded LayerNorm, activation, bf16 pack, two
        //     16-byte stores, both issued before slot 27) reads the finished accumulator in place during tile t.
        // Every phase is identical (the last one reads ahead into a chunk that is never used and passes one barrier more).
        // HL (round 5): the two residual-carrying launches of a layer on this loop as well — the residual arrives as hi / lo bf16 planes
        // (p.R reinterpreted = hi, p.Rlo), the result leaves as p.Ct (hi) + p.Clo (lo).  The four residual fragments of tile t are
        // requested in the first slots of tile t's own phase, one whole tile before its epilogue reads them; outputs leave through
        // asm stores (asm_store16, tl_common.h) so that hipcc's counted waits for those loads stay counted.
        static_assert(!PROBE && ((!HAS_R && OUT == 2 && !HL) || (HAS_R && OUT == 3 && HL && ACT == ACT_NONE)), "rolling main loop: bf16-out Linears, or the hi / lo residual form");
        struct Res { u32x4 hi[2], lo[2]; };
        Res resA, resB;                                   // residual fragments of the tile in accA / accB
        const char* Rhi = reinterpret_cast<const char*>(p.R);
        const char* Rlo = reinterpret_cast<const char*>(p.Rlo);
        auto load_res = [&](Res& r, int nt) {
            const size_t pidx = ((size_t)tb * (2 * NT) + 2 * nt) * 1024 + lane_off;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                r.hi[c] = *reinterpret_cast<const u32x4*>(Rhi + pidx + c * 1024);
                r.lo[c] = *reinterpret_cast<const u32x4*>(Rlo + pidx + c * 1024);
            }
        };
        typedef __attribute__((address_space(3))) const char* lcptr_t;
        typedef __attribute__((address_space(3))) const u32x4* lfrag_t;
        const char* lds_lane_r = smem + lane * 16;
        auto chunk_base = [&](int q) -> lcptr_t { lcptr_t b = (lcptr_t)lds_lane_r + (q & 3) * T2_CHUNK; asm volatile("" : "+v"(b)); return b; };
        u32x4 aw[2][4];
        f32x16 accA, accB;
        // loads younger than chunk q + 1's DMA behind slot 27 of phase q: chunk q + 2, the pieces of chunk q + 3 issued so far — and with
        // a residual the four fragment loads of this and of the previous tile (K = 1024: one of the two phases of either tile has them)
        constexpr int VMW = (ND == 4 ? 8 : 15) + (HAS_R ? (PH == 1 ? 8 : 4) : 0);
        auto mid_barrier = [&]() {
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)" ::"n"(VMW) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        };
        // lim: fragments of this chunk that are multiplied at all (32, or 32 - NZ in a concat tile's last phase)
        auto slot_reads = [&](int m, lcptr_t cur, lcptr_t nxt, int lim) {
            auto rd = [&](int f) { if (f < lim) aw[(f >> 2) & 1][f & 3] = *(lfrag_t)(cur + f * 1024); };
            if (m < 24) rd(m + 4);
            if (m >= 20 && m < 24) rd(m + 8);
            if (m >= 28) aw[0][m - 28] = *(lfrag_t)(nxt + (m - 28) * 1024);
        };
        auto bias_quad = [&](f32x16& a, int nt, int qi) {
            const int col = nt * 32 + 16 * (qi >> 1) + 8 * h + 4 * (qi & 1);
            f32x4 b4 = *reinterpret_cast<const f32x4*>(sbias + col);
            if (HAS_C) {                                 // the accumulator starts from the bias + the CFG-null row constant, as in the loop below
                const f32x4 c4 = *reinterpret_cast<const f32x4*>(sconst + col);
#pragma unroll
                for (int e = 0; e < 4; ++e) b4[e] = fmaf(const_on, c4[e], b4[e]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) a[4 * qi + e] = b4[e];
        };
        // the epilogue of finished tile nte (accumulator E) in pieces: quad qi = 0..3 (4 values): step 0 reads its d / c vectors,
        // steps 1..4 finish one value each, step 5 (odd quads) packs the fragment of two quads and stores it
        struct Epi { f32x4 d4, c4; float v[8]; };
        auto epi_step = [&](int nte, const f32x16& E, const Res& rs, Epi& st, int qi, int step) {
            if (step == 0) {
                if (FOLD) {
                    const int col = nte * 32 + 16 * (qi >> 1) + 8 * h + 4 * (qi & 1);
                    st.d4 = *reinterpret_cast<const f32x4*>(sbias + col);
                    st.c4 = *reinterpret_cast<const f32x4*>(sconst + col);
                }
            } else if (step <= 4) {
                const int e = step - 1;
                float x = E[4 * qi + e];
                if (FOLD) x = fmaf(x, rstd, fmaf(nmr, st.c4[e], st.d4[e]));
                if (ACT == ACT_GELU) x = gelu_fast(x);
                else if (ACT == ACT_SILU) x = x * __builtin_amdgcn_rcpf(1.0f + __expf(-x));
                asm volatile("" : "+v"(x));
                st.v[4 * (qi & 1) + e] = x;
            } else if (qi & 1) {
                if (HL) {
                    const int c = qi >> 1;
                    hl_accumulate(st.v, rs.hi[c], rs.lo[c]);
                    u32x4 oh, ol;
                    hl_split(st.v, oh, ol);
                    const unsigned vo = (unsigned)(((size_t)tb * (2 * NT) + 2 * nte + c) * 1024 + lane_off);
                    asm_store16<0>(p.Ct, vo, oh);
                    asm_store16<0>(p.Clo, vo, ol);
                } else {
                    u32x4 o;
                    o.x = pack_bf16(st.v[0], st.v[1]); o.y = pack_bf16(st.v[2], st.v[3]);
                    o.z = pack_bf16(st.v[4], st.v[5]); o.w = pack_bf16(st.v[6], st.v[7]);
                    *reinterpret_cast<u32x4*>(Ctb + ((size_t)tb * (2 * NT) + 2 * nte + (qi >> 1)) * 1024 + lane_off) = o;
                }
            }
        };
        // slot of epilogue step k = 6 qi + step (k = 11 / 23 are the two stores) inside the tile that follows tile nte.  vmcnt counts
        // stores too, so the counted wait behind slot 27 of a phase also waits for every older store: a store must be as far in front of
        // the next such wait as the schedule allows (issued one slot before it, the barrier stalled for the store's acknowledgement).
        //   PH 1 (one wait per tile, slot 27): steps 0..10 in slots 2..12, the first store in slot 13 (14 slots before the wait), steps
        //        12..22 in slots 14..24, the second store in slot 28 — right BEHIND the wait;
        //   PH 2 (waits at slots 27 and 59): one step every second slot, the stores in slots 28 and 60, right behind the waits.
        auto epi_slot = [&](int nte, const f32x16& E, const Res& rs, Epi& st, auto s_tag) {
            constexpr int s = decltype(s_tag)::value;
            static_for<24>([&](auto k_tag) {
                constexpr int k = decltype(k_tag)::value;
                if constexpr (s == (PH == 1 ? (k == 23 ? 28 : 2 + k) : (k < 12 ? 6 + 2 * k : 14 + 2 * k))) epi_step(nte, E, rs, st, k / 6, k % 6);
            });
        };
        // one tile into W; the epilogue of the previous tile (nt - 1, accumulator E) rides along; !FOLD: the bias of tile nt + 1 is
        // read into E's registers in the last four slots
        auto tile = [&](int nt, f32x16& W, f32x16& E, Res& rw, const Res& re, auto prev_tag) {
            constexpr bool HAS_PREV = decltype(prev_tag)::value;
            Epi st;
            const int pprev = HAS_PREV ? ptile(nt - 1) : 0, pnext = ptile(nt + 1 < nt1 ? nt + 1 : nt);
            if (HL) load_res(rw, ptile(nt));              // this tile's residual: read by its epilogue, one tile from here
            static_for<PH>([&](auto k_tag) {
                constexpr int k = decltype(k_tag)::value;
                const int ph = nt * PH + k;
                const lcptr_t cur = chunk_base(ph), nxt = chunk_base(ph + 1);
                const int so_next = dma_soff(ph + 3);
                char* dst_next = dma_dst(ph + 3);
                static_for<32>([&](auto m_tag) {
                    constexpr int m = decltype(m_tag)::value;
                    constexpr int lim = (k == PH - 1) ? 32 - NZ : 32;       // (x 0 adds nothing: results are unchanged)
                    if constexpr (m < lim) {
                        const bf16x8 a = __builtin_bit_cast(bf16x8, aw[(m >> 2) & 1][m & 3]), b = __builtin_bit_cast(bf16x8, frag[k * 32 + m]);
                        if (FOLD && k == 0 && m == 0) {
                            const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                            W = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, z, 0, 0, 0);
                        } else W = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, W, 0, 0, 0);
                    }
                    slot_reads(m, cur, nxt, lim);
                    if (ND == 4) { if ((m & 7) == 1) dma_buf(m >> 3, wrsrc, wvoff, so_next, dst_next); }
                    else if ((m & 3) == 1) dma_buf(m >> 2, wrsrc, wvoff, so_next, dst_next);
                    if (HAS_PREV) epi_slot(pprev, E, re, st, std::integral_constant<int, k * 32 + m>{});
                    if (!FOLD && k == PH - 1 && m >= 28) bias_quad(E, pnext, m - 28);
                    __builtin_amdgcn_sched_barrier(0);
                    if (m == 27) mid_barrier();
                });
            });
        };
        if (!FOLD) {
#pragma unroll
            for (int qi = 0; qi < 4; ++qi) bias_quad(accA, ptile(nt0), qi);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) aw[0][i] = *reinterpret_cast<const u32x4*>(lds_lane_r + (p0 & 3) * T2_CHUNK + i * 1024);
        __builtin_amdgcn_sched_barrier(0);
        tile(nt0, accA, accB, resA, resB, std::false_type{});
        int nt = nt0 + 1;
        for (; nt + 1 < nt1; nt += 2) {
            tile(nt, accB, accA, resB, resA, std::true_type{});
            tile(nt + 1, accA, accB, resA, resB, std::true_type{});
        }
        Epi st;
        if (nt < nt1) {
            tile(nt, accB, accA, resB, resA, std::true_type{});
            const int pl = ptile(nt);
            static_for<32 * PH>([&](auto s2) { epi_slot(pl, accB, resB, st, s2); });
        } else {
            const int pl = ptile(nt1 - 1);
            static_for<32 * PH>([&](auto s2) { epi_slot(pl, accA, resA, st, s2); });
        }
        trace_mark(p.trace, 2);
        return;
    }
    const char* lds_lane = smem + lane * 16;
    f32x16 prev, acc;                                            // finished values of the previous tile (stored one tile later)
#pragma unroll
    for (int e = 0; e < 16; ++e) { prev[e] = 0.f; acc[e] = 0.f; }
    // the store of piece i (0..NSTORE-1) of tile nt from `v`: fp32 pieces qi = 0..3, then the two bf16 tiles.  Ordinary stores
    // cost nothing in the instantiations that run here: without residual loads hipcc has no load to wait for in the loop, so its
    // conservative "loads and stores may complete out of order -> vmcnt(0)" never triggers.
    auto store_piece = [&](int nt, const f32x16& v, int i) {
        constexpr int NF32 = (OUT & 5) ? 4 : 0;
        if (i < NF32) {
            const int qi = i;
            // (scalar copies first: __builtin_bit_cast applied directly to an ext-vector ELEMENT expression reads element 0 — hipcc 7.2)
            const float f0 = v[4 * qi], f1 = v[4 * qi + 1], f2 = v[4 * qi + 2], f3 = v[4 * qi + 3];
            u32x4 o;
            o.x = __builtin_bit_cast(uint32_t, f0); o.y = __builtin_bit_cast(uint32_t, f1);
            o.z = __builtin_bit_cast(uint32_t, f2); o.w = __builtin_bit_cast(uint32_t, f3);
            float* dst = (OUT & 4) ? p.Cf + (size_t)row * p.ldcf + nt * 32 + 16 * (qi >> 1) + 8 * h + 4 * (qi & 1)
                                   : p.Cf + fbase + (size_t)nt * fstride + qi * 256;
            *reinterpret_cast<u32x4*>(dst) = o;
        } else {
            const int c = i - NF32;
            u32x4 o;
            o.x = pack_bf16(v[8 * c + 0], v[8 * c + 1]); o.y = pack_bf16(v[8 * c + 2], v[8 * c + 3]);
            o.z = pack_bf16(v[8 * c + 4], v[8 * c + 5]); o.w = pack_bf16(v[8 * c + 6], v[8 * c + 7]);
            char* dst = Ctb + ((size_t)tb * (2 * NT) + 2 * nt + c) * 1024 + lane_off;
            *reinterpret_cast<u32x4*>(dst) = o;
        }
    };
    // loads younger than chunk p's DMA at the top of phase p: two more chunks and the residual tiles requested meanwhile
    constexpr int YWAIT = 2 * ND + (PH == 1 ? 2 * NR : NR);
    // one 32-feature tile; FT: the block's first tile (nothing to store yet) — peeled so that the hot loop has no branches
    auto do_tile = [&](int nt, auto ft_tag) {
        constexpr bool FT = decltype(ft_tag)::value;
#pragma unroll
        for (int k = 0; k < PH; ++k) {
            const bool first = k == 0, last = k == PH - 1;                   // compile-time after unrolling
            const int ph = nt * PH + k;
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YWAIT) : "memory");      // this wave's share of chunk ph has landed
            __builtin_amdgcn_s_barrier();                                       // ... everybody's has; slot (ph - 1) & 3 is free
            asm volatile("" ::: "memory");
            if (PROBE) { pp.stamp(0); pp.fold(); }
            const char* src_next = dma_src(ph + 3);
            char* dst_next = dma_dst(ph + 3);
            if (first) {
#pragma unroll
                for (int qi = 0; qi < 4; ++qi) {                 // the accumulator starts from the bias (+ CFG-null row constant)
                    const int col = nt * 32 + 16 * (qi >> 1) + 8 * h + 4 * (qi & 1);
                    f32x4 b4;
                    if (FOLD) { b4[0] = 0.f; b4[1] = 0.f; b4[2] = 0.f; b4[3] = 0.f; }
                    else b4 = *reinterpret_cast<const f32x4*>(sbias + col);
                    if (HAS_C) {
                        const f32x4 c4 = *reinterpret_cast<const f32x4*>(sconst + col);
#pragma unroll
                        for (int e = 0; e < 4; ++e) b4[e] = fmaf(const_on, c4[e], b4[e]);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[4 * qi + e] = b4[e];
                }
            }
            // 8 groups of 4 MFMAs; group g also carries its share of this phase's vector-memory instructions:
            //   tile-first phase: residual loads (groups 0..3), stores of the previous tile; every phase: the DMA of chunk ph + 3
            const char* cur = lds_lane + (ph & 3) * T2_CHUNK;
            u32x4 aw[2][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) aw[0][i] = *reinterpret_cast<const u32x4*>(cur + i * 1024);
            if (PROBE) pp.stamp(1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                if (g + 1 < 8) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) aw[(g + 1) & 1][i] = *reinterpret_cast<const u32x4*>(cur + ((g + 1) * 4 + i) * 1024);
                }
                if (first) {
                    if (HAS_R && g < 4) rres[g] = *reinterpret_cast<const f32x4*>(p.R + fbase + (size_t)nt * fstride + g * 256);
                    if (!FT) {                                    // the previous tile's stores, one per group, EARLY in the phase: the
                        if (g < NSTORE) store_piece(nt - 1, prev, g);   // next counted wait wants them acknowledged
                    }
                }
                if (ND == 4) { if (g & 1) dma_sel(g >> 1, src_next, dst_next); }
                else dma_sel(g, src_next, dst_next);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aw[g & 1][i]),
                                                                  __builtin_bit_cast(bf16x8, frag[k * 32 + g * 4 + i]), acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (PROBE) pp.stamp(2);
            if (last) {
#pragma unroll
                for (int qi = 0; qi < 4; ++qi) {
                    f32x4 d4, c4;
                    if (FOLD) {
                        const int col = nt * 32 + 16 * (qi >> 1) + 8 * h + 4 * (qi & 1);
                        d4 = *reinterpret_cast<const f32x4*>(sbias + col);
                        c4 = *reinterpret_cast<const f32x4*>(sconst + col);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc[4 * qi + e];
                        if (FOLD) v = fmaf(v, rstd, fmaf(nmr, c4[e], d4[e]));
                        if (ACT == ACT_GELU) v = gelu_fast(v);
                        else if (ACT == ACT_SILU) v = v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
                        if (HAS_R) v += rres[qi][e];
                        prev[4 * qi + e] = v;
                    }
                }
                // Materialise `prev` in VGPRs HERE.  For a pure-copy epilogue hipcc otherwise sinks the accumulator read
                // (v_accvgpr_read of the MFMA result) to the top of the next loop iteration, and its hazard recognizer does not
                // add the MFMA -> read wait states across the loop back-edge: ~2 % of the outputs were garbage (ffn.linear2).
                asm volatile("" : "+v"(prev));
            }
            if (PROBE) { pp.stamp(3); pp.roll(); }
        }
    };
    do_tile(nt0, std::true_type{});
    for (int nt = nt0 + 1; nt < nt1; ++nt) do_tile(nt, std::false_type{});
#pragma unroll
    for (int i = 0; i < NSTORE; ++i) store_piece(nt1 - 1, prev, i);
    trace_mark(p.trace, 2);
    if (PROBE) pp.dump(p.clk, pc0, pw0);
}

// =====================================================================================================================
// q|k|v-type Linear (K = 512, folded or no LayerNorm, bf16 output, no residual) with the EPILOGUES of the two wave groups of a
// 256-token block out of phase (round 4).  tl2_linear_kernel runs all eight waves through "32 MFMAs, epilogue, barrier" together:
// the phase probe (profiles/r03_h_tl2_microbench_phase_probe.log) shows 1892 cycles of MFMA groups per tile and wave, then 713
// cycles of epilogue and 533 at the wait + barrier during which the SIMD's matrix pipe idles for BOTH of its waves — 64 % busy.
// Here waves 0..3 (group A) run a tile as "MFMAs, epilogue" and waves 4..7 (group B; wave w + 4 shares wave w's SIMD) as
// "epilogue of the PREVIOUS tile, MFMAs": between two barriers one wave of a SIMD is in its epilogue (VALU, stores) while the
// other issues MFMAs, and only the middle of the interval has both on the matrix pipe.
// (Measured first, and rejected: strict antiphase — A = MFMA(c) while B = epilogue(c - 1), then swapped, one barrier per half
//  period — 326 vs 314 us: a wave ALONE on its SIMD cannot issue 32 MFMAs + their LDS reads in 1024 cycles (the fused FFN kernel's
//  finding), the pipe needs both waves' MFMA streams most of the time.)
template <int PRO, int ACT>
__global__ __launch_bounds__(512, 2) void tl2_linear_pp_kernel(TlArgs p) {
    constexpr int KD = 512, NW = 8, NTHR = NW * 64, ND = 32 / NW;
    constexpr bool FOLD = PRO == 1;
    static_assert(PRO == 0 || PRO == 1, "plain rows or folded LayerNorm");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    trace_mark(p.trace, 0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;                                   // 0: group A, 1: group B (wave uniform)
    const int ml = lane & 31, h = lane >> 5;
    const int bx = tl_block_index(p.rev);
    const int tb = bx * NW + wave;
    const int lane_off = ml * 32 + h * 16;
    const int NT = p.N / 32;
    const char* wsrc = reinterpret_cast<const char*>(p.W) + wave * (ND * 1024) + lane * 16;
    char* wdst = smem + wave * (ND * 1024);
    auto dma_src = [&](int q) -> const char* { const int c = q < NT ? q : NT - 1; return wsrc + (size_t)c * T2_CHUNK; };
    auto dma_dst = [&](int q) -> char* { return wdst + (q & 3) * T2_CHUNK; };
    dma_kbs<ND>(dma_src(0), dma_dst(0));
    dma_kbs<ND>(dma_src(1), dma_dst(1));
    constexpr int NFRAG = KD / 16;
    u32x4 frag[NFRAG];
    {
        const char* xr = reinterpret_cast<const char*>(p.X) + (size_t)tb * (p.ldx / 16) * 1024 + lane_off;
#pragma unroll
        for (int s = 0; s < NFRAG; ++s) frag[s] = *reinterpret_cast<const u32x4*>(xr + s * 1024);
    }
    float* sbias = reinterpret_cast<float*>(smem + 4 * T2_CHUNK);
    float* sconst = sbias + p.N;
    for (int i = tid; i < p.N; i += NTHR) {
        sbias[i] = p.bias ? p.bias[i] : 0.f;
        sconst[i] = p.row_const ? p.row_const[i] : 0.f;
    }
    float rstd = 1.f, nmr = 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (FOLD) {
        float sum, sq;
        row_moments_bf16<NFRAG>(frag, sum, sq);
        const float mean = sum / (float)KD;
        sq = fmaxf(sq - sum * mean, 0.f);
        rstd = 1.0f / sqrtf(sq / (float)KD + 1e-5f);
        nmr = -mean * rstd;
    }
#pragma unroll
    for (int s = 0; s < NFRAG; ++s) asm volatile("" ::"v"(frag[s]));
    __syncthreads();                                            // tables and the first two chunks visible
    dma_kbs<ND>(dma_src(2), dma_dst(2));
    trace_mark(p.trace, 1);

    char* Ctb = reinterpret_cast<char*>(p.Ct);
    const char* lds_lane = smem + lane * 16;
    constexpr int YWAIT = 2 * ND;
    // the top of a half period; `wait`: this wave's share of the chunk that half period 2c opens must have landed first
    auto top = [&](bool wait) {
        if (wait) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YWAIT) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    f32x16 acc;
    auto mfma_half = [&](int c, bool dma) {
        const char* src_next = dma_src(c + 3);
        char* dst_next = dma_dst(c + 3);
        const char* cur = lds_lane + (c & 3) * T2_CHUNK;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        if (!FOLD) {
#pragma unroll
            for (int qi = 0; qi < 4; ++qi) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(sbias + c * 32 + 16 * (qi >> 1) + 8 * h + 4 * (qi & 1));
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[4 * qi + e] = b4[e];
            }
        }
        u32x4 aw[2][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) aw[0][i] = *reinterpret_cast<const u32x4*>(cur + i * 1024);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (g + 1 < 8) {
#pragma unroll
                for (int i = 0; i < 4; ++i) aw[(g + 1) & 1][i] = *reinterpret_cast<const u32x4*>(cur + ((g + 1) * 4 + i) * 1024);
            }
            if (dma && (g & 1)) dma_sel(g >> 1, src_next, dst_next);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aw[g & 1][i]), __builtin_bit_cast(bf16x8, frag[g * 4 + i]), acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("" : "+v"(acc));                  // the accumulator is read HERE (MFMA wait states in straight-line code)
    };
    auto epi_half = [&](int c, bool dma) {
        if (dma) dma_kbs<ND>(dma_src(c + 4), dma_dst(c + 4));
        float v[16];
#pragma unroll
        for (int qi = 0; qi < 4; ++qi) {
            f32x4 d4, c4;
            if (FOLD) {
                const int col = c * 32 + 16 * (qi >> 1) + 8 * h + 4 * (qi & 1);
                d4 = *reinterpret_cast<const f32x4*>(sbias + col);
                c4 = *reinterpret_cast<const f32x4*>(sconst + col);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = acc[4 * qi + e];
                if (FOLD) x = fmaf(x, rstd, fmaf(nmr, c4[e], d4[e]));
                if (ACT == ACT_GELU) x = gelu_fast(x);
                else if (ACT == ACT_SILU) x = x * __builtin_amdgcn_rcpf(1.0f + __expf(-x));
                v[4 * qi + e] = x;
            }
        }
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            u32x4 o;
            o.x = pack_bf16(v[8 * cc + 0], v[8 * cc + 1]); o.y = pack_bf16(v[8 * cc + 2], v[8 * cc + 3]);
            o.z = pack_bf16(v[8 * cc + 4], v[8 * cc + 5]); o.w = pack_bf16(v[8 * cc + 6], v[8 * cc + 7]);
            *reinterpret_cast<u32x4*>(Ctb + ((size_t)tb * (2 * NT) + 2 * c + cc) * 1024 + lane_off) = o;
        }
    };
    if (grp == 0) {
        for (int c = 0; c < NT; ++c) {
            top(true);                                  // chunk c landed (every wave waits for its share, then the barrier)
            mfma_half(c, true);
            epi_half(c, false);
        }
        top(false);                                     // (B's last epilogue follows its last barrier)
    } else {
        top(true);
        mfma_half(0, true);
        for (int c = 1; c < NT; ++c) {
            top(true);
            epi_half(c - 1, false);
            mfma_half(c, true);
        }
        top(false);
        epi_half(NT - 1, false);
    }
    trace_mark(p.trace, 2);
}

// =====================================================================================================================
// FFN branch of a decoder layer for 128 tokens per block (one wave per SIMD, 32 tokens each):
//   g = GELU(h16 W1^T + b1); y2 = g W2^T + b2; h <- h + Linear3(SiLU(LN(y2) (1 + scale) + shift)) (+ next layer's CFG-null constant)
// Everything between the h16 load and the h store stays in the register file: the 1024-wide hidden is produced 32 features at
// a time (GEMM1, one 32-MFMA phase) and consumed as one K chunk of linear2 (GEMM2, one phase into the 16 resident
// accumulators = the wave's 32 x 512 y2); LayerNorm statistics come from the fp32 accumulators.
//
// One weight stream of 80 chunks of 32 KB (built by finalize(), see the order below) runs through the same four-slot LDS ring
// as tl2_linear_kernel: phase p = 32 MFMAs per wave on chunk p, the DMA of chunk p + 3 issued one instruction per MFMA group,
// counted wait at the top.  Phase order (= chunk order):
//   p = 0: GEMM1(0) | p = 2j - 1: GEMM1(j) with GELU(j - 1) in its MFMA shadow | p = 2j: GEMM2(j - 1) | ... | p = 62: GEMM2(30),
//   GELU(31) | p = 63: GEMM2(31) | p = 64 + t: Linear3 tile t.
// The fp32 residual h is loaded while the LayerNorm / FiLM / SiLU pass frees the y2 accumulators, and becomes the INITIAL VALUE
// of the 16 Linear3 accumulators: after one wait before the first Linear3 phase no load is ever waited for again, so the
// output stores (ordinary, visible to hipcc) cannot drag a conservative vmcnt(0) into the loop.
// LDS: [4][32 KB] ring | folded FiLM rows of up to FFN_MAXCLIP clips | b1 [1024] | b2, b3, row_const [512].
constexpr int FFN_MAXCLIP = 3;                   // clips a 128-token block may span (frames >= 64)
constexpr int FFN_CH = 32 * 1024;
constexpr int FFN_LDS = 4 * FFN_CH + FFN_MAXCLIP * 4096 + (1024 + 3 * 512) * 4;
constexpr int FFN_NQ = 64 + 16;

template <bool PROBE>
__global__ __launch_bounds__(256, 1) void tl2_ffn_kernel(Tl2FfnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    PhaseProbe pp;
    const unsigned long long pc0 = PROBE ? __builtin_readcyclecounter() : 0, pw0 = PROBE ? wall_clock64() : 0;
    trace_mark(p.trace, 0);
    start_stagger(p.stag_groups, p.stag_sleep);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ml = lane & 31, h = lane >> 5;
    const int bx = tl_block_index(p.rev);
    const int tb = bx * (TL_TOK / 32) + wave;
    const int row = tb * 32 + ml;
    const int lane_off = ml * 32 + h * 16;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.Wffn), 0, FFN_NQ * FFN_CH, 0x00020000);
    const int wvoff = wave * (FFN_CH / 4) + lane * 16;              // this lane's position inside every chunk
    char* wdst = smem + wave * (FFN_CH / 4);
    auto dma_soff = [&](int q) -> int { return (q < FFN_NQ ? q : FFN_NQ - 1) * FFN_CH; };
    auto dma_dst = [&](int q) -> char* { return wdst + (q & 3) * FFN_CH; };
    auto dma_chunk = [&](int q) {
#pragma unroll
        for (int k = 0; k < 8; ++k) dma_buf(k, wrsrc, wvoff, dma_soff(q), dma_dst(q));
    };
    dma_chunk(0);
    dma_chunk(1);
    // folded FiLM rows (A | B) of this block's clips
    f32x4 prm[FFN_MAXCLIP];
    int clip0;
    {
        const int rb = bx * TL_TOK, rrb = rb >= p.half_row0 ? rb - p.half_row0 : rb;
        clip0 = rrb / p.frames;
        const int nclip = (rrb + TL_TOK - 1) / p.frames - clip0 + 1;
#pragma unroll
        for (int c = 0; c < FFN_MAXCLIP; ++c) {
            const int cc = c < nclip ? c : nclip - 1;
            prm[c] = *reinterpret_cast<const f32x4*>(p.film + (size_t)((clip0 + cc) % p.bmod) * p.film_ld + p.film_off + tid * 4);
        }
    }
    // the small tables are requested BEFORE the 32 row fragments: their LDS copies, the block barrier and the 256 accumulator
    // writes of the y2 bias below then run while the row loads (the block's HBM burst) are still in flight
    float tb1[4], tb2[2], tb3[2], tbc[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) tb1[i] = p.b1[tid + 256 * i];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        tb2[i] = p.b2[tid + 256 * i];
        tb3[i] = p.b3[tid + 256 * i];
        tbc[i] = p.row_const ? p.row_const[tid + 256 * i] : 0.f;
    }
    u32x4 hfr[32];
    {
        const char* xr = reinterpret_cast<const char*>(p.X) + (size_t)tb * 32 * 1024 + lane_off;
#pragma unroll
        for (int s = 0; s < 32; ++s) hfr[s] = *reinterpret_cast<const u32x4*>(xr + s * 1024);
    }
    float* sprm = reinterpret_cast<float*>(smem + 4 * FFN_CH);
    float* sb1 = sprm + FFN_MAXCLIP * 1024;
    float* sb2 = sb1 + 1024;
    float* sb3 = sb2 + 512;
    float* sconst = sb3 + 512;
#pragma unroll
    for (int c = 0; c < FFN_MAXCLIP; ++c) *reinterpret_cast<f32x4*>(sprm + 1024 * c + 4 * tid) = prm[c];
#pragma unroll
    for (int i = 0; i < 4; ++i) sb1[tid + 256 * i] = tb1[i];
#pragma unroll
    for (int i = 0; i < 2; ++i) { sb2[tid + 256 * i] = tb2[i]; sb3[tid + 256 * i] = tb3[i]; sconst[tid + 256 * i] = tbc[i]; }
    __syncthreads();                                        // bias tables visible (the row loads are still in flight)
    f32x16 acc2[16];
#pragma unroll
    for (int ot = 0; ot < 16; ++ot)
#pragma unroll
        for (int qi = 0; qi < 4; ++qi) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(sb2 + ot * 32 + 16 * (qi >> 1) + 8 * h + 4 * (qi & 1));
#pragma unroll
            for (int e = 0; e < 4; ++e) acc2[ot][4 * qi + e] = b4[e];
        }
#pragma unroll
    for (int s = 0; s < 32; ++s) asm volatile("" ::"v"(hfr[s]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // rows and the first two chunks have landed
    __syncthreads();                                        // ... for every wave
    dma_chunk(2);
    trace_mark(p.trace, 1);

    const char* lds_lane = smem + lane * 16;
    // the top of every phase: this wave's share of chunk q has landed (two younger chunks may be in flight), then everybody's
    auto phase_top = [&](int q) {
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (PROBE) { pp.stamp(0); pp.fold(); pp.stamp(1); }
        (void)q;
    };
    auto phase_end = [&]() { if (PROBE) { pp.stamp(2); pp.stamp(3); pp.roll(); } };

    // ---- phase C ------------------------------------------------------------------------------------------------------------
    f32x16 hprev;                                           // newest hidden tile (pre-activation), copied out of the MFMA accumulator
    u32x4 gfr[2];                                           // GELU(hidden tile) as two B fragments (k steps 0 / 1 of a GEMM2 chunk)
    u32x4 gnx0;                                             // first fragment of the NEXT tile's GELU, built during GEMM2
#pragma unroll
    for (int e = 0; e < 16; ++e) hprev[e] = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c) { gfr[c][0] = 0; gfr[c][1] = 0; gfr[c][2] = 0; gfr[c][3] = 0; }
    gnx0[0] = 0; gnx0[1] = 0; gnx0[2] = 0; gnx0[3] = 0;
    auto pack8 = [&](const float* v) -> u32x4 {
        u32x4 o;
        o[0] = pack_bf16(v[0], v[1]); o[1] = pack_bf16(v[2], v[3]); o[2] = pack_bf16(v[4], v[5]); o[3] = pack_bf16(v[6], v[7]);
        return o;
    };
    // One value of the GELU, pinned to the place in the instruction stream where it is written: hipcc otherwise SINKS the whole
    // polynomial to its use (the pack after the last MFMA), where nothing hides it — that is what round 2 shipped.
    auto gelu_here = [&](float x) -> float { float y = gelu_fast(x); asm volatile("" : "+v"(y)); return y; };
    // The issue pattern of a phase (ONE scheduling region per phase): every MFMA is followed by its share of the other work — one
    // A-fragment read, up to three VALU instructions of the GELU, every fourth time one DMA piece.  A wave alone on its SIMD can
    // hide about five issue slots under a 32-cycle MFMA and not one more (MI355X_MICROARCH.md); clustered, the same instructions
    // cost their full issue time (round-3 microbenchmark: 1986 -> 1676 cycles per phase with buffer DMA + this interleave).
    auto phase_pattern = [&]() {
#pragma unroll
        for (int m = 0; m < 32; ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (m < 28) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if ((m & 3) == 1) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // GEMM1 phase q on hidden tile j.  GMODE 0: no GELU rides along (first phase); 1: the whole GELU of the previous hidden tile
    // (j == 1: no GEMM2 phase ran before); 2: its second half (values 8 .. 15 -> gfr[1]; the first half was built during the
    // preceding GEMM2 phase -> gnx0)
    auto gemm1 = [&](int q, int j, auto gmode_tag) {
        constexpr int GMODE = decltype(gmode_tag)::value;
        phase_top(q);
        const int so_next = dma_soff(q + 3);
        char* dst_next = dma_dst(q + 3);
        f32x16 acc1;
#pragma unroll
        for (int qi = 0; qi < 4; ++qi) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(sb1 + j * 32 + 16 * (qi >> 1) + 8 * h + 4 * (qi & 1));
#pragma unroll
            for (int e = 0; e < 4; ++e) acc1[4 * qi + e] = b4[e];
        }
        const char* cur = lds_lane + (q & 3) * FFN_CH;
        u32x4 aw[2][4];
        float gv[16];
#pragma unroll
        for (int i = 0; i < 4; ++i) aw[0][i] = *reinterpret_cast<const u32x4*>(cur + i * 1024);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (g + 1 < 8) {
#pragma unroll
                for (int i = 0; i < 4; ++i) aw[(g + 1) & 1][i] = *reinterpret_cast<const u32x4*>(cur + ((g + 1) * 4 + i) * 1024);
            }
            dma_buf(g, wrsrc, wvoff, so_next, dst_next);
            if (GMODE == 1) { gv[2 * g] = gelu_here(hprev[2 * g]); gv[2 * g + 1] = gelu_here(hprev[2 * g + 1]); }
            if (GMODE == 2) gv[8 + g] = gelu_here(hprev[8 + g]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aw[g & 1][i]), __builtin_bit_cast(bf16x8, hfr[g * 4 + i]), acc1, 0, 0, 0);
        }
        phase_pattern();
        if (GMODE == 1) { gfr[0] = pack8(gv); gfr[1] = pack8(gv + 8); }
        if (GMODE == 2) { gfr[0] = gnx0; gfr[1] = pack8(gv + 8); }
        hprev = acc1;
        asm volatile("" : "+v"(hprev));                     // the accumulator read happens HERE (MFMA wait states in straight-line code)
        phase_end();
    };
    // GEMM2 phase q: K chunk (32 hidden features, gfr) into the 16 resident accumulators.  WITH_HALF: the first half of the GELU of
    // the newest hidden tile (hprev, written by the GEMM1 phase just before) rides along -> gnx0
    auto gemm2 = [&](int q, auto half_tag) {
        constexpr bool WITH_HALF = decltype(half_tag)::value;
        phase_top(q);
        const int so_next = dma_soff(q + 3);
        char* dst_next = dma_dst(q + 3);
        const char* cur = lds_lane + (q & 3) * FFN_CH;
        u32x4 aw[2][4];
        float gn[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) aw[0][i] = *reinterpret_cast<const u32x4*>(cur + i * 1024);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 8; ++g) {                       // group g: output tiles 2 g, 2 g + 1 (x 2 k steps): fragments 4 g .. 4 g + 3
            if (g + 1 < 8) {
#pragma unroll
                for (int i = 0; i < 4; ++i) aw[(g + 1) & 1][i] = *reinterpret_cast<const u32x4*>(cur + ((g + 1) * 4 + i) * 1024);
            }
            dma_buf(g, wrsrc, wvoff, so_next, dst_next);
            if (WITH_HALF) gn[g] = gelu_here(hprev[g]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc2[2 * g + (i >> 1)] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aw[g & 1][i]), __builtin_bit_cast(bf16x8, gfr[i & 1]),
                                                                                  acc2[2 * g + (i >> 1)], 0, 0, 0);
        }
        phase_pattern();
        if (WITH_HALF) gnx0 = pack8(gn);
        phase_end();
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    typedef std::integral_constant<int, 2> I2;
    gemm1(0, 0, I0{});
    gemm1(1, 1, I1{});                                      // GELU(0) -> gfr
    gemm2(2, std::true_type{});                             // consumes hidden tile 0; first half of GELU(1)
    for (int j = 2; j < 32; ++j) {
        gemm1(2 * j - 1, j, I2{});                          // second half of GELU(j - 1) -> gfr
        gemm2(2 * j, std::true_type{});                     // consumes hidden tile j - 1; first half of GELU(j)
    }
    {   // the second half of the last hidden tile's GELU has no GEMM1 left to hide under: exposed once per block
        float gv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) gv[e] = gelu_fast(hprev[8 + e]);
        gfr[0] = gnx0; gfr[1] = pack8(gv);
    }
    gemm2(63, std::false_type{});
    if (PROBE) pp.dump(p.clk, pc0, pw0);                    // phase C only

    // ---- LayerNorm statistics from the fp32 accumulators; folded FiLM + SiLU; packed bf16 B fragments.  As the y2 tiles are
    //      consumed, the residual tiles are requested: they are the initial values of the Linear3 accumulators -----------------
    u32x4 yfr[32];
    f32x16 a3[16];
    const size_t fbase = ((size_t)tb * 16 * 4 * 64 + lane) * 4;           // + nt * 1024 floats + qi * 256
    {
        // the residual tiles are requested as the y2 accumulators are consumed (their registers become the Linear3 accumulators);
        // requesting the first half up front, into the registers the h16 fragments vacate, made hipcc spill 117 registers
        // (LayerNorm stage 35 k -> 49 k cycles), round 3
        auto load_r = [&](int ot) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 r4 = *reinterpret_cast<const f32x4*>(p.R + fbase + (size_t)ot * 1024 + q * 256);
#pragma unroll
                for (int e = 0; e < 4; ++e) a3[ot][4 * q + e] = r4[e];
            }
        };
        // raw moments in one pass over the accumulators (fp32: the cancellation error of E[x^2] - mean^2 is ~1e-7 (1 + mean^2 / var))
        float sum = 0.f, sq = 0.f;
#pragma unroll
        for (int ot = 0; ot < 16; ++ot)
#pragma unroll
            for (int e = 0; e < 16; ++e) { const float v = acc2[ot][e]; sum += v; sq = fmaf(v, v, sq); }
        sum += __shfl_xor(sum, 32, 64);
        sq += __shfl_xor(sq, 32, 64);
        const float mean = sum * (1.0f / 512.f);
        const float var = fmaxf(sq * (1.0f / 512.f) - mean * mean, 0.f);
        const float rstd = 1.0f / sqrtf(var + 1e-5f);
        const float nmr = -mean * rstd;
        const int rr = row >= p.half_row0 ? row - p.half_row0 : row;
        int ci = rr / p.frames - clip0;
        ci = ci < FFN_MAXCLIP ? ci : FFN_MAXCLIP - 1;
        const float* ca = sprm + ci * 1024 + 8 * h;
        const float* cb = ca + 512;
#pragma unroll
        for (int ot = 0; ot < 16; ++ot) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {             // fragment s = 2 ot + c holds features 32 ot + 16 c + 8 h + (0..7)
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(ca + 32 * ot + 16 * c), a1 = *reinterpret_cast<const f32x4*>(ca + 32 * ot + 16 * c + 4);
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(cb + 32 * ot + 16 * c), b1 = *reinterpret_cast<const f32x4*>(cb + 32 * ot + 16 * c + 4);
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float t = fmaf(acc2[ot][8 * c + e], rstd, nmr);
                    const float y = fmaf(t, e < 4 ? a0[e & 3] : a1[e & 3], e < 4 ? b0[e & 3] : b1[e & 3]);
                    v[e] = y * __builtin_amdgcn_rcpf(1.0f + __expf(-y));
                }
                u32x4 o;
                o.x = pack_bf16(v[0], v[1]); o.y = pack_bf16(v[2], v[3]); o.z = pack_bf16(v[4], v[5]); o.w = pack_bf16(v[6], v[7]);
                yfr[2 * ot + c] = o;
            }
            load_r(ot);
        }
    }
    // the one wait for the residual (a full drain of this wave's queue, paid once per block)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long pc_ln = PROBE ? __builtin_readcyclecounter() : 0;          // probe: end of the LayerNorm / SiLU stage

    // ---- phase D: h <- h + Linear3(yfr): accumulators start from the residual, bias (+ CFG-null constant) added in the epilogue.
    //      The epilogue + stores of tile t - 1 ride in the first MFMA groups of tile t (its accumulator a3[t - 1] stays put), so
    //      that at the next counted wait — which, counting only younger LOADS, implies that every older store has been
    //      acknowledged — they are already a phase old. -------------------------------------------------------------------------
    char* Ctb = reinterpret_cast<char*>(p.Ct);
    const float const_on = (p.row_const != nullptr && row < p.n_const_rows) ? 1.0f : 0.0f;
    // quad qi (fp32 piece) of tile t: add bias, store; returns the 4 values for the bf16 tile
    auto finish_quad = [&](int t, const f32x16& a, int qi, float* v4) {
        const int col = t * 32 + 16 * (qi >> 1) + 8 * h + 4 * (qi & 1);
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(sb3 + col);
        const f32x4 c4 = *reinterpret_cast<const f32x4*>(sconst + col);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[e] = a[4 * qi + e] + fmaf(const_on, c4[e], b4[e]); v4[e] = o[e]; }
        *reinterpret_cast<f32x4*>(p.Cf + fbase + (size_t)t * 1024 + qi * 256) = o;
    };
    auto store_bf16 = [&](int t, int c, const float* v8) {
        u32x4 o;
        o.x = pack_bf16(v8[0], v8[1]); o.y = pack_bf16(v8[2], v8[3]); o.z = pack_bf16(v8[4], v8[5]); o.w = pack_bf16(v8[6], v8[7]);
        *reinterpret_cast<u32x4*>(Ctb + ((size_t)tb * 32 + 2 * t + c) * 1024 + lane_off) = o;
    };
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int q = 64 + t;
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");          // two younger chunks may be in flight; older stores acknowledged
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int so_next = dma_soff(q + 3);
        char* dst_next = dma_dst(q + 3);
        const char* cur = lds_lane + (q & 3) * FFN_CH;
        u32x4 aw[2][4];
        float v8[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) aw[0][i] = *reinterpret_cast<const u32x4*>(cur + i * 1024);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (g + 1 < 8) {
#pragma unroll
                for (int i = 0; i < 4; ++i) aw[(g + 1) & 1][i] = *reinterpret_cast<const u32x4*>(cur + ((g + 1) * 4 + i) * 1024);
            }
            if (t > 0 && g < 4) {                                   // tile t - 1: quad g; bf16 tile c = g >> 1 once both its quads are done
                finish_quad(t - 1, a3[t > 0 ? t - 1 : 0], g, v8 + 4 * (g & 1));
                if (g & 1) store_bf16(t - 1, g >> 1, v8);
            }
            dma_buf(g, wrsrc, wvoff, so_next, dst_next);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                a3[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aw[g & 1][i]), __builtin_bit_cast(bf16x8, yfr[g * 4 + i]), a3[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);      // (a per-MFMA issue pattern like phase C's was measured SLOWER here: 38.6k vs 33k cycles)
        }
    }
    {
        float v8[8];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            finish_quad(15, a3[15], g, v8 + 4 * (g & 1));
            if (g & 1) store_bf16(15, g >> 1, v8);
        }
    }
    trace_mark(p.trace, 2);
    if (PROBE && p.clk && threadIdx.x == 0)             // word 7 of the probe record: cycles since block start at the end of the
        p.clk[(size_t)blockIdx.x * 8 + 7] = ((pc_ln - pc0) & 0xffffffffull) | ((__builtin_readcyclecounter() - pc0) << 32);   // LN stage | at the end
}

// ---- launchers -------------------------------------------------------------------------------------------------------
// DSH_STAGGER="groups,sleep[,mask]": first-round start stagger (tl_common.h); mask bit 0: fused FFN, 1: q|k|v, 2: the other tl2 Linears
void tl_stagger_config(int which, int* groups, int* sleep) {
    struct Cfg { int g = 0, sl = 0, mask = 7; };
    static const Cfg cfg = [] {                       // (a magic static: contexts launching from several host threads race on nothing)
        Cfg c;
        if (const char* e = getenv("DSH_STAGGER")) { int a = 0, b = 0, m = 7; const int n = sscanf(e, "%d,%d,%d", &a, &b, &m); if (n >= 2) { c.g = a; c.sl = b; if (n >= 3) c.mask = m; } }
        return c;
    }();
    const bool on = cfg.g > 1 && ((cfg.mask >> which) & 1);
    *groups = on ? cfg.g : 0; *sleep = on ? cfg.sl : 0;
}

int g_tl_last_variant = -1;

int launch_tl2_linear(const TlArgs& a, int pro, hipStream_t s) {
    DSH_REQUIRE(a.M > 0 && a.N > 0 && a.N % 32 == 0, "tl2_linear: N must be a positive multiple of 32");
    DSH_REQUIRE(a.K == 512 || a.K == 1024, "tl2_linear: K must be 512 or 1024");
    DSH_REQUIRE(a.ldx == (pro == 3 ? 512 : a.K), "tl2_linear: the tiled input must be exactly K features wide");
    DSH_REQUIRE(((uintptr_t)a.X % 16) == 0 && ((uintptr_t)a.W % 16) == 0, "tl2_linear: operands must be 16-byte aligned");
    DSH_REQUIRE(!a.Cf || !a.cf_rowmajor || a.ldcf % 4 == 0, "tl2_linear: row-major output leading dim");
    DSH_REQUIRE(!(a.cf_rowmajor && (a.R || a.Ct)), "tl2_linear: the row-major fp32 output has no residual / bf16 shadow");
    DSH_REQUIRE(pro >= 0 && pro <= 3, "tl2_linear: unknown prologue");
    // round 6: the LDS-tiled kernel class (tl4.hip) for feat_proj.1 / feat_proj.3 / q|k|v at whole-chip token counts — DSH_TL4 = bit mask
    // (1 feat_proj.1, 2 feat_proj.3, 4 q|k|v), read per launch (the op-level tests flip it inside one process); bit-identical results
    {
        const char* t4 = getenv("DSH_TL4");
        const int mask = t4 ? atoi(t4) : 0;
        const char* mr = getenv("DSH_TL4_MIN_ROWS");
        const int min_rows = mr ? atoi(mr) : 16384;
        const int bit = pro == 3 ? 1 : (pro == 0 ? 2 : (pro == 1 ? 4 : 0));
        if ((mask & bit) && a.M >= min_rows && !a.clk && !a.trace && tl4_linear_supported(a, pro)) return launch_tl4_linear(a, pro, s);
    }
    const int tok = a.K == 512 ? 256 : 128;            // tokens per block: row buffers must be allocated to a multiple of this
    DSH_REQUIRE(pro != 1 && pro != 3 || (a.bias && a.row_const), "tl2_linear: folded LayerNorm needs d (bias) and c (row_const) vectors");
    DSH_REQUIRE(pro != 2 || (a.film && a.frames > 0 && a.bmod > 0 && a.film_ld % 4 == 0 && a.film_off % 4 == 0),
                "tl2_linear: FiLM prologue needs the folded film table");
    DSH_REQUIRE(pro != 2 || std::min((tok - 1) / a.frames + 2, a.bmod) <= T2_MAXCLIP,
                "tl2_linear: FiLM prologue: too many clips per 256-token block (clips shorter than 29 frames need batch <= 10)");
    DSH_REQUIRE(pro != 2 || a.K == 512, "tl2_linear: FiLM prologue is instantiated for K = 512");
    DSH_REQUIRE(!a.row_const || pro == 1 || pro == 3 || (pro == 2 && a.R && a.act == ACT_NONE),
                "tl2_linear: row_const is the CFG-null constant of the StylizationBlock instantiation or the c vector of a folded LayerNorm");
    DSH_REQUIRE(pro != 3 || (a.K == 1024 && a.X1 && a.X2 && a.kreal > 896 - 1 && a.kreal <= 1024), "tl2_linear: concat prologue arguments");
    // N is split over grid.y only when the token blocks alone cannot fill the chip (window-chain batches)
    // (one block per CU is resident: the finest split whose grid still fits ONE round of 256 blocks — 276 blocks were measured
    //  at two rounds' cost for a 23-token-block launch)
    const int mblocks = ceil_div(a.M, tok), ntiles = a.N / 32;
    int tpb = ntiles;
    if (mblocks < 128) { tpb = 1; while (tpb < ntiles && mblocks * ceil_div(ntiles, tpb) > 256) ++tpb; }
    TlArgs b = a;
    b.tiles_per_block = tpb;
    { const char* re = getenv("DSH_TL2_ROT"); b.rot = (re && atoi(re) != 0) ? 1 : 0; }      // rotated weight-stream order per block (rolling loop only); read per launch
    tl_stagger_config(pro == 1 ? 1 : 2, &b.stag_groups, &b.stag_sleep);
    if (mblocks < 256) { b.stag_groups = 0; b.stag_sleep = 0; }
    const dim3 grid(mblocks, ceil_div(ntiles, tpb)), block(a.K == 512 ? 512 : 256);
    const int lds = 4 * T2_CHUNK + 2 * a.N * 4;
    DSH_REQUIRE(lds <= 160 * 1024, "tl2_linear: N too large for the LDS bias table");
    typedef void (*kern_t)(TlArgs);
    struct Variant { int k, pro, has_r, out, act; kern_t fn; };
#define TLV(P, R, O, A) {512, P, R, O, A, tl2_linear_kernel<512, P, (R) != 0, O, A>}
#define TLV1K(P, R, O, A) {1024, P, R, O, A, tl2_linear_kernel<1024, P, (R) != 0, O, A>}
    static const Variant variants[] = {
        TLV(1, 0, 2, ACT_NONE),   // sa_block: (folded) LayerNorm -> q|k|v               (bf16 out)
        TLV(2, 1, 3, ACT_NONE),   // StylizationBlock: LN+FiLM+SiLU -> Linear -> +h     (fp32 h + bf16 shadow)
        TLV(0, 0, 2, ACT_GELU),   // ffn.linear1 + GELU                                  (bf16 out)
        TLV(0, 0, 2, ACT_NONE), TLV(0, 1, 3, ACT_NONE), TLV(0, 0, 2, ACT_SILU), TLV(1, 0, 1, ACT_NONE),
        TLV(2, 0, 2, ACT_NONE), TLV(0, 1, 1, ACT_NONE), TLV(0, 0, 1, ACT_NONE),
        TLV(0, 0, 4, ACT_NONE),   // encoder `out` head: plain rows -> fp32 row-major
        TLV1K(0, 0, 2, ACT_NONE),  // ffn.linear2                                        (bf16 out)
        TLV1K(0, 0, 2, ACT_SILU),
        TLV1K(0, 1, 3, ACT_NONE),  // feat_proj.3 + residual                             (fp32 h + bf16 shadow)
        TLV1K(0, 0, 1, ACT_NONE), TLV1K(0, 1, 1, ACT_NONE),
        TLV1K(3, 0, 2, ACT_SILU),  // feat_proj: concat + (folded) LayerNorm -> Linear -> SiLU
    };
#undef TLV
#undef TLV1K
    constexpr int NV = sizeof(variants) / sizeof(variants[0]);
    static bool attr = false;
    if (!attr) {
        for (int i = 0; i < NV; ++i)
            DSH_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(variants[i].fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    const int out = (a.Cf ? (a.cf_rowmajor ? 4 : 1) : 0) | (a.Ct ? 2 : 0), has_r = a.R ? 1 : 0;
    kern_t fn = nullptr;
    int variant = 0;
    for (int i = 0; i < NV; ++i)
        if (variants[i].k == a.K && variants[i].pro == pro && variants[i].has_r == has_r && variants[i].out == out && variants[i].act == a.act) fn = variants[i].fn;
    if (a.clk) {     // bench only: phase-probe instantiations of the q|k|v, StylizationBlock and ffn.linear2 kernels
        kern_t pf = nullptr;
        if (a.K == 512 && pro == 1 && !has_r && out == 2) pf = tl2_linear_kernel<512, 1, false, 2, ACT_NONE, true>;
        if (a.K == 512 && pro == 2 && has_r && out == 3) pf = tl2_linear_kernel<512, 2, true, 3, ACT_NONE, true>;
        if (a.K == 1024 && pro == 0 && !has_r && out == 2 && a.act == ACT_NONE) pf = tl2_linear_kernel<1024, 0, false, 2, ACT_NONE, true>;
        if (pf) {
            DSH_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(pf), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            fn = pf;
        } else b.clk = nullptr;
    }
    DSH_REQUIRE(fn != nullptr || a.Rlo, "tl2_linear: this (prologue, residual, outputs, activation) combination is not instantiated");
    // out-of-phase epilogues (tl2_linear_pp_kernel) for the q|k|v-type instantiations at whole-chip token counts; DSH_TL2_PP=1: on
    // (measured, round 4: q|k|v alone 324.8 -> 310.5 us, but the 950-clip step 609.1 -> 612.5 ms on three streams — off by default)
    static const bool pp_on = [] { const char* e = getenv("DSH_TL2_PP"); return e && atoi(e) != 0; }();
    if (pp_on && a.K == 512 && !has_r && out == 2 && tpb == ntiles && ntiles >= 4 && !a.clk && (pro == 0 || pro == 1)) {
        kern_t pf = nullptr;
        if (pro == 1 && a.act == ACT_NONE) pf = tl2_linear_pp_kernel<1, ACT_NONE>;
        else if (pro == 0 && a.act == ACT_GELU) pf = tl2_linear_pp_kernel<0, ACT_GELU>;
        else if (pro == 0 && a.act == ACT_NONE) pf = tl2_linear_pp_kernel<0, ACT_NONE>;
        if (pf) {
            static bool pattr = false;
            if (!pattr) {
                DSH_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(tl2_linear_pp_kernel<1, ACT_NONE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                DSH_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(tl2_linear_pp_kernel<0, ACT_GELU>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                DSH_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(tl2_linear_pp_kernel<0, ACT_NONE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                pattr = true;
            }
            fn = pf;
            variant = 3;
        }
    }
    // rolling main loop (round 5) for the MFMA-bound bf16-out instantiations the step runs at whole-chip token counts: q|k|v (folded
    // LayerNorm) and feat_proj.1 (folded concat-LayerNorm, SiLU).  DSH_TL2_ROLL=0: the round-2 loop.  Results are bit-identical (same MFMA order, same epilogue expressions).
    const char* roll_e = getenv("DSH_TL2_ROLL");       // (read per launch: the op-level tests flip it inside one process)
    const bool roll_on = !(roll_e && atoi(roll_e) == 0);
    if (roll_on && !has_r && out == 2 && !b.clk && tpb == ntiles && ntiles >= 2) {
        kern_t rf = nullptr;
        if (a.K == 512 && pro == 1 && a.act == ACT_NONE) rf = tl2_linear_kernel<512, 1, false, 2, ACT_NONE, false, true>;
        else if (a.K == 1024 && pro == 3 && a.act == ACT_SILU) {
            // trailing all-zero fragments of the concat row are not multiplied (DSH_TL2_KSKIP=0: all 64)
            const char* ke = getenv("DSH_TL2_KSKIP");
            const int nz = (ke && atoi(ke) == 0) ? 0 : (1024 - round_up(a.kreal, 16)) / 16;
            rf = nz >= 8 ? (kern_t)tl2_linear_kernel<1024, 3, false, 2, ACT_SILU, false, true, false, 8>
               : nz >= 1 ? (kern_t)tl2_linear_kernel<1024, 3, false, 2, ACT_SILU, false, true, false, 1> : (kern_t)tl2_linear_kernel<1024, 3, false, 2, ACT_SILU, false, true>;
        }
        else if (a.K == 1024 && pro == 0 && a.act == ACT_NONE) rf = tl2_linear_kernel<1024, 0, false, 2, ACT_NONE, false, true>;   // ffn.linear2 (unfused path)
        if (rf) {
            static const bool rattr = [] {
                bool ok = true;
                auto set = [&](kern_t f) { ok &= hipFuncSetAttribute(reinterpret_cast<const void*>(f), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess; };
                set(tl2_linear_kernel<512, 1, false, 2, ACT_NONE, false, true>); set(tl2_linear_kernel<1024, 3, false, 2, ACT_SILU, false, true>);
                set(tl2_linear_kernel<1024, 3, false, 2, ACT_SILU, false, true, false, 8>); set(tl2_linear_kernel<1024, 3, false, 2, ACT_SILU, false, true, false, 1>);
                set(tl2_linear_kernel<1024, 0, false, 2, ACT_NONE, false, true>);
                return ok;
            }();
            DSH_REQUIRE(rattr, "tl2_linear: hipFuncSetAttribute failed for the rolling instantiations");
            fn = rf;
            variant = 1;
        }
    }
    // the two residual-carrying launches of a layer (StylizationBlock of the attention branch, feat_proj.3) with the residual stream as
    // hi / lo planes: rolling loop only (round 5; DSH_TL2_HL=0 keeps them on the first-generation kernels, tl_linear.hip)
    if (a.Rlo) {
        DSH_REQUIRE(a.R && a.Clo && a.Ct && !a.Cf && a.act == ACT_NONE && ((a.K == 512 && pro == 2) || (a.K == 1024 && pro == 0)),
                    "tl2_linear: hi / lo residual planes are instantiated for the StylizationBlock (K = 512) and feat_proj.3 (K = 1024) launches");
        kern_t hf = a.K == 512 ? (kern_t)tl2_linear_kernel<512, 2, true, 3, ACT_NONE, false, true, true> : (kern_t)tl2_linear_kernel<1024, 0, true, 3, ACT_NONE, false, true, true>;
        static const bool hattr = [] {
            return hipFuncSetAttribute(reinterpret_cast<const void*>(tl2_linear_kernel<512, 2, true, 3, ACT_NONE, false, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess &&
                   hipFuncSetAttribute(reinterpret_cast<const void*>(tl2_linear_kernel<1024, 0, true, 3, ACT_NONE, false, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
        }();
        DSH_REQUIRE(hattr, "tl2_linear: hipFuncSetAttribute failed for the hi / lo instantiations");
        b.clk = nullptr;
        fn = hf;
        variant = 2;
    }
    g_tl_last_variant = variant;
    hipLaunchKernelGGL(fn, grid, block, lds, s, b);
    DSH_HIP_CHECK(hipGetLastError());
    return 0;
}

bool tl2_ffn_supported(int M, int frames, int bmod) {
    // whole-chip token counts only (no N split), and at most FFN_MAXCLIP clips per 128-token block.  (Against the three separate
    // launches at 48 / 64 / 100 / 200 clips = 8.6 k / 11.3 k / 17.6 k / 35.2 k rows: -2.8 % / -2.5 % / +5.3 % / +5.2 %.  The limit stays
    // at 8192 rows all the same: the fused kernel keeps the hidden layer in fp32 registers where the separate launches round it to
    // bf16, so moving the limit moves which batch sizes agree bit for bit with their sub-batches.)
    return M >= 128 * 64 && std::min((TL_TOK - 1) / frames + 2, bmod) <= FFN_MAXCLIP;
}

int launch_tl2_ffn(const Tl2FfnArgs& a, hipStream_t s) {
    DSH_REQUIRE(a.M > 0 && a.X && a.Wffn && a.b1 && a.b2 && a.b3 && a.film && a.R && a.Cf && a.Ct, "tl2_ffn: null operand");
    DSH_REQUIRE(a.frames > 0 && a.bmod > 0 && a.film_ld % 4 == 0 && a.film_off % 4 == 0, "tl2_ffn: folded FiLM table");
    DSH_REQUIRE(std::min((TL_TOK - 1) / a.frames + 2, a.bmod) <= FFN_MAXCLIP, "tl2_ffn: too many clips per 128-token block");
    static bool attr = false;
    if (!attr) {
        DSH_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(tl2_ffn_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, FFN_LDS));
        DSH_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(tl2_ffn_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, FFN_LDS));
        attr = true;
    }
    Tl2FfnArgs b = a;
    tl_stagger_config(0, &b.stag_groups, &b.stag_sleep);
    if (a.clk) hipLaunchKernelGGL(tl2_ffn_kernel<true>, dim3(ceil_div(a.M, TL_TOK)), dim3(256), FFN_LDS, s, b);
    else hipLaunchKernelGGL(tl2_ffn_kernel<false>, dim3(ceil_div(a.M, TL_TOK)), dim3(256), FFN_LDS, s, b);
    DSH_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---- host-side packing ---------------------------------------------------------------------------------------------------
// element index (in bf16 units) of W'[32 nt + n][k] (rows already pi-permuted: n is the STORED row inside the tile) in the
// fragment-ordered copy of a [N, K] weight
size_t tl2_frag_index(int K, int nt, int n, int k) {
    const int s = k >> 4, hh = (k >> 3) & 1, j = k & 7;
    return (((size_t)nt * (K / 16) + s) * 64 + (n + 32 * hh)) * 8 + j;
}

}  // namespace dsh
