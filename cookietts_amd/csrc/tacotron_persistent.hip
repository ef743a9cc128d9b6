// Persistent Tacotron2-TM decoder loop for gfx950: ONE launch runs a whole block of decoder steps
// (model.py:668-767 per step) on all 256 CUs, instead of six dependent launches per step.
//
// Why: a decoder step is a chain of small dependent mat-vecs (27 M fp32 weights, batch <= 4).  As six launches per
// step it costs 85 us/step: every launch pays a dependent boundary + a cold ramp of its weight stream, and the two
// one-workgroup-per-utterance stages serialise the rest.  Here every workgroup (one per CU) owns a fixed slice of the
// work for the whole block of steps:
//   * workgroups 0..251 ("LSTM workgroups"): 5-6 attention-RNN units, 3-4 decoder-RNN units and 3-4 second-decoder-RNN
//     units each (4 gate rows per unit); plus one row of the query projection (workgroups < attention_dim), one or two
//     rows of the projection row set [mel | gate | first prenet layer folded through the mel projection] and one or
//     two rows of the second prenet layer;
//   * workgroups 252..255: the windowed location-sensitive attention of utterance b = workgroup - 252, nothing else.
// The columns of every mat-vec whose input is produced in the SAME step ("fresh": prenet -> attention RNN, context ->
// decoder RNN, decoder hidden -> second decoder RNN, and the small rows) are weight-stationary on the CU for the whole
// launch (LDS, the second decoder RNN's in registers); the columns whose input was produced earlier (recurrent states,
// previous context: 79 % of the weights) are streamed from L2 / Infinity Cache after the workgroup has published its
// result and before it polls for the next vector, into per-lane partial sums - off the critical path.
// Exchanges: a vector produced by many workgroups and needed by all (att_h, q, ctx, dec_h, d2_h, h1, prenet) is
// all-gathered through 8-byte {tag = step + 1, value} granules written with ONE agent-scope (write-through) store each
// and polled with agent-scope loads: the data is the flag, no fences, no grid barrier (MI355X_MICROARCH.md
// "handoff" / "allgather" rows; cdna_hip_programming.md Guideline 16 R2).  Every poll loop is bounded; on a timeout
// the workgroup records (code, workgroup, phase, step) in the control words and the whole grid drains.
// Granule tags are zeroed by the host wrapper before EVERY launch; the control words are sticky.
#include "tacotron_plan.h"

namespace ctts {
namespace {

using namespace taco;

constexpr int PD_WG = 256;          // workgroups == CUs of an MI355X
constexpr int PD_LWG = 252;         // LSTM workgroups; the last PD_NB are the attention workgroups
constexpr int PD_T = 512;           // threads per workgroup (8 waves, 2 per SIMD)
constexpr int PD_NB = 4;            // batch, padded
constexpr int PD_RA = 1280, PD_RD = 768, PD_P = 256, PD_DM = 512;
constexpr int PD_UA = 6, PD_UD = 4; // max LSTM units per workgroup: ceil(1280 / 252), ceil(768 / 252)
constexpr int PD_AMAX = 192, PD_TMAX = 1024, PD_W = 33, PD_FMAX = 32, PD_KMAX = 31;
// LDS vector store X of an LSTM workgroup: [b][n] per vector
constexpr int XP = 0, XCTX = XP + PD_NB * PD_P, XDEC = XCTX + PD_NB * PD_DM, XATT = XDEC + PD_NB * PD_RD,
              XD2 = XATT + PD_NB * PD_RA, XH1 = XD2 + PD_NB * PD_RD, X_FLOATS = XH1 + PD_NB * PD_P;
// weight-stationary LDS images, one float4 per lane: [row][j][lane]
constexpr int WFA = X_FLOATS;                                  // attention RNN, prenet columns: [24][1][64] float4
constexpr int WFD = WFA + 4 * PD_UA * 1 * 64 * 4;              // decoder RNN, context columns:   [16][2][64] float4
constexpr int WQ = WFD + 4 * PD_UD * 2 * 64 * 4;               // query row                        [5][64] float4
constexpr int WPR = WQ + 5 * 64 * 4;                           // two projection rows              [2][5][64] float4
constexpr int WW2 = WPR + 2 * 5 * 64 * 4;                      // two second-prenet rows           [2][1][64] float4
constexpr int LSTM_FLOATS = WW2 + 2 * 64 * 4;
constexpr int PD_DBG_SLOTS = 24;             // ctts_taco_decoder_persistent_debug: [PD_WG][64 steps][PD_DBG_SLOTS] stamps
constexpr unsigned PD_SPIN_LIMIT = 400000;   // polls per gather before giving up (~0.5 s)

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
typedef __attribute__((address_space(1))) unsigned gu32;

// Kernel arguments: three base pointers + 32-bit offsets (in floats / granules) - a struct of ~50 pointers would cost
// ~100 SGPRs and spill; most offsets are used once, at entry or exit.
struct PdArgs {
    const float* blob;          // packed weights (ctts_taco_decoder_pack)
    float* ws;                  // decoder workspace (state, memory, processed memory)
    u64* xb;                    // exchange granules + control words
    const unsigned char* keep;
    float *mel_out, *gate_out, *align_out;
    unsigned att_wih, att_whh, att_bih, att_bhh, dec_wih, dec_whh, dec_bih, dec_bhh, d2_wih, d2_whh, d2_bih, d2_bhh;
    unsigned Wq, Wproj, bproj, W2, v, Wloc, Wd, scalars;                                  // blob offsets
    unsigned memory, pm, lengths, att_h_in, dec_h_in, d2_h_in, att_h_out, dec_h_out, d2_h_out, att_c, dec_c, d2_c, ctx,
        prenet, w, cum, pos;                                                              // ws offsets
    unsigned g_p, g_atth, g_q, g_ctx, g_dech, g_d2h, g_h1, ctl;                           // xb offsets (u64 words)
    u64* dbg;                   // optional [PD_WG][64 steps][PD_DBG_SLOTS] stamps of s_memrealtime (100 MHz), NULL = off
    int A, F, K, R, n_mel, T, batch, nbc, step0, n_steps, max_steps, pd_rows;   // nbc: batch rows the workspace holds
};

__device__ __forceinline__ float pd_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
// hardware exp2 / rcp forms (abs error ~1e-7, far inside the 1e-4 mel bound; the libm calls cost ~1 us per cell here)
__device__ __forceinline__ float fast_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}
__device__ __forceinline__ float fast_tanh(float x) {
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * 2.8853900817779268f));
}

__device__ __forceinline__ void publish(u64* g, int idx, unsigned epoch, float v) {
    __hip_atomic_store((gu64*)g + idx, ((u64)epoch << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}

// All-gather receive: thread t owns granules t, t + 512, ...; a granule is accepted when its tag equals `epoch`.
// Returns false on timeout / abort (after recording it).  The caller follows with a workgroup barrier.
template <int NPT>
__device__ __forceinline__ bool gather(const u64* g, int count, float* dst, unsigned epoch, unsigned* ctl, int t,
                                       unsigned phase, unsigned step) {
    unsigned done = 0;
    for (unsigned spins = 0;; ++spins) {
        bool ok = true;
#pragma unroll
        for (int k = 0; k < NPT; ++k) {
            const int i = t + PD_T * k;
            if (i < count && !((done >> k) & 1u)) {
                const u64 x = __hip_atomic_load((const gu64*)g + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned)(x >> 32) == epoch) { dst[i] = __uint_as_float((unsigned)x); done |= 1u << k; }
                else ok = false;
            }
        }
        if (__all(ok)) return true;
        if ((spins & 255u) == 255u) {
            if (__hip_atomic_load((gu32*)ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
            if (spins > PD_SPIN_LIMIT) {
                if ((t & 63) == 0 &&
                    atomicCAS(ctl, 0u, 1u) == 0u) { ctl[1] = blockIdx.x; ctl[2] = phase; ctl[3] = step; }
                return false;
            }
        }
        __builtin_amdgcn_s_sleep(2);
    }
}

// 64-lane reductions on the DPP network (v_add_f32_dpp: quad_perm x2, row_half_mirror, row_mirror, row_bcast15,
// row_bcast31; the total lands in lanes 48..63 and is read back with v_readlane, so the result is wave-uniform): ~12
// issue slots per value instead of six ds_bpermute round trips through the LDS crossbar, which every phase of the step
// used to pay in sequence (360 ds_bpermute in the round-2 ISA).  N independent values interleave level by level.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_max(float v) {     // disabled rows / lanes see their own value
    return fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xf, false)));
}
template <int N>
__device__ __forceinline__ void wave_totals(float (&v)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = dpp_add<0xB1, 0xf>(v[i]);      // quad_perm [1,0,3,2]
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = dpp_add<0x4E, 0xf>(v[i]);      // quad_perm [2,3,0,1]
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = dpp_add<0x141, 0xf>(v[i]);     // row_half_mirror
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = dpp_add<0x140, 0xf>(v[i]);     // row_mirror: every lane holds its row's sum
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = dpp_add<0x142, 0xa>(v[i]);     // row_bcast15 into rows 1, 3
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = dpp_add<0x143, 0xc>(v[i]);     // row_bcast31 into rows 2, 3: row 3 = total
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[i]), 63));
}
__device__ __forceinline__ float wave_total(float v) {
    float a[1] = {v};
    wave_totals<1>(a);
    return a[0];
}
__device__ __forceinline__ float wave_max(float v) {
    v = dpp_max<0xB1, 0xf>(v); v = dpp_max<0x4E, 0xf>(v); v = dpp_max<0x141, 0xf>(v); v = dpp_max<0x140, 0xf>(v);
    v = dpp_max<0x142, 0xa>(v); v = dpp_max<0x143, 0xc>(v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Units of a cell with hidden size H on LSTM workgroup wg: 252 workgroups, the first H - 252 * (H / 252) take one more.
__device__ __forceinline__ int unit_count(int H, int wg) { return H / PD_LWG + (wg < H % PD_LWG ? 1 : 0); }
__device__ __forceinline__ int unit_first(int H, int wg) { return wg * (H / PD_LWG) + min(wg, H % PD_LWG); }

// Issue this wave's NR x NJ sixteen-byte weight loads of columns [col0, col0 + 256 NJ) of weight rows `rows` (< 0:
// the wave has fewer rows); the values are consumed by fma_rows, so the L2 / Infinity-Cache latency of the whole
// chunk is paid once.
template <int NR, int NJ>
__device__ __forceinline__ void issue_rows(const float* __restrict__ blob, unsigned mat, int ldw, int col0, const int (&rows)[NR],
                                           int lane, float4 (&w)[NR][NJ]) {
    // rows are wave-uniform: the row base is scalar arithmetic and every load shares one per-lane offset
    const unsigned lane_off = 4u * (unsigned)lane;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int row = __builtin_amdgcn_readfirstlane(rows[i]);
        const bool valid = row >= 0;                                      // wave-uniform
        const float* rp = blob + mat + (size_t)(valid ? row : 0) * ldw + col0;    // scalar
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            w[i][j] = valid ? *reinterpret_cast<const float4*>(rp + 256 * j + lane_off) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// acc[i][b] += sum over this lane's columns of w[i][j] . x[b][xoff + c]   (per-lane partial sums, no reduction)
template <int NR, int NJ>
__device__ __forceinline__ void fma_rows(const float4 (&w)[NR][NJ], const float* xs, int n, int xoff, int lane,
                                         float (&acc)[NR][PD_NB]) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        float4 x[PD_NB];
#pragma unroll
        for (int b = 0; b < PD_NB; ++b) x[b] = *reinterpret_cast<const float4*>(xs + b * n + xoff + 4 * (lane + 64 * j));
#pragma unroll
        for (int i = 0; i < NR; ++i)
#pragma unroll
            for (int b = 0; b < PD_NB; ++b)     // explicit fma chain: the same arithmetic in every inlined copy, so a block of
                                                // steps split over several launches reproduces one long launch bit for bit
                acc[i][b] = fmaf(w[i][j].w, x[b].w, fmaf(w[i][j].z, x[b].z, fmaf(w[i][j].y, x[b].y, fmaf(w[i][j].x, x[b].x, acc[i][b]))));
    }
}

template <int NR, int NJ>
__device__ __forceinline__ void early_rows(const float* __restrict__ blob, unsigned mat, int ldw, int col0, const int (&rows)[NR],
                                           const float* xs, int n, int xoff, int lane, float (&acc)[NR][PD_NB]) {
    float4 w[NR][NJ];
    issue_rows<NR, NJ>(blob, mat, ldw, col0, rows, lane, w);
    fma_rows<NR, NJ>(w, xs, n, xoff, lane, acc);
}

// gates[local row][b] = reduce over the wave (early partial sums + fresh columns) + bias
template <int NR, int NJ>
__device__ __forceinline__ void fresh_gates(const float4 (&wf)[NR][NJ], const float (&bias)[NR], const float* xs, int n,
                                            const float (&early)[NR][PD_NB], float (*gates)[PD_NB], int nrows, int wave,
                                            int lane) {
    float tot[NR][PD_NB];
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
        for (int b = 0; b < PD_NB; ++b) tot[i][b] = early[i][b];
    fma_rows<NR, NJ>(wf, xs, n, 0, lane, tot);
    wave_totals<NR * PD_NB>(reinterpret_cast<float (&)[NR * PD_NB]>(tot));
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int r = wave + 8 * i;
        if (r < nrows && lane < PD_NB)
            gates[r][lane] = (lane == 0 ? tot[i][0] : lane == 1 ? tot[i][1] : lane == 2 ? tot[i][2] : tot[i][3]) + bias[i];
    }
}

template <int NR>
__device__ __forceinline__ void zero_rows(float (&acc)[NR][PD_NB]) {
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
        for (int b = 0; b < PD_NB; ++b) acc[i][b] = 0.f;
}

// LSTM cell update of the workgroup's U units (layers.py:308-372, gate order i, f, g, o; local row = gate * U + unit);
// publishes h' at [b][first + unit].
__device__ __forceinline__ void cell_update(const float (*gates)[PD_NB], float (*cst)[PD_NB], float (*hown)[PD_NB], u64* g,
                                            unsigned epoch, int H, int U, int first, int t) {
    if (t < U * PD_NB) {
        const int u = t / PD_NB, b = t % PD_NB;
        const float gi = fast_sigmoid(gates[0 * U + u][b]), gf = fast_sigmoid(gates[1 * U + u][b]);
        const float gg = fast_tanh(gates[2 * U + u][b]), go = fast_sigmoid(gates[3 * U + u][b]);
        const float c = gf * cst[u][b] + gi * gg;
        const float h = go * fast_tanh(c);
        cst[u][b] = c;
        hown[u][b] = h;
        publish(g, b * H + first + u, epoch, h);
    }
}

// one row (NJ float4 per lane, LDS-resident) against NB staged vectors -> per-lane partial dot products
template <int NJ>
__device__ __forceinline__ void row_dots(const float* wrow, const float* xs, int n, int lane, float (&acc)[PD_NB]) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const float4 w = *reinterpret_cast<const float4*>(wrow + (j * 64 + lane) * 4);
#pragma unroll
        for (int b = 0; b < PD_NB; ++b) {
            const float4 x = *reinterpret_cast<const float4*>(xs + b * n + 4 * (lane + 64 * j));
            acc[b] = fmaf(w.w, x.w, fmaf(w.z, x.z, fmaf(w.y, x.y, fmaf(w.x, x.x, acc[b]))));
        }
    }
}

struct AttnLds {
    __attribute__((aligned(16))) float pmw[PD_W * PD_AMAX];
    float memw[PD_W * PD_DM];                     // the memory window (context operand), staged before the query arrives
    float loc[PD_W][PD_FMAX + 4];                 // rows 16-byte aligned: read as float4 broadcasts
    float wcat[2][PD_W + PD_KMAX - 1 + 1];
    float q[PD_AMAX];
    float en[64], wts[64];
    float w[PD_TMAX], cum[PD_TMAX];
    float ctx[PD_DM];
    float pos;
};
static_assert(PD_T == PD_DM, "the context pass maps one thread to one memory dimension");
static_assert(sizeof(AttnLds) <= LSTM_FLOATS * sizeof(float), "the attention scratch shares the LSTM workgroups' LDS");

#define PD_STAMP(k)                                                                                         \
    do {                                                                                                    \
        if constexpr (DBG) if (a.dbg && t == 0 && step - a.step0 < 64)                                      \
            a.dbg[((size_t)wg * 64 + (step - a.step0)) * PD_DBG_SLOTS + (k)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
// publish-time stamps of the waves that are not wave 0 (lane 0 of the publishing wave)
#define PD_STAMP_LANE0(k)                                                                                   \
    do {                                                                                                    \
        if constexpr (DBG) if (a.dbg && lane == 0 && step - a.step0 < 64)                                   \
            a.dbg[((size_t)wg * 64 + (step - a.step0)) * PD_DBG_SLOTS + (k)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)

// Windowed location-sensitive attention of utterance b (model.py:93-161, 49-65) on a dedicated 512-thread workgroup.
// Everything that does not change between steps lives on the CU for the whole launch: previous / cumulative weights and
// the position in LDS, this thread's 62 location-conv taps (filter t % 32) and this lane's column of the location-dense
// weight (attention dim `ad`) in registers.  Per step: one burst for the 33-row window of the processed memory, the
// location conv as 33 x 32 outputs over 16 position groups, the energies as a 33 x 32 x A contraction with the
// location features read as 16-byte LDS broadcasts, softmax on one wave, the context from a second burst (memory
// window), published as granules.
struct AttnRegs {
    float wl[2 * PD_KMAX];      // location conv taps [c][j] of filter t % 32
    float wd[PD_FMAX];          // location-dense weight column [f] of attention dim `ad`
    float va3[3];               // v[lane], v[lane + 64], v[lane + 128]: the energies pass
    int ad, pg;                 // location-dense pass: attention dim, position group (-1: this wave sits it out)
};

// Part 1, BEFORE the query of this step is known (it depends only on the previous step's weights and position, so it
// runs while the LSTM workgroups are still in their attention-RNN phase): window start, the bursts for the 33-row
// windows of the processed memory and of the memory, the location conv.  Returns the window start.
constexpr int PD_WH = (PD_W + 1) / 2;     // window positions per energies wave

__device__ __forceinline__ int pd_attention_pre(const PdArgs& a, AttnLds& s, const AttnRegs& r, int b) {
    const int t = threadIdx.x;
    const int W = 2 * a.R + 1, padk = (a.K - 1) / 2;
    const int len = reinterpret_cast<const int*>(a.ws + a.lengths)[b];
    float cur = s.pos;
    const float off = (a.blob + a.scalars)[0];
    if (off != 0.f) cur += off;
    cur = fminf(fmaxf(cur, (float)a.R), (float)(len - 1 - a.R));
    const int s0 = (int)rintf(fmaxf(cur - (float)a.R, 0.f));
    {
        const int a4 = a.A / 4;
        for (int i = t; i < W * a4; i += PD_T) {
            const int tt = i / a4, c4 = i % a4;
            const int pos = min(s0 + tt, a.T - 1);
            *reinterpret_cast<float4*>(s.pmw + tt * a.A + c4 * 4) =
                *reinterpret_cast<const float4*>((a.ws + a.pm) + ((size_t)b * a.T + pos) * a.A + c4 * 4);
        }
        for (int i = t; i < W * (PD_DM / 4); i += PD_T) {      // rows clamped: a masked row has weight exactly 0
            const int tt = i / (PD_DM / 4), d4 = i % (PD_DM / 4);
            const int pos = min(s0 + tt, a.T - 1);
            *reinterpret_cast<float4*>(s.memw + tt * PD_DM + d4 * 4) =
                *reinterpret_cast<const float4*>((a.ws + a.memory) + ((size_t)b * a.T + pos) * PD_DM + d4 * 4);
        }
        for (int i = t; i < 2 * (W + a.K - 1); i += PD_T) {
            const int c = i / (W + a.K - 1), j = i % (W + a.K - 1);
            const int pos = s0 - padk + j;
            const float* src = c == 0 ? s.w : s.cum;
            s.wcat[c][j] = (pos >= 0 && pos < a.T) ? src[pos] : 0.f;
        }
    }
    __syncthreads();
    {   // location conv (model.py:56-60): thread = (filter f, position group g of 16); taps out of registers
        const int f = t & 31, g = t >> 5;
        for (int tt = g; tt < W; tt += 16) {
            float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
#pragma unroll
            for (int j = 0; j + 1 < PD_KMAX; j += 2) {
                acc0 = fmaf(r.wl[j], j < a.K ? s.wcat[0][tt + j] : 0.f, acc0);
                acc1 = fmaf(r.wl[PD_KMAX + j], j < a.K ? s.wcat[1][tt + j] : 0.f, acc1);
                acc2 = fmaf(r.wl[j + 1], j + 1 < a.K ? s.wcat[0][tt + j + 1] : 0.f, acc2);
                acc3 = fmaf(r.wl[PD_KMAX + j + 1], j + 1 < a.K ? s.wcat[1][tt + j + 1] : 0.f, acc3);
            }
            acc0 = fmaf(r.wl[PD_KMAX - 1], PD_KMAX - 1 < a.K ? s.wcat[0][tt + PD_KMAX - 1] : 0.f, acc0);
            acc1 = fmaf(r.wl[2 * PD_KMAX - 1], PD_KMAX - 1 < a.K ? s.wcat[1][tt + PD_KMAX - 1] : 0.f, acc1);
            s.loc[tt][f] = (acc0 + acc2) + (acc1 + acc3);
        }
    }
    __syncthreads();
    // everything of the pre-activation that does not need the query: location-dense row of this lane's attention dim on
    // the location features (33 x 32 x A contraction, features read as 16-byte LDS broadcasts), added IN PLACE to the
    // staged processed-memory window (each (position, dim) element belongs to exactly one thread).  It used to run
    // after the query arrived (5.4 us of the step's critical path, LDS-broadcast bound); here it hides in the ~40 us
    // this workgroup waits for the query anyway.
    if (r.pg >= 0 && r.ad < a.A) {
        const int t0 = r.pg == 0 ? 0 : (W + 1) / 2, t1 = r.pg == 0 ? (W + 1) / 2 : W;
#pragma unroll 1
        for (int tt = t0; tt < t1; ++tt) {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
            for (int f4 = 0; f4 < PD_FMAX / 4; ++f4) {
                const float4 l = *reinterpret_cast<const float4*>(&s.loc[tt][4 * f4]);     // wave-uniform address: broadcast
                a0 = fmaf(r.wd[4 * f4 + 0], l.x, a0); a1 = fmaf(r.wd[4 * f4 + 1], l.y, a1);
                a2 = fmaf(r.wd[4 * f4 + 2], l.z, a2); a3 = fmaf(r.wd[4 * f4 + 3], l.w, a3);
            }
            s.pmw[tt * a.A + r.ad] += (a0 + a2) + (a1 + a3);
        }
    }
    return s0;      // (the caller's q gather ends with a workgroup barrier: pmw / memw are visible after it)
}

// Part 2, on the critical path between the query and the context: tanh + v-weighted sum over the attention dims,
// softmax, context, publish.
template <bool DBG>
__device__ __forceinline__ void pd_attention_post(const PdArgs& a, AttnLds& s, const AttnRegs& r, int b, int s0, unsigned epoch,
                                                  int step, u64* g_ctx) {
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int wg = blockIdx.x;
    const int W = 2 * a.R + 1;
    const int len = reinterpret_cast<const int*>(a.ws + a.lengths)[b];
    {   // energies (model.py:107-112): wave wv takes window positions wv, wv + 8, ...; a lane sums its three attention
        // dims (lane, lane + 64, lane + 128) first, so a position costs ONE 64-lane reduction, not three (round 2 / the
        // first round-3 cut: 17 reductions per wave; now at most 5)
        float qv[3], ev[5];
#pragma unroll
        for (int j = 0; j < 3; ++j) qv[j] = lane + 64 * j < a.A ? s.q[lane + 64 * j] : 0.f;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int tt = min(wv + 8 * i, W - 1);
            float e = 0.f;
#pragma unroll
            for (int j = 0; j < 3; ++j)     // pmw = processed memory + location term (pd_attention_pre)
                e += lane + 64 * j < a.A ? r.va3[j] * fast_tanh(s.pmw[tt * a.A + lane + 64 * j] + qv[j]) : 0.f;
            ev[i] = e;
        }
        wave_totals<5>(ev);
#pragma unroll
        for (int i = 0; i < 5; ++i)
            if (lane == 0 && wv + 8 * i < W) s.en[wv + 8 * i] = ev[i];
    }
    __syncthreads();
    PD_STAMP(3);
    if (wv == 0) {
        const int pos_l = s0 + lane;
        float e = -INFINITY;
        if (lane < W && pos_l < len && pos_l < a.T) e = s.en[lane];
        const float m = wave_max(e);
        // exp(e - m) on the hardware exp2 (abs error ~1e-7 of a value <= 1); masked lanes: exp2(-inf) = 0
        const float pexp = lane < W ? __builtin_amdgcn_exp2f((e - m) * 1.4426950408889634f) : 0.f;
        float sums[2] = {pexp, pexp * (float)(s0 + lane)};     // normaliser and expected position in one pass
        wave_totals<2>(sums);
        const float inv = 1.0f / sums[0];
        if (lane < W) s.wts[lane] = pexp * inv;
        if (lane == 0) {
            const float sf = pd_sigmoid((a.blob + a.scalars)[1]);
            s.pos = s.pos * sf + (sums[1] * inv) * (1.0f - sf);
        }
    }
    __syncthreads();
    PD_STAMP(4);
    {   // context = sum_t w[t] * memory[t] out of the staged window: one thread per dimension, two chains, no second pass
        const int d = t;      // PD_T == PD_DM
        float c0 = 0.f, c1 = 0.f;
#pragma unroll
        for (int tt = 0; tt + 1 < PD_W; tt += 2) {
            c0 = fmaf(tt < W ? s.wts[tt] : 0.f, s.memw[tt * PD_DM + d], c0);
            c1 = fmaf(tt + 1 < W ? s.wts[tt + 1] : 0.f, s.memw[(tt + 1) * PD_DM + d], c1);
        }
        c0 = fmaf(PD_W - 1 < W ? s.wts[PD_W - 1] : 0.f, s.memw[(PD_W - 1) * PD_DM + d], c0);
        const float c = c0 + c1;
        s.ctx[d] = c;
        publish(g_ctx, b * PD_DM + d, epoch, c);
    }
    PD_STAMP(5);
    for (int p = t; p < a.T; p += PD_T) {
        const float wgt = (p >= s0 && p < s0 + W) ? s.wts[p - s0] : 0.f;
        s.w[p] = wgt;
        s.cum[p] += wgt;
        a.align_out[((size_t)b * a.max_steps + step) * a.T + p] = wgt;
    }
    __syncthreads();
}

// ---- the four attention workgroups --------------------------------------------------------------------------------
template <bool DBG>
__device__ __forceinline__ void attention_workgroup(const PdArgs& a, AttnLds& att, int wg) {
    const int t = threadIdx.x;
    const int b = wg - PD_LWG;
    const bool real = b < a.batch;
    unsigned* ctl = reinterpret_cast<unsigned*>(a.xb + a.ctl);
    for (int p = t; p < a.T; p += PD_T) {
        att.w[p] = real ? (a.ws + a.w)[(size_t)b * a.T + p] : 0.f;
        att.cum[p] = real ? (a.ws + a.cum)[(size_t)b * a.T + p] : 0.f;
    }
    AttnRegs r;
    {
        const int f = t & 31, lane = t & 63, wv = t >> 6;
#pragma unroll
        for (int j = 0; j < PD_KMAX; ++j) {
            r.wl[j] = (f < a.F && j < a.K) ? (a.blob + a.Wloc)[(f * 2 + 0) * a.K + j] : 0.f;
            r.wl[PD_KMAX + j] = (f < a.F && j < a.K) ? (a.blob + a.Wloc)[(f * 2 + 1) * a.K + j] : 0.f;
        }
        r.pg = wv < 6 ? wv / 3 : -1;
        r.ad = (wv % 3) * 64 + lane;
        const bool live = r.pg >= 0 && r.ad < a.A;
#pragma unroll
        for (int ff = 0; ff < PD_FMAX; ++ff) r.wd[ff] = (live && ff < a.F) ? (a.blob + a.Wd)[(size_t)ff * a.A + r.ad] : 0.f;
#pragma unroll
        for (int j = 0; j < 3; ++j) r.va3[j] = lane + 64 * j < a.A ? (a.blob + a.v)[lane + 64 * j] : 0.f;
    }
    for (int i = t; i < PD_W * (PD_FMAX + 4); i += PD_T) (&att.loc[0][0])[i] = 0.f;     // filters >= F stay zero
    for (int d = t; d < PD_DM; d += PD_T) att.ctx[d] = 0.f;
    if (t == 0) att.pos = real ? (a.ws + a.pos)[b] : 0.f;
    __syncthreads();
    for (int step = a.step0; step < a.step0 + a.n_steps; ++step) {
        const unsigned epoch = (unsigned)step + 1u;
        const int par = step & 1;
        PD_STAMP(0);
        const int s0 = real ? pd_attention_pre(a, att, r, b) : 0;
        PD_STAMP(1);
        const bool ok_ = gather<1>((a.xb + a.g_q) + (size_t)par * PD_NB * a.A + (size_t)b * a.A, a.A, att.q, epoch, ctl, t, 2u,
                                   (unsigned)step);
        if (__syncthreads_or(ok_ ? 0 : 1)) return;
        PD_STAMP(2);
        if (real) {
            pd_attention_post<DBG>(a, att, r, b, s0, epoch, step, (a.xb + a.g_ctx) + (size_t)par * PD_NB * PD_DM);
        } else {
            for (int d = t; d < PD_DM; d += PD_T) publish((a.xb + a.g_ctx) + (size_t)par * PD_NB * PD_DM, b * PD_DM + d, epoch, 0.f);
        }
        PD_STAMP(6);
    }
    if (real) {
        for (int p = t; p < a.T; p += PD_T) { (a.ws + a.w)[(size_t)b * a.T + p] = att.w[p]; (a.ws + a.cum)[(size_t)b * a.T + p] = att.cum[p]; }
        if (t == 0) (a.ws + a.pos)[b] = att.pos;
        for (int d = t; d < PD_DM; d += PD_T) (a.ws + a.ctx)[b * PD_DM + d] = att.ctx[d];
    }
}

// DBG = true: the same kernel with the s_memrealtime stamps of ctts_taco_decoder_persistent_debug compiled in.  They are
// NOT in the product instantiation: a dozen conditional stores were enough to push the register allocator from 54 to
// ~160 spilled VGPRs (40.5 -> 45.7 us per step, measured), so a profiled run is ~10 % slower than the product.
template <bool DBG>
__global__ __launch_bounds__(PD_T, 2) void taco_persistent_kernel(const PdArgs a) {
    __shared__ __attribute__((aligned(16))) float L[LSTM_FLOATS];
    __shared__ float gates[4 * PD_UA][PD_NB];
    __shared__ float cA[PD_UA][PD_NB], cD[PD_UD][PD_NB], c2[PD_UD][PD_NB];
    __shared__ float hA[PD_UA][PD_NB], hD[PD_UD][PD_NB], h2[PD_UD][PD_NB];
    __shared__ float pown[2][PD_NB];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wg = blockIdx.x;
    unsigned* ctl = reinterpret_cast<unsigned*>(a.xb + a.ctl);
    if (__hip_atomic_load((gu32*)ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)
        return;                                  // an earlier launch on this exchange buffer gave up: stay down
    if (wg >= PD_LWG) {
        attention_workgroup<DBG>(a, *reinterpret_cast<AttnLds*>(L), wg);
        return;
    }
    float* X = L;
    const int I_att = PD_P + PD_DM + PD_RD, I_dec = PD_RA + PD_DM, Dp = PD_RD + PD_DM;
    const int UA = unit_count(PD_RA, wg), FA = unit_first(PD_RA, wg);
    const int UD = unit_count(PD_RD, wg), FD = unit_first(PD_RD, wg);

    // this wave's weight rows (row-per-wave: local rows wave, wave + 8, wave + 16; local row r = gate r / U of unit r % U)
    int rowA[3], rowD[2];
#pragma unroll
    for (int i = 0; i < 3; ++i) { const int r = wave + 8 * i; rowA[i] = r < 4 * UA ? (r / UA) * PD_RA + FA + r % UA : -1; }
#pragma unroll
    for (int i = 0; i < 2; ++i) { const int r = wave + 8 * i; rowD[i] = r < 4 * UD ? (r / UD) * PD_RD + FD + r % UD : -1; }

    // ---- weight-stationary part.  LDS: fresh columns of the attention / decoder RNN and the small rows; registers: the
    // second decoder RNN's fresh columns (its whole W_ih) and the biases.
    for (int i = t; i < 4 * UA * 64; i += PD_T) {
        const int r = i >> 6, l = i & 63;
        const int row = (r / UA) * PD_RA + FA + r % UA;
        *reinterpret_cast<float4*>(L + WFA + i * 4) = *reinterpret_cast<const float4*>((a.blob + a.att_wih) + (size_t)row * I_att + 4 * l);
    }
    for (int i = t; i < 4 * UD * 2 * 64; i += PD_T) {
        const int r = i >> 7, j = (i >> 6) & 1, l = i & 63;
        const int row = (r / UD) * PD_RD + FD + r % UD;
        *reinterpret_cast<float4*>(L + WFD + i * 4) =
            *reinterpret_cast<const float4*>((a.blob + a.dec_wih) + (size_t)row * I_dec + PD_RA + 4 * (l + 64 * j));
    }
    const int q_row = wg < a.A ? wg : -1;
    const int pr_row0 = wg, pr_row1 = wg + PD_LWG < a.pd_rows ? wg + PD_LWG : -1;
    const int w2_row0 = wg, w2_row1 = wg + PD_LWG < PD_P ? wg + PD_LWG : -1;
    for (int i = t; i < 5 * 64; i += PD_T) {
        const int j = i >> 6, l = i & 63;
        if (q_row >= 0)
            *reinterpret_cast<float4*>(L + WQ + i * 4) = *reinterpret_cast<const float4*>((a.blob + a.Wq) + (size_t)q_row * PD_RA + 4 * (l + 64 * j));
        *reinterpret_cast<float4*>(L + WPR + i * 4) = *reinterpret_cast<const float4*>((a.blob + a.Wproj) + (size_t)pr_row0 * Dp + 4 * (l + 64 * j));
        if (pr_row1 >= 0)
            *reinterpret_cast<float4*>(L + WPR + 5 * 64 * 4 + i * 4) =
                *reinterpret_cast<const float4*>((a.blob + a.Wproj) + (size_t)pr_row1 * Dp + 4 * (l + 64 * j));
    }
    if (t < 64) {
        *reinterpret_cast<float4*>(L + WW2 + t * 4) = *reinterpret_cast<const float4*>((a.blob + a.W2) + (size_t)w2_row0 * PD_P + 4 * t);
        if (w2_row1 >= 0)
            *reinterpret_cast<float4*>(L + WW2 + 64 * 4 + t * 4) = *reinterpret_cast<const float4*>((a.blob + a.W2) + (size_t)w2_row1 * PD_P + 4 * t);
    }
    float bA[3], bD[2], b2[2];
#pragma unroll
    for (int i = 0; i < 3; ++i) bA[i] = rowA[i] >= 0 ? (a.blob + a.att_bih)[rowA[i]] + (a.blob + a.att_bhh)[rowA[i]] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const bool valid = rowD[i] >= 0;
        const int row = valid ? rowD[i] : 0;
        bD[i] = valid ? (a.blob + a.dec_bih)[row] + (a.blob + a.dec_bhh)[row] : 0.f;
        b2[i] = valid ? (a.blob + a.d2_bih)[row] + (a.blob + a.d2_bhh)[row] : 0.f;
    }
    const float bpr = wave == 1 ? (a.blob + a.bproj)[pr_row0] : (wave == 2 && pr_row1 >= 0) ? (a.blob + a.bproj)[pr_row1] : 0.f;

    // ---- entry: state of step0 from the workspace (written by the init / the previous launch).  The workspace holds
    // nbc <= 4 batch rows per state array; the rows above stay zero here and are never written back.
    const int nbc = a.nbc;
    for (int i = t; i < PD_NB * PD_P; i += PD_T) X[XP + i] = i / PD_P < nbc ? (a.ws + a.prenet)[i] : 0.f;
    for (int i = t; i < PD_NB * PD_DM; i += PD_T) X[XCTX + i] = i / PD_DM < nbc ? (a.ws + a.ctx)[i] : 0.f;
    for (int i = t; i < PD_NB * PD_RD; i += PD_T) {
        X[XDEC + i] = i / PD_RD < nbc ? (a.ws + a.dec_h_in)[i] : 0.f;
        X[XD2 + i] = i / PD_RD < nbc ? (a.ws + a.d2_h_in)[i] : 0.f;
    }
    for (int i = t; i < PD_NB * PD_RA; i += PD_T) X[XATT + i] = i / PD_RA < nbc ? (a.ws + a.att_h_in)[i] : 0.f;
    if (t < UA * PD_NB) {
        const int u = t / PD_NB, b = t % PD_NB;
        cA[u][b] = b < nbc ? (a.ws + a.att_c)[b * PD_RA + FA + u] : 0.f;
        hA[u][b] = b < nbc ? (a.ws + a.att_h_in)[b * PD_RA + FA + u] : 0.f;
    }
    if (t < UD * PD_NB) {
        const int u = t / PD_NB, b = t % PD_NB;
        cD[u][b] = b < nbc ? (a.ws + a.dec_c)[b * PD_RD + FD + u] : 0.f;
        hD[u][b] = b < nbc ? (a.ws + a.dec_h_in)[b * PD_RD + FD + u] : 0.f;
        c2[u][b] = b < nbc ? (a.ws + a.d2_c)[b * PD_RD + FD + u] : 0.f;
        h2[u][b] = b < nbc ? (a.ws + a.d2_h_in)[b * PD_RD + FD + u] : 0.f;
    }
    if (t < 2 * PD_NB) {
        const int k = t / PD_NB, b = t % PD_NB;
        const int row = k == 0 ? w2_row0 : w2_row1;
        pown[k][b] = (row >= 0 && b < nbc) ? (a.ws + a.prenet)[b * PD_P + row] : 0.f;
    }
    __syncthreads();
    // per-lane partial sums of the early columns of the next cell evaluations (reduced over the wave only in the fresh
    // phase): one generation of each is live at a time.  Early parts of step0, in the order the loop accumulates them:
    float eA[3][PD_NB], eD[2][PD_NB], e2[2][PD_NB];
    zero_rows<3>(eA); zero_rows<2>(eD); zero_rows<2>(e2);
    early_rows<3, 3>(a.blob, a.att_whh, PD_RA, 0, rowA, X + XATT, PD_RA, 0, lane, eA);
    early_rows<3, 2>(a.blob, a.att_whh, PD_RA, 768, rowA, X + XATT, PD_RA, 768, lane, eA);
    early_rows<3, 2>(a.blob, a.att_wih, I_att, PD_P, rowA, X + XCTX, PD_DM, 0, lane, eA);
    early_rows<3, 3>(a.blob, a.att_wih, I_att, PD_P + PD_DM, rowA, X + XDEC, PD_RD, 0, lane, eA);
    early_rows<2, 3>(a.blob, a.dec_whh, PD_RD, 0, rowD, X + XDEC, PD_RD, 0, lane, eD);
    early_rows<2, 3>(a.blob, a.d2_whh, PD_RD, 0, rowD, X + XD2, PD_RD, 0, lane, e2);
    __syncthreads();

    bool fail = false;
#define PD_GATHER(NPT, buf, count, dst, phase)                                                                       \
    do {                                                                                                               \
        const bool ok_ = gather<NPT>((buf) + (size_t)par * (count), (count), (dst), epoch, ctl, t, (phase), (unsigned)step); \
        if (__syncthreads_or(ok_ ? 0 : 1)) { fail = true; }                                                            \
    } while (0)

    for (int step = a.step0; step < a.step0 + a.n_steps && !fail; ++step) {
        const unsigned epoch = (unsigned)step + 1u;
        const int par = step & 1;
        PD_STAMP(0);
        // the prenet's dropout keep-bytes of step + 1 for this wave's row (waves 1, 2: first layer, rows of the folded
        // projection; waves 3, 4: second layer), lane = batch item.  Loaded HERE, a whole step ahead of their use: they
        // are first-touch HBM bytes, and the load used to sit between the row's dot product and its publish - ~1.5 us
        // on the critical path of both prenet exchanges.
        const bool have_next = step + 1 < a.max_steps;
        unsigned char keep_next = 0;
        if (have_next && lane < a.batch) {
            if (wave == 1 || (wave == 2 && pr_row1 >= 0)) {
                const int j = (wave == 1 ? pr_row0 : pr_row1) - a.n_mel - 1;
                if (j >= 0) keep_next = a.keep[(((size_t)(step + 1) * 2 + 0) * a.batch + lane) * PD_P + j];
            } else if (wave == 3 || (wave == 4 && w2_row1 >= 0)) {
                keep_next = a.keep[(((size_t)(step + 1) * 2 + 1) * a.batch + lane) * PD_P + (wave == 3 ? w2_row0 : w2_row1)];
            }
        }
        // ---- A: attention RNN on the fresh prenet columns (model.py:707-717)
        {
            float4 wf[3][1];
#pragma unroll
            for (int i = 0; i < 3; ++i) wf[i][0] = *reinterpret_cast<const float4*>(L + WFA + (((wave + 8 * i) % (4 * PD_UA)) * 64 + lane) * 4);
            fresh_gates<3, 1>(wf, bA, X + XP, PD_P, eA, gates, 4 * UA, wave, lane);
        }
        zero_rows<3>(eA);
        __syncthreads();
        cell_update(gates, cA, hA, (a.xb + a.g_atth) + (size_t)par * PD_NB * PD_RA, epoch, PD_RA, UA, FA, t);
        PD_STAMP(1);
        // (weight loads whose vector operand is the one being gathered are issued BEFORE the gather: their L2 / Infinity
        // Cache latency runs out during the wait, the FMAs follow the barrier)
        float4 pfB[2][3], pfB2[2][2];
        issue_rows<2, 3>(a.blob, a.dec_wih, I_dec, 0, rowD, lane, pfB);
        issue_rows<2, 2>(a.blob, a.dec_wih, I_dec, 768, rowD, lane, pfB2);
        PD_GATHER(10, (a.xb + a.g_atth), PD_NB * PD_RA, X + XATT, 1u);
        PD_STAMP(2);
        if (fail) break;
        // ---- B: query row (model.py:126 query_layer); then, while the attention workgroups work, the columns that
        // multiply att_h(step): the decoder RNN's (needed in C) and the attention RNN's recurrent ones (step + 1)
        if (wave == 0 && q_row >= 0) {
            float q[PD_NB] = {0.f, 0.f, 0.f, 0.f};
            row_dots<5>(L + WQ, X + XATT, PD_RA, lane, q);
            wave_totals<PD_NB>(q);
            if (lane < PD_NB)
                publish((a.xb + a.g_q) + (size_t)par * PD_NB * a.A, lane * a.A + q_row, epoch,
                        lane == 0 ? q[0] : lane == 1 ? q[1] : lane == 2 ? q[2] : q[3]);
            PD_STAMP_LANE0(17);
        }
        fma_rows<2, 3>(pfB, X + XATT, PD_RA, 0, lane, eD);
        fma_rows<2, 2>(pfB2, X + XATT, PD_RA, 768, lane, eD);
        // the attention RNN's recurrent columns of step + 1 (W_hh . att_h(step)): streamed HERE, in front of the longest
        // wait of the step (query exchange + attention + context exchange), not between the projection's publish and
        // the prenet gather where they delayed a 1.5 us exchange by 3 us (scripts/micro/allgather_floor.hip: the
        // prenet-sized all-gather alone costs 1.5 us, the att_h-sized one 3.7)
        if (have_next) {
            early_rows<3, 3>(a.blob, a.att_whh, PD_RA, 0, rowA, X + XATT, PD_RA, 0, lane, eA);
            early_rows<3, 2>(a.blob, a.att_whh, PD_RA, 768, rowA, X + XATT, PD_RA, 768, lane, eA);
        }
        PD_STAMP(3);
        float4 pfC[3][2];
        issue_rows<3, 2>(a.blob, a.att_wih, I_att, PD_P, rowA, lane, pfC);
        PD_GATHER(4, (a.xb + a.g_ctx), PD_NB * PD_DM, X + XCTX, 3u);
        PD_STAMP(4);
        if (fail) break;
        // ---- C: decoder RNN on the fresh context columns (model.py:741-747)
        {
            float4 wf[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    wf[i][j] = *reinterpret_cast<const float4*>(L + WFD + ((((wave + 8 * i) % (4 * PD_UD)) * 2 + j) * 64 + lane) * 4);
            fresh_gates<2, 2>(wf, bD, X + XCTX, PD_DM, eD, gates, 4 * UD, wave, lane);
        }
        zero_rows<2>(eD);
        __syncthreads();
        cell_update(gates, cD, hD, (a.xb + a.g_dech) + (size_t)par * PD_NB * PD_RD, epoch, PD_RD, UD, FD, t);
        PD_STAMP(13);
        fma_rows<3, 2>(pfC, X + XCTX, PD_DM, 0, lane, eA);
        PD_STAMP(5);
        float4 wf2[2][3];      // the second decoder RNN's fresh columns (its whole W_ih, L2-resident): same trick
        issue_rows<2, 3>(a.blob, a.d2_wih, PD_RD, 0, rowD, lane, wf2);
        PD_GATHER(6, (a.xb + a.g_dech), PD_NB * PD_RD, X + XDEC, 4u);
        PD_STAMP(6);
        if (fail) break;
        // ---- D: second decoder RNN on the fresh decoder-hidden columns (model.py:749-755)
        fresh_gates<2, 3>(wf2, b2, X + XDEC, PD_RD, e2, gates, 4 * UD, wave, lane);
        zero_rows<2>(e2);
        __syncthreads();
        cell_update(gates, c2, h2, (a.xb + a.g_d2h) + (size_t)par * PD_NB * PD_RD, epoch, PD_RD, UD, FD, t);
        PD_STAMP(14);
        // the attention RNN's decoder-hidden columns of step + 1: streamed after the publish, inside the d2_h exchange (a
        // 3072-granule all-gather, ~2.5 us alone), instead of being held in 36 registers across the dec_h gather
        early_rows<3, 3>(a.blob, a.att_wih, I_att, PD_P + PD_DM, rowA, X + XDEC, PD_RD, 0, lane, eA);
        PD_STAMP(7);
        float4 pfE[2][3], pfE2[2][3];      // recurrent columns of both decoder RNNs for step + 1
        issue_rows<2, 3>(a.blob, a.d2_whh, PD_RD, 0, rowD, lane, pfE);
        issue_rows<2, 3>(a.blob, a.dec_whh, PD_RD, 0, rowD, lane, pfE2);
        PD_GATHER(6, (a.xb + a.g_d2h), PD_NB * PD_RD, X + XD2, 5u);
        PD_STAMP(8);
        if (fail) break;
        // ---- E: projection row set on [dec_h + d2_h | ctx] (model.py:757-765; rows: mel, gate, folded prenet layer 1)
        if (wave == 1 || (wave == 2 && pr_row1 >= 0)) {
            const int row = wave == 1 ? pr_row0 : pr_row1;
            const float* wrow = L + WPR + (wave == 1 ? 0 : 5 * 64 * 4);
            float o[PD_NB] = {0.f, 0.f, 0.f, 0.f}, o2[PD_NB] = {0.f, 0.f, 0.f, 0.f};
            row_dots<3>(wrow, X + XDEC, PD_RD, lane, o);                  // W[:, :768] . dec_h
            row_dots<3>(wrow, X + XD2, PD_RD, lane, o2);                  // W[:, :768] . d2_h   (the residual sum, model.py:755)
            row_dots<2>(wrow + 3 * 64 * 4, X + XCTX, PD_DM, lane, o);     // W[:, 768:] . ctx
#pragma unroll
            for (int b = 0; b < PD_NB; ++b) o[b] += o2[b];
            wave_totals<PD_NB>(o);
#pragma unroll
            for (int b = 0; b < PD_NB; ++b) o[b] += bpr;
            if (lane < PD_NB) {
                const int b = lane;
                const float val = lane == 0 ? o[0] : lane == 1 ? o[1] : lane == 2 ? o[2] : o[3];
                if (row < a.n_mel) {
                    if (b < a.batch) a.mel_out[((size_t)b * a.n_mel + row) * a.max_steps + step] = val;
                } else if (row == a.n_mel) {
                    if (b < a.batch) a.gate_out[(size_t)b * a.max_steps + step] = val;
                } else if (have_next) {     // first prenet layer of step + 1: relu, always-on dropout (model.py:187-190)
                    const int j = row - a.n_mel - 1;
                    const bool kp = keep_next != 0;        // (lanes >= batch loaded nothing: 0)
                    publish((a.xb + a.g_h1) + (size_t)par * PD_NB * PD_P, b * PD_P + j, epoch, kp ? fmaxf(val, 0.f) * 2.0f : 0.0f);
                }
            }
            if (wave == 1) PD_STAMP_LANE0(15);
        }
        fma_rows<2, 3>(pfE, X + XD2, PD_RD, 0, lane, e2);
        fma_rows<2, 3>(pfE2, X + XDEC, PD_RD, 0, lane, eD);
        if (have_next) {
            // the two prenet exchanges below are the shortest of the step (1.5 us alone): nothing is streamed in them
            PD_STAMP(9);
            PD_GATHER(2, (a.xb + a.g_h1), PD_NB * PD_P, X + XH1, 6u);
            PD_STAMP(10);
            if (fail) break;
            // ---- F: second prenet layer rows
            if (wave == 3 || (wave == 4 && w2_row1 >= 0)) {
                const int k = wave - 3;
                const int row = k == 0 ? w2_row0 : w2_row1;
                float o[PD_NB] = {0.f, 0.f, 0.f, 0.f};
                row_dots<1>(L + WW2 + k * 64 * 4, X + XH1, PD_P, lane, o);
                wave_totals<PD_NB>(o);
                if (lane < PD_NB) {
                    const int b = lane;
                    const float val = lane == 0 ? o[0] : lane == 1 ? o[1] : lane == 2 ? o[2] : o[3];
                    const bool kp = keep_next != 0;
                    const float pv = kp ? fmaxf(val, 0.f) * 2.0f : 0.0f;
                    pown[k][b] = pv;
                    publish((a.xb + a.g_p) + (size_t)par * PD_NB * PD_P, b * PD_P + row, epoch, pv);
                }
                if (wave == 3) PD_STAMP_LANE0(16);
            }
            PD_STAMP(11);
            PD_GATHER(2, (a.xb + a.g_p), PD_NB * PD_P, X + XP, 7u);
            PD_STAMP(12);
            if (fail) break;
        }
    }
#undef PD_GATHER
    if (fail) return;
    // ---- exit: this workgroup's slices of the state for the next launch
    __syncthreads();
    if (t < UA * PD_NB) {
        const int u = t / PD_NB, b = t % PD_NB;
        if (b < nbc) {
            (a.ws + a.att_h_out)[b * PD_RA + FA + u] = hA[u][b];
            (a.ws + a.att_c)[b * PD_RA + FA + u] = cA[u][b];
        }
    }
    if (t < UD * PD_NB) {
        const int u = t / PD_NB, b = t % PD_NB;
        if (b < nbc) {
            (a.ws + a.dec_h_out)[b * PD_RD + FD + u] = hD[u][b];
            (a.ws + a.dec_c)[b * PD_RD + FD + u] = cD[u][b];
            (a.ws + a.d2_h_out)[b * PD_RD + FD + u] = h2[u][b];
            (a.ws + a.d2_c)[b * PD_RD + FD + u] = c2[u][b];
        }
    }
    if (t < 2 * PD_NB) {
        const int k = t / PD_NB, b = t % PD_NB;
        const int row = k == 0 ? w2_row0 : w2_row1;
        if (row >= 0 && b < nbc) (a.ws + a.prenet)[b * PD_P + row] = pown[k][b];
    }
}
#undef PD_STAMP
#undef PD_STAMP_LANE0

void* g_pd_debug = nullptr;      // ctts_taco_decoder_persistent_debug: stamp buffer, not part of the product path

struct Xchg { size_t ctl, p, atth, q, ctx, dech, d2h, h1, total; };   // offsets in u64 words

void xchg_layout(int A, Xchg& x) {
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o += (n + 15) / 16 * 16; return r; };
    x.p = take(2 * PD_NB * PD_P);
    x.atth = take(2 * PD_NB * PD_RA);
    x.q = take((size_t)2 * PD_NB * A);
    x.ctx = take(2 * PD_NB * PD_DM);
    x.dech = take(2 * PD_NB * PD_RD);
    x.d2h = take(2 * PD_NB * PD_RD);
    x.h1 = take(2 * PD_NB * PD_P);
    x.ctl = o;                // control words = exactly the LAST 8 words (64 bytes) of the buffer, as the header says:
    o += 8;                   // sticky across launches (zeroed by the owner of the buffer, once)
    x.total = o;
}

bool pd_supported(const DecPlan& p, int batch, int T) {
    const auto& c = p.c;
    return c.attention_rnn_dim == PD_RA && c.decoder_rnn_dim == PD_RD && c.second_decoder_rnn_dim == PD_RD &&
           c.prenet_dim == PD_P && c.memory_dim == PD_DM && c.attention_dim <= PD_AMAX && c.attention_dim % 4 == 0 &&
           c.location_n_filters <= PD_FMAX && c.location_kernel_size <= PD_KMAX && c.window_range == 16 &&
           p.pd_rows <= 2 * PD_LWG && c.n_mel_channels + 1 <= PD_LWG && batch >= 1 && batch <= PD_NB && T >= 1 && T <= PD_TMAX;
}

}  // namespace
}  // namespace ctts

using namespace ctts;

extern "C" {

size_t ctts_taco_decoder_persistent_bytes(const ctts_taco_decoder_config* cfg, int32_t batch, int32_t text_len) {
    DecPlan p;
    if (make_dec_plan(cfg, p) || !pd_supported(p, batch, text_len)) return 0;
    // a current device with fewer than 256 CUs (a CPX / DPX partition) cannot keep the grid co-resident: per-launch form.
    // Without any device (a host-only size query) the answer is the shape's.
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus < PD_WG) {
        (void)hipGetLastError();
        return 0;
    }
    (void)hipGetLastError();
    Xchg x;
    xchg_layout(p.c.attention_dim, x);
    return x.total * sizeof(u64);
}

int ctts_taco_decoder_steps_persistent_f32(const ctts_taco_decoder_config* cfg, const void* packed,
                                           const uint8_t* keep_masks, float* mel_out, float* gate_out, float* align_out,
                                           int32_t batch, int32_t text_len, int32_t step0, int32_t n_steps,
                                           int32_t max_steps, void* workspace, void* exchange, size_t exchange_bytes,
                                           void* stream) {
    DecPlan p; DecWs w;
    int rc = make_dec_plan(cfg, p); if (rc) return rc;
    CTTS_CHECK_ARG(p.total < (1ull << 32), "persistent decoder: packed blob too large for 32-bit offsets");
    CTTS_CHECK_ARG(packed && keep_masks && mel_out && gate_out && align_out && workspace && exchange,
                   "persistent decoder: NULL pointer");
    CTTS_CHECK_ARG(pd_supported(p, batch, text_len), "persistent decoder: shape not built (repo-default decoder, batch <= 4, "
                                                      "text <= %d symbols only)", PD_TMAX);
    CTTS_CHECK_ARG(step0 >= 0 && n_steps >= 1 && step0 + n_steps <= max_steps, "persistent decoder: step range");
    int dev = 0, cus = 0;
    CTTS_CHECK_HIP(hipGetDevice(&dev));
    CTTS_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    CTTS_CHECK_ARG(cus >= PD_WG, "persistent decoder: needs %d CUs resident at once, device has %d", PD_WG, cus);
    Xchg x;
    xchg_layout(p.c.attention_dim, x);
    if (x.total * sizeof(u64) > exchange_bytes) {
        set_error("persistent decoder: exchange buffer %zu bytes < required %zu", exchange_bytes, x.total * sizeof(u64));
        return CTTS_E_WORKSPACE;
    }
    dec_carve(p, batch, text_len, static_cast<float*>(workspace), w);
    hipStream_t s = as_stream(stream);
    const float* blob = static_cast<const float*>(packed);
    u64* xb = static_cast<u64*>(exchange);
    CTTS_CHECK_HIP(hipMemsetAsync(exchange, 0, x.ctl * sizeof(u64), s));        // every tag, before every launch
    const auto& c = p.c;
    PdArgs a{};
    a.blob = blob; a.ws = static_cast<float*>(workspace); a.xb = xb;
    auto wo = [&](const float* q) { return (unsigned)(q - static_cast<float*>(workspace)); };
    a.att_wih = (unsigned)p.att[0]; a.att_whh = (unsigned)p.att[1]; a.att_bih = (unsigned)p.att[2]; a.att_bhh = (unsigned)p.att[3];
    a.dec_wih = (unsigned)p.dec[0]; a.dec_whh = (unsigned)p.dec[1]; a.dec_bih = (unsigned)p.dec[2]; a.dec_bhh = (unsigned)p.dec[3];
    a.d2_wih = (unsigned)p.d2[0]; a.d2_whh = (unsigned)p.d2[1]; a.d2_bih = (unsigned)p.d2[2]; a.d2_bhh = (unsigned)p.d2[3];
    a.Wq = (unsigned)p.query_w; a.Wproj = (unsigned)p.pd_proj_w; a.bproj = (unsigned)p.pd_proj_b; a.W2 = (unsigned)p.pd_w2;
    a.v = (unsigned)p.v_w; a.Wloc = (unsigned)p.loc_conv_w; a.Wd = (unsigned)p.loc_dense_w; a.scalars = (unsigned)p.scalars;
    a.memory = wo(w.memory); a.pm = wo(w.pm); a.lengths = wo(reinterpret_cast<float*>(w.lengths));
    const int cur = step0 & 1, fin = (step0 + n_steps) & 1;
    a.att_h_in = wo(w.att_h[cur]); a.dec_h_in = wo(w.dec_h[cur]); a.d2_h_in = wo(w.d2_h[cur]);
    a.att_h_out = wo(w.att_h[fin]); a.dec_h_out = wo(w.dec_h[fin]); a.d2_h_out = wo(w.d2_h[fin]);
    a.att_c = wo(w.att_c); a.dec_c = wo(w.dec_c); a.d2_c = wo(w.d2_c); a.ctx = wo(w.ctx); a.prenet = wo(w.prenet);
    a.w = wo(w.w); a.cum = wo(w.cum); a.pos = wo(w.pos);
    a.ctl = (unsigned)x.ctl;
    a.g_p = (unsigned)x.p; a.g_atth = (unsigned)x.atth; a.g_q = (unsigned)x.q; a.g_ctx = (unsigned)x.ctx;
    a.g_dech = (unsigned)x.dech; a.g_d2h = (unsigned)x.d2h; a.g_h1 = (unsigned)x.h1;
    a.keep = keep_masks; a.mel_out = mel_out; a.gate_out = gate_out; a.align_out = align_out;
    a.A = c.attention_dim; a.F = c.location_n_filters; a.K = c.location_kernel_size; a.R = c.window_range;
    a.n_mel = c.n_mel_channels; a.T = text_len; a.batch = batch; a.nbc = pad_batch(batch); a.step0 = step0; a.n_steps = n_steps;
    a.max_steps = max_steps; a.pd_rows = p.pd_rows;
    a.dbg = reinterpret_cast<u64*>(g_pd_debug);
    if (a.dbg) hipLaunchKernelGGL(taco_persistent_kernel<true>, dim3(PD_WG), dim3(PD_T), 0, s, a);
    else hipLaunchKernelGGL(taco_persistent_kernel<false>, dim3(PD_WG), dim3(PD_T), 0, s, a);
    CTTS_CHECK_LAUNCH("taco_persistent");
    return CTTS_OK;
}

/* Profiling aid: device buffer of 256 x 64 x 16 uint64 that receives s_memrealtime stamps at the phase boundaries of the
 * first 64 steps of every following persistent launch (NULL switches it off). */
int ctts_taco_decoder_persistent_debug(void* stamps) {
    g_pd_debug = stamps;
    return CTTS_OK;
}

}  // extern "C"
