// Persistent, WEIGHT-RESIDENT Tacotron2-TM decoder loop for gfx950: ONE launch runs a whole block of decoder steps
// (model.py:668-767 per step) on all 256 CUs, and every LSTM weight stays ON its compute unit for the whole launch.
//
// Why: a decoder step is a chain of small dependent mat-vecs (27 M fp32 weights = 108 MB, batch <= 4).  As six launches
// per step it costs 85 us/step.  The first persistent form (round 2-3) kept 21 % of the weights on the CU and streamed
// the other 85 MB from L2 / Infinity Cache every step: 39.6 us/step, with the all-gathers of the step queueing behind
// the workgroup's own weight stream in the CU's memory pipeline.  But the chip holds 256 x (512 KB registers + 160 KB
// LDS) = 168 MB on-CU - more than the weights.  This form puts them there:
//   * workgroups 0..251 ("LSTM workgroups", 256 threads = one wave per SIMD, so each wave owns the SIMD's whole
//     512-entry register file): every wave owns two or three whole LSTM units (all four gate rows, ALL columns) of the
//     workgroup's 5-6 attention-RNN units, 3-4 decoder-RNN units and 3-4 second-decoder-RNN units: 304-384 weights per
//     lane in VGPRs + AGPRs, the rest (up to 36 tiles of 1 KB per wave) in LDS; the products run on the matrix pipe
//     (v_mfma_f32_4x4x1_16b_f32: gate rows x batch items, 16 column blocks per instruction), the gate sums of a unit end
//     up in the lanes that update its cell - no barrier between product and cell.  Four workgroup classes (6/2/3,
//     5/4/2, 5/3/4, 5/3/3 units) spread the 1280 + 768 + 768 units over 252 workgroups, one decoder-RNN unit per wave at most.
//     Plus, as before: one row of the query projection (workgroups < attention_dim), one or two rows of the projection
//     row set [mel | gate | first prenet layer folded through the mel projection] and one or two rows of the second
//     prenet layer, LDS-resident.
//   * workgroups 252..255: the windowed location-sensitive attention of utterance b = workgroup - 252, nothing else.
// Per step NOTHING is streamed but the 33-row attention windows and the prenet's dropout bytes: the weight bytes are
// read once per launch (108 MB at entry).  The per-lane partial sums of every product are accumulated as soon as the
// product's input vector has arrived ("early" products of the next cell evaluation run right after a publish, inside
// the next exchange's latency) and reduced over the wave (DPP) only in the fresh phase.
// Exchanges: a vector produced by many workgroups and needed by all (att_h, q, ctx, dec_h, d2_h, h1, prenet) is
// all-gathered through granules written with ONE agent-scope (write-through) store each - q: 8 bytes {tag = step + 1, value};
// the X vectors since round 5: self-flagging 4-byte values (publish_x below) -
// and polled with agent-scope loads: the data is the flag, no fences, no grid barrier (MI355X_MICROARCH.md
// "handoff" / "allgather" rows; cdna_hip_programming.md Guideline 16 R2).  With no weight stream in the CU's memory
// pipeline the polls are the only traffic ("parked" column of the price list).  Every poll loop is bounded; on a
// timeout the workgroup records (code, workgroup, phase, step) in the control words and the whole grid drains.
// The granule area is filled with 0xFF by the host wrapper before EVERY launch; the control words are sticky.
#include "tacotron_plan.h"
#include "taco_math.h"
#include "tuning.h"

namespace ctts {
namespace {

using namespace taco;
using namespace tmath;

constexpr int PD_WG = 256;          // workgroups == CUs of an MI355X
constexpr int PD_LWG = 252;         // LSTM workgroups; the last PD_NB are the attention workgroups
constexpr int PD_T = 256;           // threads per workgroup: 4 waves, ONE per SIMD (512 registers per lane)
constexpr int PD_NW = PD_T / 64;
constexpr int PD_NB = 4;            // batch, padded
constexpr int PD_RA = 1280, PD_RD = 768, PD_P = 256, PD_DM = 512;
constexpr int PD_AMAX = 192, PD_TMAX = 1024, PD_W = 33, PD_FMAX = 32, PD_KMAX = 31;
// workgroup classes: units of (attention RNN, decoder RNN, second decoder RNN) per workgroup
//   class 0: wg   0..19  (6, 2, 3)     class 1: wg 20..51  (5, 4, 2)     class 2: wg 52..95  (5, 3, 4)     class 3: wg 96..251 (5, 3, 3)
// 20 * 6 + 232 * 5 = 1280;  20 * 2 + 32 * 4 + 200 * 3 = 768;  20 * 3 + 32 * 2 + 44 * 4 + 156 * 3 = 768.  Chosen so that NO
// wave holds more than one decoder-RNN unit or more than one second-decoder unit (the fresh phases C and D are then one
// cell per wave; three decoder cells on one wave delayed the dec_h exchange of the whole chip by ~1.5 us) and at most
// two attention-RNN units, and every class fits 92 register tiles per wave + 76 LDS tiles per workgroup.
constexpr int PD_C0 = 20, PD_C1 = 52, PD_C2 = 96;
// LDS vector store X of an LSTM workgroup: [item][n + 16] per vector (padded rows, see pd_xs)
constexpr int XP = 0, XCTX = XP + PD_NB * (PD_P + 16), XDEC = XCTX + PD_NB * (PD_DM + 16), XATT = XDEC + PD_NB * (PD_RD + 16),
              XD2 = XATT + PD_NB * (PD_RA + 16), XH1 = XD2 + PD_NB * (PD_RD + 16), X_FLOATS = XH1 + PD_NB * (PD_P + 16);
// LDS-resident single rows (plain row layout)
constexpr int WQ = X_FLOATS;                                   // query row                        [1280]
constexpr int WPR = WQ + PD_RA;                                // two projection rows              [2][1280]
constexpr int WW2 = WPR + 2 * (PD_RD + PD_DM);                 // two second-prenet rows           [2][256]
constexpr int WLT = WW2 + 2 * PD_P;                            // LSTM weight tiles kept in LDS: [tile][lane] float4, 76 tiles
constexpr int LSTM_FLOATS = WLT + 76 * 64 * 4;
constexpr int PD_DBG_SLOTS = 24;             // ctts_taco_decoder_persistent_debug: [PD_WG][64 steps][PD_DBG_SLOTS] stamps
#ifndef PD_LIGHT_SAMPLES_BIG
#define PD_LIGHT_SAMPLES_BIG 1
#endif
#ifndef PD_PB_MAX
#define PD_PB_MAX 6
#endif
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
    int dly[6];                 // low 16 bits: s_sleep(1) units (64 cycles) before the FIRST poll of att_h, ctx, dec_h, d2_h, h1, p; + 0x10000: no light phase (see gather_x)
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

// X vectors (att_h, ctx, dec_h, d2_h, h1, prenet: produced by many workgroups, needed by all 252) travel as SELF-FLAGGING 4-byte
// values (round 5): a float IS the granule, PD_SENT (a NaN pattern no arithmetic here produces) means "not yet".  Two buffers per
// vector by step parity; whoever publishes slot i of step s also puts PD_SENT back into slot i of the OTHER buffer, which holds
// step s - 1: every consumer has finished with it (each publish follows a gather that needed a later publish of every consumer),
// and it is polled again in step s + 1 - after the publisher's own later gathers, whose vmcnt waits cover the reset store, and at
// least one more exchange between the two workgroups.  Half the bytes of the {tag, value} form (which q keeps), and att_h is ONE
// round of five 16-byte loads per thread instead of two.  (Built once before the polls were timed: no gain then - r5_53 - because the
// early polls were the bound.)  The host fills the buffers with 0xFF before every launch.
constexpr unsigned PD_SENT = 0xFFFFFFFFu;
typedef __attribute__((address_space(1))) unsigned gu32w;
// A value can never BE the sentinel: PD_SENT is a valid quiet NaN (sign set, full payload), and AMD's NaN propagation keeps an
// input NaN's sign and payload - a 0xFF-filled or uninitialised weight / state buffer would publish exactly PD_SENT, every
// consumer would spin to its timeout and the launch would abort instead of producing NaN frames like the other forms.
__device__ __forceinline__ unsigned pd_bits(float v) {
    const unsigned b = __float_as_uint(v);
    return b == PD_SENT ? 0x7FC00000u : b;
}
// (The reset store of the other parity relies on a workgroup's stores to one address retiring in order - vmcnt, in-order
// return of same-type memory operations - and on the exchanges in between: see the paragraph above.)
__device__ __forceinline__ void publish_x(float* cur, float* oth, int idx, float v) {
    __hip_atomic_store((gu32w*)cur + idx, pd_bits(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store((gu32w*)oth + idx, PD_SENT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the two parity buffers of an X vector of N values per item inside its exchange region (offset in u64 words)
__device__ __forceinline__ float* pd_xbuf(u64* xb, unsigned off, int par, int N) { return reinterpret_cast<float*>(xb + off) + (size_t)par * PD_NB * N; }

// All-gather receive: thread t owns granules t, t + 256, ...; a granule is accepted when its tag equals `epoch`.
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
    __attribute__((aligned(16))) float memw[PD_W * PD_DM];   // the memory window (context operand), staged before the query arrives
    float loc[PD_W][PD_FMAX + 4];                 // rows 16-byte aligned: read as float4 broadcasts
    float wcat[2][PD_W + PD_KMAX - 1 + 1];
    float q[PD_AMAX];
    float en[64];
    __attribute__((aligned(16))) float wts[64];   // read as float4 by the context pass
    float w[PD_TMAX], cum[PD_TMAX];
    float ctx[PD_DM];
    float pos;
};
static_assert(PD_DM == 2 * PD_T, "the context pass maps one thread to two memory dimensions");
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
        if (s0 + W <= a.T) {
            // the 33 rows of a window are consecutive rows of the [T][A] / [T][512] arrays: two linear copies, no index math
            const float4* pmsrc = reinterpret_cast<const float4*>((a.ws + a.pm) + ((size_t)b * a.T + s0) * a.A);
            const float4* msrc = reinterpret_cast<const float4*>((a.ws + a.memory) + ((size_t)b * a.T + s0) * PD_DM);
            for (int i = t; i < W * a4; i += PD_T) reinterpret_cast<float4*>(s.pmw)[i] = pmsrc[i];
#pragma unroll 6
            for (int i = t; i < W * (PD_DM / 4); i += PD_T) reinterpret_cast<float4*>(s.memw)[i] = msrc[i];
        } else {
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
        }
        for (int i = t; i < 2 * (W + a.K - 1); i += PD_T) {
            const int c = i / (W + a.K - 1), j = i % (W + a.K - 1);
            const int pos = s0 - padk + j;
            const float* src = c == 0 ? s.w : s.cum;
            s.wcat[c][j] = (pos >= 0 && pos < a.T) ? src[pos] : 0.f;
        }
    }
    __syncthreads();
    {   // location conv (model.py:56-60): thread = (filter f, position group g of 8); taps out of registers
        const int f = t & 31, g = t >> 5;
        for (int tt = g; tt < W; tt += PD_T / 32) {
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
        // three positions per pass with independent chains (one position per pass was bound by the LDS round trips: read
        // the features, 32 dependent-by-four FMAs, read-modify-write the pre-activation)
#pragma unroll 1
        for (int tt0 = 0; tt0 < W; tt0 += 3) {
            float acc[3][4];
#pragma unroll
            for (int u = 0; u < 3; ++u)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[u][k] = 0.f;
#pragma unroll
            for (int f4 = 0; f4 < PD_FMAX / 4; ++f4) {
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const float4 l = *reinterpret_cast<const float4*>(&s.loc[min(tt0 + u, W - 1)][4 * f4]);   // wave-uniform address: broadcast
                    acc[u][0] = fmaf(r.wd[4 * f4 + 0], l.x, acc[u][0]); acc[u][1] = fmaf(r.wd[4 * f4 + 1], l.y, acc[u][1]);
                    acc[u][2] = fmaf(r.wd[4 * f4 + 2], l.z, acc[u][2]); acc[u][3] = fmaf(r.wd[4 * f4 + 3], l.w, acc[u][3]);
                }
            }
#pragma unroll
            for (int u = 0; u < 3; ++u)
                if (tt0 + u < W) s.pmw[(tt0 + u) * a.A + r.ad] += (acc[u][0] + acc[u][2]) + (acc[u][1] + acc[u][3]);
        }
    }
    return s0;      // (the caller's q gather ends with a workgroup barrier: pmw / memw are visible after it)
}

// Part 2, on the critical path between the query and the context: tanh + v-weighted sum over the attention dims,
// softmax, context, publish.
template <bool DBG>
__device__ __forceinline__ void pd_attention_post(const PdArgs& a, AttnLds& s, const AttnRegs& r, int b, int s0, unsigned epoch,
                                                  int step, float* g_ctx) {
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int wg = blockIdx.x;
    const int W = 2 * a.R + 1;
    const int len = reinterpret_cast<const int*>(a.ws + a.lengths)[b];
    // energies (model.py:107-112): wave wv takes window positions wv, wv + 4, ...; a lane sums its attention dims (lane, lane + 64,
    // lane + 128) first, so a position costs ONE 64-lane reduction, not three.  NJ = 2 when attention_dim <= 128 (the repo default:
    // the third dim of every lane is masked, and its nine exp2 / rcp chains per lane were a third of this phase - round 5); the masked
    // terms were exact zeros, so the sums are bit-identical.
    auto energies = [&](auto njc) {
        constexpr int NJ = decltype(njc)::value;
        constexpr int NE = (PD_W + PD_NW - 1) / PD_NW;
        float qv[NJ], ev[NE];
#pragma unroll
        for (int j = 0; j < NJ; ++j) qv[j] = lane + 64 * j < a.A ? s.q[lane + 64 * j] : 0.f;
        // staged by hand: all loads, then all exp2, then all rcp, then the sums.  Written as one loop per position the
        // wave executed 27 dependent exp2 -> add -> rcp -> fma chains back to back (2.1 us: a wave issues in order and the
        // transcendental pipe has ~40 cycles of latency); the order of every sum is unchanged
        // (tanhf here: arbiter distance of band 1 1.6e-4 -> 1.1e-4, step 32 -> 36 us with libm in the cells as well: not taken)
        float x[NE][NJ];
#pragma unroll
        for (int i = 0; i < NE; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)     // pmw = processed memory + location term (pd_attention_pre)
                x[i][j] = lane + 64 * j < a.A ? s.pmw[min(wv + PD_NW * i, W - 1) * a.A + lane + 64 * j] + qv[j] : 0.f;
#pragma unroll
        for (int i = 0; i < NE; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) x[i][j] = __builtin_amdgcn_exp2f(x[i][j] * 2.8853900817779268f);
#pragma unroll
        for (int i = 0; i < NE; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) x[i][j] = __builtin_amdgcn_rcpf(1.0f + x[i][j]);
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            float e = 0.f;
#pragma unroll
            for (int j = 0; j < NJ; ++j) e += lane + 64 * j < a.A ? r.va3[j] * (1.0f - 2.0f * x[i][j]) : 0.f;
            ev[i] = e;
        }
        PD_STAMP(18);
        wave_totals<NE>(ev);
        PD_STAMP(19);
#pragma unroll
        for (int i = 0; i < NE; ++i)
            if (lane == 0 && wv + PD_NW * i < W) s.en[wv + PD_NW * i] = ev[i];
    };
    if (a.A > 128) energies(std::integral_constant<int, 3>{});
    else energies(std::integral_constant<int, 2>{});
    __syncthreads();
    PD_STAMP(3);
    if (wv == 0) {
        const int pos_l = s0 + lane;
        float e = -INFINITY;
        if (lane < W && pos_l < len && pos_l < a.T) e = s.en[lane];
        const float m = wave_max(e);
        const float pexp = lane < W ? expf(e - m) : 0.f;          // masked lanes: exp(-inf) = 0
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
    {   // context = sum_t w[t] * memory[t] out of the staged window: dimensions 2 t and 2 t + 1 per thread, two chains each (the
        // same two-chain order per dimension as the one-dimension-per-thread form).  Round 5: adjacent dimensions (one 8-byte
        // LDS read per window row instead of two 4-byte ones) and the 33 weights as nine 16-byte reads instead of 33 broadcasts:
        // 42 LDS instructions per thread instead of 99 on the critical path between q and ctx.
        float wreg[36];
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const float4 w4 = *reinterpret_cast<const float4*>(&s.wts[4 * i]);
            wreg[4 * i] = w4.x; wreg[4 * i + 1] = w4.y; wreg[4 * i + 2] = w4.z; wreg[4 * i + 3] = w4.w;
        }
        float c0[2] = {0.f, 0.f}, c1[2] = {0.f, 0.f};
        const float* mw = s.memw + 2 * t;
#pragma unroll
        for (int tt = 0; tt + 1 < PD_W; tt += 2) {
            const float w0 = tt < W ? wreg[tt] : 0.f, w1 = tt + 1 < W ? wreg[tt + 1] : 0.f;
            const float2 m0 = *reinterpret_cast<const float2*>(mw + tt * PD_DM), m1 = *reinterpret_cast<const float2*>(mw + (tt + 1) * PD_DM);
            c0[0] = fmaf(w0, m0.x, c0[0]); c0[1] = fmaf(w0, m0.y, c0[1]);
            c1[0] = fmaf(w1, m1.x, c1[0]); c1[1] = fmaf(w1, m1.y, c1[1]);
        }
        const float wl = PD_W - 1 < W ? wreg[PD_W - 1] : 0.f;
        const float2 ml = *reinterpret_cast<const float2*>(mw + (PD_W - 1) * PD_DM);
        c0[0] = fmaf(wl, ml.x, c0[0]); c0[1] = fmaf(wl, ml.y, c0[1]);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int d = 2 * t + k;
            const float c = c0[k] + c1[k];
            s.ctx[d] = c;
            __hip_atomic_store((gu32w*)g_ctx + (b * PD_DM + d), pd_bits(c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
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
        r.pg = wv < 3 ? 0 : -1;          // waves 0..2: attention dims 0..191, every window position
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
        // ctx of step - 1 (other parity) back to "not yet": q(step) has arrived, so every LSTM workgroup is past its step - 1.  This
        // workgroup publishes nothing but ctx, so the reset goes out HERE, a whole attention evaluation ahead of the data stores
        // of this step that a consumer waits for before it can come back to this buffer.
        for (int d = t; d < PD_DM; d += PD_T)
            __hip_atomic_store((gu32w*)pd_xbuf(a.xb, a.g_ctx, par ^ 1, PD_DM) + (b * PD_DM + d), PD_SENT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (real) {
            pd_attention_post<DBG>(a, att, r, b, s0, epoch, step, pd_xbuf(a.xb, a.g_ctx, par, PD_DM));
        } else {
            for (int d = t; d < PD_DM; d += PD_T)
                __hip_atomic_store((gu32w*)pd_xbuf(a.xb, a.g_ctx, par, PD_DM) + (b * PD_DM + d), __float_as_uint(0.f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        PD_STAMP(6);
    }
    if (real) {
        for (int p = t; p < a.T; p += PD_T) { (a.ws + a.w)[(size_t)b * a.T + p] = att.w[p]; (a.ws + a.cum)[(size_t)b * a.T + p] = att.cum[p]; }
        if (t == 0) (a.ws + a.pos)[b] = att.pos;
        for (int d = t; d < PD_DM; d += PD_T) (a.ws + a.ctx)[b * PD_DM + d] = att.ctx[d];
    }
}


// ---- the weight-resident LSTM wave -----------------------------------------------------------------------------------
// A wave owns whole LSTM units ("slots": all four gate rows i, f, g~, o of one unit of one cell) and evaluates them on the
// matrix pipe with v_mfma_f32_4x4x1_16b_f32 (16 independent 4x4x1 blocks; lane l = 4 blk + sub; measured operand layout,
// scripts/micro/mfma_4x4x1_layout.hip: A[blk][row = sub], B[blk][col = sub], D[vgpr = row][lane = 4 blk + col]):
//     A[blk][i] = W[gate i of the unit][column k],   B[blk][j] = x[item j][column k],   k = 64 tile + 4 blk + e  (e = 0..3)
// so one "tile" = one float4 of weights per lane (64 columns of the unit's four rows) + one 16-byte LDS read of x per lane
// feeds four MFMAs, and D[i] in lane (blk, j) accumulates gate i / item j over the columns of block blk.  After the last
// product the 16 blocks are summed across lanes; every lane with sub = j then holds the four pre-activations of item j:
// the cell update runs right there - no LDS round trip, no workgroup barrier between the gate sums and the cell.
// The weights are ordinary values that only MFMAs consume: the register allocator spreads them over VGPRs AND AGPRs (512
// per lane at one wave per SIMD; AGPR-held ones are copied by v_accvgpr_read on the otherwise idle VALU).
typedef float pd_f4 __attribute__((ext_vector_type(4)));


template <int I> struct IC { static constexpr int value = I; };
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) { f(IC<B>{}); static_for<B + 1, E>(f); }
}

// Units of workgroup wg (class table above) and its first unit per cell
__device__ __forceinline__ int pd_first_att(int wg) { return wg < PD_C0 ? 6 * wg : 6 * PD_C0 + 5 * (wg - PD_C0); }
__device__ __forceinline__ int pd_first_dec(int wg) {
    return wg < PD_C0 ? 2 * wg : wg < PD_C1 ? 2 * PD_C0 + 4 * (wg - PD_C0) : 2 * PD_C0 + 4 * (PD_C1 - PD_C0) + 3 * (wg - PD_C1);
}
__device__ __forceinline__ int pd_first_d2(int wg) {
    return wg < PD_C0   ? 3 * wg
           : wg < PD_C1 ? 3 * PD_C0 + 2 * (wg - PD_C0)
           : wg < PD_C2 ? 3 * PD_C0 + 2 * (PD_C1 - PD_C0) + 4 * (wg - PD_C1)
                        : 3 * PD_C0 + 2 * (PD_C1 - PD_C0) + 4 * (PD_C2 - PD_C1) + 3 * (wg - PD_C2);
}

// cell types of a slot.  Tiles of a slot, fresh segment first:
//   ATT (44): prenet 4 [fresh] | context 8 | decoder hidden 12 | own hidden (W_hh) 20
//   DEC (40): context 8 [fresh] | attention hidden 20 | own hidden 12
//   D2  (24): decoder hidden 12 [fresh] | own hidden 12
constexpr int CELL_NONE = 0, CELL_ATT = 1, CELL_DEC = 2, CELL_D2 = 3;
__host__ __device__ constexpr int pd_cell_tiles(int c) { return c == CELL_ATT ? 44 : c == CELL_DEC ? 40 : c == CELL_D2 ? 24 : 0; }

// A wave signature: up to three slots (cell types) and the number of its trailing tiles that live in LDS instead of
// registers.  The trailing tiles are always "early" (off the critical path) tiles of the last slot.
template <int C0, int C1, int C2, int NL>
struct PdSig {
    static constexpr int cell[3] = {C0, C1, C2};
    static constexpr int off[4] = {0, pd_cell_tiles(C0), pd_cell_tiles(C0) + pd_cell_tiles(C1),
                                   pd_cell_tiles(C0) + pd_cell_tiles(C1) + pd_cell_tiles(C2)};
    static constexpr int NT = off[3], NLDS = NL, NREG = NT - NL;
    static_assert(NREG <= 96, "more than 384 weight registers per lane");
};
using SigSDA = PdSig<CELL_D2, CELL_DEC, CELL_ATT, 16>;     // 108 tiles, 92 in registers (the common wave: 3 of 4 in every class)
using SigSAA = PdSig<CELL_D2, CELL_ATT, CELL_ATT, 20>;     // 112 - 20 = 92
using SigDAA = PdSig<CELL_DEC, CELL_ATT, CELL_ATT, 36>;    // 128 - 36 = 92
using SigDA = PdSig<CELL_DEC, CELL_ATT, CELL_NONE, 0>;     // 84
using SigAA = PdSig<CELL_ATT, CELL_ATT, CELL_NONE, 0>;     // 88
constexpr int PD_LT_WAVE = 36;                             // most LDS tiles of one wave; LDS tile slots per workgroup:
constexpr int PD_LT = 76;                                  // 16+16+16+28 (class 0), 16+16+12+28, 16+16+12+16, 16+16+12+0

// padded X rows: item j of a vector of n floats starts at j (n + 16): the 16-byte reads of the four items of a block fall
// into different banks
__host__ __device__ constexpr int pd_xs(int n) { return n + 16; }

struct PdSlotRt {          // runtime side of a slot
    int unit;              // unit index inside its cell (row of gate i: i * H + unit)
    pd_f4 acc;             // D accumulators: gate i of item (lane & 3), block (lane >> 2).  ONE chain per slot: a second
                           // interleaved chain was measured (A 1.06 -> 0.98 us) and cost more in spilled weights than it won
    float bias[4];
    float c, h;            // cell state / hidden value of item (lane & 3)
};

// all-gather receive into a padded X vector (item stride n + 16): four values per 16-byte agent-scope load, each its own flag
// (see publish_x).  Every sweep of a vector by all 256 CUs is 256 x 4 N x 4 bytes of fabric traffic that the publishers' stores and
// every other CU's sweep queue behind, so NOTHING is polled before the vector can be complete: the first poll goes out `dly` x 64
// cycles after wave 0's own publish (round 5).  A vector cannot be complete before its slowest publisher is done - for att_h
// 1.6-2.0 us after the first ones with early polls on the fabric, 0.5 us without (profiles/r5_53, r5_59) - and a poll that cannot
// succeed yet is traffic that publisher's write-through stores queue behind: holding the first polls back took the STEP from 28.4
// to 24.0 us (r5_59); then straight to the FULL sweep (dly + 0x10000, the default) instead of a light phase first, 22.4 (r5_62);
// this 4-byte format - att_h in ONE round of five loads per thread instead of two - 21.2 (r5_64).  The sweep re-reads what was
// still missing.
// The loads are raw buffer loads with the sc1 bit (what an agent-scope atomic load lowers to), so that the compiler
// tracks them itself (inline-asm loads cost ~125 spilled weight registers around every gather).
typedef unsigned pd_u4 __attribute__((ext_vector_type(4)));
constexpr int PD_AUX_SC1 = 16;
template <int N>
__device__ __forceinline__ bool gather_x(const float* g, float* dst, unsigned* ctl, int t, unsigned phase, unsigned step, int dly) {
    constexpr int QUADS = PD_NB * N / 4, PPT = (QUADS + PD_T - 1) / PD_T;
    static_assert(QUADS >= PD_T && N % 4 == 0 && PPT <= 5, "every thread owns one to five quads, a quad lies inside one item");
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g), 0, PD_NB * N * 4, 0x00020000);
    auto timed_out = [&](unsigned spins) -> bool {
        if ((spins & 255u) != 255u) return false;
        if (__hip_atomic_load((gu32*)ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return true;
        if (spins > PD_SPIN_LIMIT) {
            if ((t & 63) == 0 && atomicCAS(ctl, 0u, 1u) == 0u) { ctl[1] = blockIdx.x; ctl[2] = phase; ctl[3] = step; }
            return true;
        }
        return false;
    };
    auto landed = [](const pd_u4& v) { return v[0] != PD_SENT && v[1] != PD_SENT && v[2] != PD_SENT && v[3] != PD_SENT; };
    {
        // ONE wave holds the first poll back (and, without the 0x10000 flag, then watches 64 quads spread over item 0 - the light
        // phase of round 4, kept for A/B) while the other three wait at the barrier: the sweep of ALL waves starts `dly` after wave
        // 0's publish, which absorbs the lateness of a wave with two cells (every wave waiting for itself: 22.4 -> 25.7 us/step)
        if (t < 64) {
            for (int i = 0; i < (dly & 0xffff); ++i) __builtin_amdgcn_s_sleep(1);
            if (!(dly & 0x10000)) {
                const int ql = t * (N / 4 / 64);                            // quads of item 0, spread over its units
                for (unsigned spins = 0;; ++spins) {
                    const pd_u4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, 16 * ql, 0, PD_AUX_SC1);
                    if (__all(landed(v))) break;
                    if (timed_out(spins)) break;          // recorded in the control words: the sweep below gives up on them
                    __builtin_amdgcn_s_sleep(1);
                }
            }
        }
        __syncthreads();
    }
    unsigned done = 0;
    for (unsigned spins = 0;; ++spins) {            // full sweep
        pd_u4 v[PPT];
#pragma unroll
        for (int k = 0; k < PPT; ++k)
            if (t + PD_T * k < QUADS && !((done >> k) & 1u))
                v[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, 16 * (t + PD_T * k), 0, PD_AUX_SC1);
            else v[k] = pd_u4{0u, 0u, 0u, 0u};
        bool ok = true;
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int i = 4 * (t + PD_T * k);
            if (t + PD_T * k < QUADS && !((done >> k) & 1u)) {
                if (landed(v[k])) {
                    *reinterpret_cast<pd_u4*>(dst + i + 16 * (i / N)) = v[k];
                    done |= 1u << k;
                } else {
                    ok = false;
                }
            }
        }
        if (__all(ok)) break;
        if (timed_out(spins)) return false;
        __builtin_amdgcn_s_sleep(1);
    }
    return true;
}

// sum of the 16 blocks: every lane ends with the total of its item (lane & 3).  Within a row of 16 lanes on the DPP
// network (row_ror:4, row_ror:8); across the four rows with gfx950's v_permlane16_swap / v_permlane32_swap (round 5): with both
// operands = v the swap leaves {even row, even row} in one and {odd row, odd row} in the other, so their sum is v + v[lane ^ 16]
// (resp. ^ 32) in every lane - the same two additions as the ds_bpermute form it replaces (__shfl_xor: two LDS-crossbar round
// trips per sum, in the tail of every phase), bit for bit (scripts/micro/permlane_swap_sum.hip).
__device__ __forceinline__ float pd_block_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false));     // row_ror:4
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));     // row_ror:8
    const auto r16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r16[0]) + __uint_as_float(r16[1]);
    const auto r32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r32[0]) + __uint_as_float(r32[1]);
}

#define PD_MFMA(acc, wv, xv) (acc) = __builtin_amdgcn_mfma_f32_4x4x1f32((wv), (xv), (acc), 0, 0, 0)

template <class SIG>
struct PdWaveW {
    // register tiles 0 .. NREG-1 as single floats (VGPRs and AGPRs, the allocator's choice; 128-bit tuples instead made it
    // spill 40 tiles per wave to scratch: an aligned AGPR quadruple per tile is much harder to place than four singles)
    float r[SIG::NREG > 0 ? 4 * SIG::NREG : 4];
};

// tile T of this wave
template <class SIG, int T>
__device__ __forceinline__ pd_f4 pd_tile(const PdWaveW<SIG>& w, const float* lt, int lane) {
    if constexpr (T < SIG::NREG) return pd_f4{w.r[4 * T], w.r[4 * T + 1], w.r[4 * T + 2], w.r[4 * T + 3]};
    else return *reinterpret_cast<const pd_f4*>(lt + ((T - SIG::NREG) * 64 + lane) * 4);
}

// One segment [T0, T0 + NTILE) of slot tiles against NTILE consecutive 64-column chunks of an X vector starting at chunk X0:
// for every slot s of the signature whose cell is CELL and whose segment starts at tile SEG inside the slot.  Slots of
// the same cell share the x read and interleave their MFMAs (independent accumulators).
template <class SIG, int CELL, int SEG, int NTILE>
__device__ __forceinline__ void pd_segment(const PdWaveW<SIG>& w, const float* lt, const float* xlane, int lane, PdSlotRt (&sl)[3]) {
    static_for<0, NTILE>([&](auto tt) {
        constexpr int tau = decltype(tt)::value;
        const pd_f4 x = *reinterpret_cast<const pd_f4*>(xlane + 64 * tau);
        static_for<0, 3>([&](auto ss) {
            constexpr int s = decltype(ss)::value;
            if constexpr (SIG::cell[s] == CELL) {
                const pd_f4 wv = pd_tile<SIG, SIG::off[s] + SEG + tau>(w, lt, lane);
                PD_MFMA(sl[s].acc, wv[0], x[0]); PD_MFMA(sl[s].acc, wv[1], x[1]);
                PD_MFMA(sl[s].acc, wv[2], x[2]); PD_MFMA(sl[s].acc, wv[3], x[3]);
            }
        });
        // a fence for the instruction scheduler every second tile: without it the x reads (and the AGPR copies) of a whole
        // segment are hoisted in front of its first MFMA and the weights end up in scratch
        if constexpr (tau % 2 == 1) __builtin_amdgcn_sched_barrier(0);
    });
}

// LSTM cell update of every slot of cell CELL (layers.py:308-372, gate order i, f, g, o): block sums, activations,
// publish h' of item (lane & 3) at [item][first + unit] from the lanes of block 0.
template <class SIG, int CELL>
__device__ __forceinline__ void pd_cells(PdSlotRt (&sl)[3], float* g, float* goth, int H, int first, int lane) {
    static_for<0, 3>([&](auto ss) {
        constexpr int s = decltype(ss)::value;
        if constexpr (SIG::cell[s] == CELL) {
            float pre[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) pre[i] = pd_block_sum(sl[s].acc[i]) + sl[s].bias[i];
            // ~2-ulp forms (acc_*): the 1e-7 ABSOLUTE error of the plain exp2 / rcp forms is what the sharp-attention
            // trajectory amplifies most (tests/test_tacotron_long.py, arbiter)
            const float gi = acc_sigmoid(pre[0]), gf = acc_sigmoid(pre[1]);
            const float gg = acc_tanh(pre[2]), go = acc_sigmoid(pre[3]);
            sl[s].c = gf * sl[s].c + gi * gg;
            sl[s].h = go * acc_tanh(sl[s].c);
            if (lane < PD_NB) publish_x(g, goth, lane * H + first + sl[s].unit, sl[s].h);
            sl[s].acc = pd_f4{0.f, 0.f, 0.f, 0.f};
        }
    });
}

// one LDS-resident single row (query / projection / second prenet layer) on the matrix pipe: A = the row in sub-lane 0,
// zero in the three other rows of the block.  wrow: [64 m4 + 4 blk + e], xlane as above.  Accumulates into acc[0].
template <int NTILE>
__device__ __forceinline__ void pd_row(const float* wrow, const float* xlane, int lane, pd_f4& acc) {
    const bool row0 = (lane & 3) == 0;
    pd_f4 c1 = {0.f, 0.f, 0.f, 0.f};                              // two interleaved chains: a single row IS the critical path
#pragma unroll
    for (int tau = 0; tau < NTILE; ++tau) {
        pd_f4 wv = *reinterpret_cast<const pd_f4*>(wrow + 64 * tau + 4 * (lane >> 2));
        const pd_f4 x = *reinterpret_cast<const pd_f4*>(xlane + 64 * tau);
        if (!row0) wv = pd_f4{0.f, 0.f, 0.f, 0.f};
        PD_MFMA(acc, wv[0], x[0]); PD_MFMA(c1, wv[1], x[1]); PD_MFMA(acc, wv[2], x[2]); PD_MFMA(c1, wv[3], x[3]);
    }
    acc[0] += c1[0];
}

struct LstmLds {
    float pown[2][PD_NB];                         // this workgroup's one or two prenet outputs (state for the next launch)
};

// runtime description of a wave's slots: unit index inside its cell (or -1)
struct PdSlots { int unit[3]; };

template <class SIG, bool DBG>
__device__ __forceinline__ void lstm_wave(const PdArgs& a, float* L, LstmLds& S, int wg, const PdSlots& su, int lt_tile0) {
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int blk = lane >> 2, sub = lane & 3;
    float* X = L;
    const float* lt = L + WLT + lt_tile0 * 64 * 4;                      // this wave's LDS tiles
    constexpr int I_att = PD_P + PD_DM + PD_RD, I_dec = PD_RA + PD_DM, Dp = PD_RD + PD_DM;
    const int FA = pd_first_att(wg), FD = pd_first_dec(wg), F2 = pd_first_d2(wg);
    unsigned* ctl = reinterpret_cast<unsigned*>(a.xb + a.ctl);
    // this lane's read position in every X vector: item sub, columns 4 blk ..
    const float* xP = X + XP + sub * pd_xs(PD_P) + 4 * blk;
    const float* xC = X + XCTX + sub * pd_xs(PD_DM) + 4 * blk;
    const float* xD = X + XDEC + sub * pd_xs(PD_RD) + 4 * blk;
    const float* xA = X + XATT + sub * pd_xs(PD_RA) + 4 * blk;
    const float* x2 = X + XD2 + sub * pd_xs(PD_RD) + 4 * blk;
    const float* xH = X + XH1 + sub * pd_xs(PD_P) + 4 * blk;

    // ---- entry: every weight of this wave's units, once per launch.  Tile tau of a slot = float4 at
    // blob[mat + (gate * H + first + unit) * ld + col(tau) + 4 blk], gate = sub.
    PdWaveW<SIG> w;
    PdSlotRt sl[3];
    const int nbc = a.nbc;
    static_for<0, 3>([&](auto ss) {
        constexpr int s = decltype(ss)::value;
        constexpr int cell = SIG::cell[s];
        sl[s].unit = su.unit[s];
        sl[s].acc = pd_f4{0.f, 0.f, 0.f, 0.f};
        sl[s].c = 0.f; sl[s].h = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) sl[s].bias[i] = 0.f;
        if constexpr (cell != CELL_NONE) {
            constexpr int H = cell == CELL_ATT ? PD_RA : PD_RD;
            const int first = cell == CELL_ATT ? FA : cell == CELL_DEC ? FD : F2;
            const int urow = first + su.unit[s];
            const unsigned wih = cell == CELL_ATT ? a.att_wih : cell == CELL_DEC ? a.dec_wih : a.d2_wih;
            const unsigned whh = cell == CELL_ATT ? a.att_whh : cell == CELL_DEC ? a.dec_whh : a.d2_whh;
            const unsigned bih = cell == CELL_ATT ? a.att_bih : cell == CELL_DEC ? a.dec_bih : a.d2_bih;
            const unsigned bhh = cell == CELL_ATT ? a.att_bhh : cell == CELL_DEC ? a.dec_bhh : a.d2_bhh;
            constexpr int ldi = cell == CELL_ATT ? I_att : cell == CELL_DEC ? I_dec : PD_RD;
            const float* rih = a.blob + wih + (size_t)(sub * H + urow) * ldi + 4 * blk;
            const float* rhh = a.blob + whh + (size_t)(sub * H + urow) * H + 4 * blk;
            // column base of tile tau in [W_ih | W_hh] per cell (see the table above)
            static_for<0, pd_cell_tiles(cell)>([&](auto tt) {
                constexpr int tau = decltype(tt)::value;
                constexpr int T = SIG::off[s] + tau;
                const float* src;
                if constexpr (cell == CELL_ATT) {
                    if constexpr (tau < 4) src = rih + 64 * tau;                                   // prenet columns
                    else if constexpr (tau < 12) src = rih + PD_P + 64 * (tau - 4);                // context
                    else if constexpr (tau < 24) src = rih + PD_P + PD_DM + 64 * (tau - 12);       // decoder hidden
                    else src = rhh + 64 * (tau - 24);
                } else if constexpr (cell == CELL_DEC) {
                    if constexpr (tau < 8) src = rih + PD_RA + 64 * tau;                           // context
                    else if constexpr (tau < 28) src = rih + 64 * (tau - 8);                       // attention hidden
                    else src = rhh + 64 * (tau - 28);
                } else {
                    if constexpr (tau < 12) src = rih + 64 * tau;                                  // decoder hidden
                    else src = rhh + 64 * (tau - 12);
                }
                if constexpr (T < SIG::NREG) {
                    // four SCALAR loads (volatile: the vectoriser must not merge them): a value that is born as a quarter
                    // of a 128-bit load stays a sub-register of the tuple and the allocator then spills whole tuples; a
                    // value pinned by inline asm ("=a" / "=v") cannot move between the register files.  Plain scalar
                    // values that only MFMAs consume are spread over VGPRs and AGPRs by the allocator itself.
#pragma unroll
                    for (int e = 0; e < 4; ++e) w.r[4 * T + e] = *reinterpret_cast<const volatile float*>(src + e);
                } else {
                    *reinterpret_cast<pd_f4*>(L + WLT + ((lt_tile0 + T - SIG::NREG) * 64 + lane) * 4) = *reinterpret_cast<const pd_f4*>(src);
                }
                if constexpr (tau % 8 == 7) __builtin_amdgcn_sched_barrier(0);      // loads in batches of eight
            });
#pragma unroll
            for (int i = 0; i < 4; ++i)
                sl[s].bias[i] = __int_as_float(__builtin_amdgcn_readfirstlane(
                    __float_as_int((a.blob + bih)[i * H + urow] + (a.blob + bhh)[i * H + urow])));
            // state of step0: cell state and own hidden value of item sub
            const unsigned cst = cell == CELL_ATT ? a.att_c : cell == CELL_DEC ? a.dec_c : a.d2_c;
            const unsigned hin = cell == CELL_ATT ? a.att_h_in : cell == CELL_DEC ? a.dec_h_in : a.d2_h_in;
            if (sub < nbc) { sl[s].c = (a.ws + cst)[sub * H + urow]; sl[s].h = (a.ws + hin)[sub * H + urow]; }
        }
    });
    // the LDS-resident single rows (plain row layout)
    const int q_row = wg < a.A ? wg : -1;
    const int pr_row0 = wg, pr_row1 = wg + PD_LWG < a.pd_rows ? wg + PD_LWG : -1;
    const int w2_row0 = wg, w2_row1 = wg + PD_LWG < PD_P ? wg + PD_LWG : -1;
    for (int i = t; i < PD_RA / 4; i += PD_T) {
        if (q_row >= 0) reinterpret_cast<float4*>(L + WQ)[i] = reinterpret_cast<const float4*>((a.blob + a.Wq) + (size_t)q_row * PD_RA)[i];
        reinterpret_cast<float4*>(L + WPR)[i] = reinterpret_cast<const float4*>((a.blob + a.Wproj) + (size_t)pr_row0 * Dp)[i];
        if (pr_row1 >= 0)
            reinterpret_cast<float4*>(L + WPR + Dp)[i] = reinterpret_cast<const float4*>((a.blob + a.Wproj) + (size_t)pr_row1 * Dp)[i];
    }
    if (t < PD_P / 4) {
        reinterpret_cast<float4*>(L + WW2)[t] = reinterpret_cast<const float4*>((a.blob + a.W2) + (size_t)w2_row0 * PD_P)[t];
        if (w2_row1 >= 0) reinterpret_cast<float4*>(L + WW2 + PD_P)[t] = reinterpret_cast<const float4*>((a.blob + a.W2) + (size_t)w2_row1 * PD_P)[t];
    }
    const float bpr = wave == 1 ? (a.blob + a.bproj)[pr_row0] : (wave == 2 && pr_row1 >= 0) ? (a.blob + a.bproj)[pr_row1] : 0.f;

    // ---- entry: the vectors of step0 from the workspace (written by the init / the previous launch).  The workspace
    // holds nbc <= 4 batch rows per state array; the rows above stay zero here and are never written back.
    auto load_x = [&](int xo, int n, unsigned src) {
        for (int i = t; i < PD_NB * n; i += PD_T) X[xo + i + 16 * (i / n)] = i / n < nbc ? (a.ws + src)[i] : 0.f;
    };
    load_x(XP, PD_P, a.prenet); load_x(XCTX, PD_DM, a.ctx); load_x(XDEC, PD_RD, a.dec_h_in); load_x(XD2, PD_RD, a.d2_h_in);
    load_x(XATT, PD_RA, a.att_h_in);
    if (t < 2 * PD_NB) {
        const int k = t / PD_NB, b = t % PD_NB;
        const int row = k == 0 ? w2_row0 : w2_row1;
        S.pown[k][b] = (row >= 0 && b < nbc) ? (a.ws + a.prenet)[b * PD_P + row] : 0.f;
    }
    __syncthreads();
    // the early products of step0, in the order the loop accumulates them
    pd_segment<SIG, CELL_ATT, 24, 20>(w, lt, xA, lane, sl);
    pd_segment<SIG, CELL_ATT, 4, 8>(w, lt, xC, lane, sl);
    pd_segment<SIG, CELL_ATT, 12, 12>(w, lt, xD, lane, sl);
    pd_segment<SIG, CELL_DEC, 28, 12>(w, lt, xD, lane, sl);
    pd_segment<SIG, CELL_D2, 12, 12>(w, lt, x2, lane, sl);
    // early part of the projection rows: W[:, :768] . dec_h + W[:, 768:] . ctx (the d2_h part waits for the d2_h exchange)
    pd_f4 accP = {0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    bool fail = false;
#define PD_GATHER(NPT, N, buf, dst, phase)                                                                          \
    do {                                                                                                               \
        const bool ok_ = gather_x<N>(pd_xbuf(a.xb, (buf), par, (N)), (dst), ctl, t, (phase), (unsigned)step, a.dly[(phase) == 1u ? 0 : (phase) - 2u]); \
        if (__syncthreads_or(ok_ ? 0 : 1)) { fail = true; }                                                            \
    } while (0)

    for (int step = a.step0; step < a.step0 + a.n_steps && !fail; ++step) {
        const unsigned epoch = (unsigned)step + 1u;
        const int par = step & 1;
        PD_STAMP(0);
        // the prenet's dropout keep-bytes of step + 1 for this wave's row (waves 1, 2: first layer, rows of the folded
        // projection; waves 3, 0: second layer), lane = batch item.  Loaded HERE, a whole step ahead of their use: they
        // are first-touch HBM bytes.
        const bool have_next = step + 1 < a.max_steps;
        unsigned char keep_next = 0;
        if (have_next && lane < a.batch) {
            if (wave == 1 || (wave == 2 && pr_row1 >= 0)) {
                const int j = (wave == 1 ? pr_row0 : pr_row1) - a.n_mel - 1;
                if (j >= 0) keep_next = a.keep[(((size_t)(step + 1) * 2 + 0) * a.batch + lane) * PD_P + j];
            } else if (wave == 3 || (wave == 0 && w2_row1 >= 0)) {
                keep_next = a.keep[(((size_t)(step + 1) * 2 + 1) * a.batch + lane) * PD_P + (wave == 3 ? w2_row0 : w2_row1)];
            }
        }
        // ---- A: attention RNN on the fresh prenet columns (model.py:707-717)
        pd_segment<SIG, CELL_ATT, 0, 4>(w, lt, xP, lane, sl);
        pd_cells<SIG, CELL_ATT>(sl, pd_xbuf(a.xb, a.g_atth, par, PD_RA), pd_xbuf(a.xb, a.g_atth, par ^ 1, PD_RA), PD_RA, FA, lane);
        PD_STAMP(1);
        PD_STAMP_LANE0(20 + wave);                                               // (debug build: when did EACH wave publish its att_h units)
        PD_GATHER(20, PD_RA, a.g_atth, X + XATT, 1u);
        PD_STAMP(2);
        if (fail) break;
        // ---- B: query row (model.py:126 query_layer); then, while the attention workgroups work, the products on
        // att_h(step): the decoder RNN's (needed in C) and the attention RNN's recurrent ones (step + 1)
        if (wave == 0 && q_row >= 0) {
            pd_f4 q = {0.f, 0.f, 0.f, 0.f};
            pd_row<20>(L + WQ, xA, lane, q);
            const float qv = pd_block_sum(q[0]);
            if (lane < PD_NB) publish((a.xb + a.g_q) + (size_t)par * PD_NB * a.A, lane * a.A + q_row, epoch, qv);
            PD_STAMP_LANE0(17);
        }
        pd_segment<SIG, CELL_DEC, 8, 20>(w, lt, xA, lane, sl);
        pd_segment<SIG, CELL_ATT, 24, 20>(w, lt, xA, lane, sl);
        PD_STAMP(3);
        PD_GATHER(8, PD_DM, a.g_ctx, X + XCTX, 3u);
        PD_STAMP(4);
        if (fail) break;
        // ---- C: decoder RNN on the fresh context columns (model.py:741-747)
        pd_segment<SIG, CELL_DEC, 0, 8>(w, lt, xC, lane, sl);
        pd_cells<SIG, CELL_DEC>(sl, pd_xbuf(a.xb, a.g_dech, par, PD_RD), pd_xbuf(a.xb, a.g_dech, par ^ 1, PD_RD), PD_RD, FD, lane);
        PD_STAMP(13);
        pd_segment<SIG, CELL_ATT, 4, 8>(w, lt, xC, lane, sl);                   // attention RNN of step + 1 on ctx(step)
        PD_STAMP(5);
        PD_GATHER(12, PD_RD, a.g_dech, X + XDEC, 4u);
        PD_STAMP(6);
        if (fail) break;
        // ---- D: second decoder RNN on the fresh decoder-hidden columns (model.py:749-755)
        pd_segment<SIG, CELL_D2, 0, 12>(w, lt, xD, lane, sl);
        pd_cells<SIG, CELL_D2>(sl, pd_xbuf(a.xb, a.g_d2h, par, PD_RD), pd_xbuf(a.xb, a.g_d2h, par ^ 1, PD_RD), PD_RD, F2, lane);
        PD_STAMP(14);
        // the part of the projection rows that does not need d2_h.  (The products of step + 1 on dec_h(step) - the attention RNN's
        // decoder-hidden columns, the decoder RNN's recurrent ones - ran HERE until round 5: 1.2 us between this workgroup's d2_h
        // publish and its d2_h gather, more than the exchange needs.  They now hide in the h1 / prenet exchanges below, which had
        // nothing to cover; every slot still accumulates its segments in the same order: bit-identical.)
        if (wave == 1 || (wave == 2 && pr_row1 >= 0)) {
            const float* wrow = L + WPR + (wave == 1 ? 0 : Dp);
            pd_row<12>(wrow, xD, lane, accP);                                    // W[:, :768] . dec_h
            pd_row<8>(wrow + PD_RD, xC, lane, accP);                             // W[:, 768:] . ctx
        }
        PD_STAMP(7);
        PD_GATHER(12, PD_RD, a.g_d2h, X + XD2, 5u);
        PD_STAMP(8);
        if (fail) break;
        // ---- E: projection row set on [dec_h + d2_h | ctx] (model.py:757-765; rows: mel, gate, folded prenet layer 1)
        if (wave == 1 || (wave == 2 && pr_row1 >= 0)) {
            const int row = wave == 1 ? pr_row0 : pr_row1;
            const float* wrow = L + WPR + (wave == 1 ? 0 : Dp);
            pd_row<12>(wrow, x2, lane, accP);                                    // W[:, :768] . d2_h   (the residual sum, model.py:755)
            const float val = pd_block_sum(accP[0]) + bpr;
            accP = pd_f4{0.f, 0.f, 0.f, 0.f};
            if (lane < PD_NB) {
                const int b = lane;
                if (row < a.n_mel) {
                    if (b < a.batch) a.mel_out[((size_t)b * a.n_mel + row) * a.max_steps + step] = val;
                } else if (row == a.n_mel) {
                    if (b < a.batch) a.gate_out[(size_t)b * a.max_steps + step] = val;
                } else if (have_next) {     // first prenet layer of step + 1: relu, always-on dropout (model.py:187-190)
                    const int j = row - a.n_mel - 1;
                    const bool kp = keep_next != 0;        // (lanes >= batch loaded nothing: 0)
                    publish_x(pd_xbuf(a.xb, a.g_h1, par, PD_P), pd_xbuf(a.xb, a.g_h1, par ^ 1, PD_P), b * PD_P + j, kp ? fmaxf(val, 0.f) * 2.0f : 0.0f);
                }
            }
            if (wave == 1) PD_STAMP_LANE0(15);
        }
        pd_segment<SIG, CELL_D2, 12, 12>(w, lt, x2, lane, sl);                  // second decoder RNN of step + 1 on d2_h(step)
        pd_segment<SIG, CELL_ATT, 12, 12>(w, lt, xD, lane, sl);                 // attention RNN of step + 1 on dec_h(step) (see D)
        if (have_next) {
            PD_STAMP(9);
            PD_GATHER(4, PD_P, a.g_h1, X + XH1, 6u);
            PD_STAMP(10);
            if (fail) break;
            // ---- F: second prenet layer rows
            if (wave == 3 || (wave == 0 && w2_row1 >= 0)) {
                const int k = wave == 3 ? 0 : 1;
                const int row = k == 0 ? w2_row0 : w2_row1;
                pd_f4 o = {0.f, 0.f, 0.f, 0.f};
                pd_row<4>(L + WW2 + k * PD_P, xH, lane, o);
                const float val = pd_block_sum(o[0]);
                if (lane < PD_NB) {
                    const int b = lane;
                    const bool kp = keep_next != 0;
                    const float pv = kp ? fmaxf(val, 0.f) * 2.0f : 0.0f;
                    S.pown[k][b] = pv;
                    publish_x(pd_xbuf(a.xb, a.g_p, par, PD_P), pd_xbuf(a.xb, a.g_p, par ^ 1, PD_P), b * PD_P + row, pv);
                }
                if (wave == 3) PD_STAMP_LANE0(16);
            }
            pd_segment<SIG, CELL_DEC, 28, 12>(w, lt, xD, lane, sl);             // decoder RNN of step + 1: recurrent columns on dec_h(step) (see D)
            PD_STAMP(11);
            PD_GATHER(4, PD_P, a.g_p, X + XP, 7u);
            PD_STAMP(12);
            if (fail) break;
        }
    }
#undef PD_GATHER
    if (fail) return;
    // ---- exit: this wave's slices of the state for the next launch
    __syncthreads();
    static_for<0, 3>([&](auto ss) {
        constexpr int s = decltype(ss)::value;
        constexpr int cell = SIG::cell[s];
        if constexpr (cell != CELL_NONE) {
            constexpr int H = cell == CELL_ATT ? PD_RA : PD_RD;
            const int urow = (cell == CELL_ATT ? FA : cell == CELL_DEC ? FD : F2) + sl[s].unit;
            const unsigned cst = cell == CELL_ATT ? a.att_c : cell == CELL_DEC ? a.dec_c : a.d2_c;
            const unsigned hout = cell == CELL_ATT ? a.att_h_out : cell == CELL_DEC ? a.dec_h_out : a.d2_h_out;
            if (lane < nbc) { (a.ws + hout)[lane * H + urow] = sl[s].h; (a.ws + cst)[lane * H + urow] = sl[s].c; }
        }
    });
    if (t < 2 * PD_NB) {
        const int k = t / PD_NB, b = t % PD_NB;
        const int row = k == 0 ? w2_row0 : w2_row1;
        if (row >= 0 && b < nbc) (a.ws + a.prenet)[b * PD_P + row] = S.pown[k][b];
    }
}

// The four waves of an LSTM workgroup of each class (slot = unit index inside the workgroup's units of that cell):
//   class 0 (6,2,3): (S0 D0 A0) (S1 D1 A1) (S2 A2 A3) (A4 A5)         class 1 (5,4,2): (S0 D0 A0) (S1 D1 A1) (D2 A2) (D3 A3 A4)
//   class 2 (5,3,4): (S0 D0 A0) (S1 D1 A1) (S2 D2 A2) (S3 A3 A4)      class 3 (5,3,3): (S0 D0 A0) (S1 D1 A1) (S2 D2 A2) (A3 A4)
// LDS tiles: an S-D-A wave 16 at 16 * wave; the S-A-A wave 20 at 32 (class 0) / 48 (class 2); the D-A-A wave 36 at 32.
template <bool DBG>
__device__ __forceinline__ void lstm_workgroup(const PdArgs& a, float* L, LstmLds& S, int wg) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cls = wg < PD_C0 ? 0 : wg < PD_C1 ? 1 : wg < PD_C2 ? 2 : 3;
    // ONE call (= one inlined copy of the step loop) per signature; slots and the first LDS tile are runtime values.  (Two
    // inlined copies of one signature with different constants mis-computed one slot: profiles/HISTORY.md, round 4.)
    PdSlots su;
    int lt0 = 16 * wave;
    if (wave < 2 || (wave == 2 && cls >= 2)) {
        su.unit[0] = wave; su.unit[1] = wave; su.unit[2] = wave;
        lstm_wave<SigSDA, DBG>(a, L, S, wg, su, lt0);
    } else if ((wave == 2 && cls == 0) || (wave == 3 && cls == 2)) {
        su.unit[0] = wave; su.unit[1] = wave == 2 ? 2 : 3; su.unit[2] = wave == 2 ? 3 : 4;
        lstm_wave<SigSAA, DBG>(a, L, S, wg, su, lt0);
    } else if (wave == 3 && cls == 1) {
        su.unit[0] = 3; su.unit[1] = 3; su.unit[2] = 4; lt0 = 32;
        lstm_wave<SigDAA, DBG>(a, L, S, wg, su, lt0);
    } else if (wave == 2 && cls == 1) {
        su.unit[0] = 2; su.unit[1] = 2; su.unit[2] = -1;
        lstm_wave<SigDA, DBG>(a, L, S, wg, su, lt0);
    } else {        // wave 3 of classes 0 and 3
        su.unit[0] = cls == 0 ? 4 : 3; su.unit[1] = cls == 0 ? 5 : 4; su.unit[2] = -1;
        lstm_wave<SigAA, DBG>(a, L, S, wg, su, lt0);
    }
}

// DBG = true: the same kernel with the s_memrealtime stamps of ctts_taco_decoder_persistent_debug compiled in.
template <bool DBG>
__global__ __launch_bounds__(PD_T, 1) void taco_persistent_kernel(const PdArgs a) {
    __shared__ __attribute__((aligned(16))) float L[LSTM_FLOATS];
    __shared__ LstmLds S;
    const int wg = blockIdx.x;
    unsigned* ctl = reinterpret_cast<unsigned*>(a.xb + a.ctl);
    if (__hip_atomic_load((gu32*)ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)
        return;                                  // an earlier launch on this exchange buffer gave up: stay down
    if (wg >= PD_LWG) attention_workgroup<DBG>(a, *reinterpret_cast<AttnLds*>(L), wg);
    else lstm_workgroup<DBG>(a, L, S, wg);
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
    CTTS_CHECK_HIP(hipMemsetAsync(exchange, 0xFF, x.ctl * sizeof(u64), s));     // every granule "not yet" (X vectors: PD_SENT; q: a tag no step has), before every launch
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
    {
        const Tuning tune = tuning();
        for (int i = 0; i < 6; ++i) a.dly[i] = tune.taco_poll_delay[i];
    }
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
