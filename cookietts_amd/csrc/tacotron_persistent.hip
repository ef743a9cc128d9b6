// Persistent Tacotron2-TM decoder loop for gfx950: ONE launch runs a whole block of decoder steps
// (model.py:668-767 per step) on all 256 CUs, instead of six dependent launches per step.
//
// Why: a decoder step is a chain of small dependent mat-vecs (27 M fp32 weights, batch <= 4).  As six launches per
// step it cost 85 us/step: every launch pays a dependent boundary + a cold ramp of its weight stream, and the two
// one-workgroup-per-utterance stages serialise the rest.  Here every workgroup (one per CU) owns a fixed slice of
// the work for the whole block:
//   * 5 attention-RNN units, 3 decoder-RNN units, 3 second-decoder-RNN units (4 gate rows each),
//   * one row of the query projection (workgroups < attention_dim), one or two rows of the projection row set
//     [mel | gate | first prenet layer folded through the mel projection], one row of the second prenet layer,
//   * workgroups 0..3 additionally run the windowed location-sensitive attention of utterance b = workgroup id.
// The columns of every mat-vec whose input is produced in the SAME step ("fresh": prenet -> attention RNN, context ->
// decoder RNN, decoder hidden -> second decoder RNN, and the small rows) are weight-stationary in registers for the
// whole launch; the columns whose input was produced earlier (recurrent states, previous context: 79 % of the weights)
// are streamed from L2 / Infinity Cache in the gaps while the workgroup would otherwise wait for an exchange, into
// partial pre-activations - so they never sit on the critical path.
// Exchanges: a vector produced by many workgroups and needed by all (att_h, q, ctx, dec_h, d2_h, h1, prenet) is
// all-gathered through 8-byte {tag = step + 1, value} granules written with ONE agent-scope (write-through) store
// each and polled with agent-scope loads: the data is the flag, no fences, no grid barrier (MI355X_MICROARCH.md,
// "handoff"/"allgather" rows; cdna_hip_programming.md Guideline 16 R2).  Every poll loop is bounded; on a timeout
// the workgroup records (code, workgroup, phase, step) in the control words and the whole grid drains.
// Granule buffers and control words are zeroed by the host wrapper before EVERY launch.
#include "tacotron_plan.h"

namespace ctts {
namespace {

using namespace taco;

constexpr int PD_WG = 256;          // workgroups == CUs of an MI355X
constexpr int PD_T = 512;           // threads per workgroup (8 waves, 2 per SIMD)
constexpr int PD_NB = 4;            // batch, padded
constexpr int PD_UA = 5, PD_UD = 3; // LSTM units per workgroup: 1280 / 256, 768 / 256
constexpr int PD_RA = PD_UA * PD_WG, PD_RD = PD_UD * PD_WG, PD_P = 256, PD_DM = 512;
constexpr int PD_AMAX = 256, PD_TMAX = 1024, PD_W = 33, PD_FMAX = 32, PD_KMAX = 31;
// LDS vector store X: [b][n] per vector
constexpr int XP = 0, XCTX = XP + PD_NB * PD_P, XDEC = XCTX + PD_NB * PD_DM, XATT = XDEC + PD_NB * PD_RD,
              XD2 = XATT + PD_NB * PD_RA, XH1 = XD2 + PD_NB * PD_RD, X_FLOATS = XH1 + PD_NB * PD_P;
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
    int A, F, K, R, n_mel, T, batch, nbc, step0, n_steps, max_steps, pd_rows;   // nbc: batch rows the workspace holds
};

__device__ __forceinline__ float pd_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

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
        __builtin_amdgcn_s_sleep(1);
    }
}

__device__ __forceinline__ float wave_total(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// This wave's rows of one LSTM cell: local rows r = wave + 8 i (i < NR) of the workgroup's 4 U rows; local row r is
// gate r / U of unit r % U, i.e. weight row (r / U) * H + wg * U + r % U.
template <int U>
__device__ __forceinline__ int cell_row(int r, int wg) { return (r / U) * (U * PD_WG) + wg * U + (r % U); }

// acc[i][b] += sum over columns [col0, col0 + 256 NJ) of W[row_i][c] * x[b][c - col0 + xoff] for this lane's columns
// (no cross-lane reduction).  NJ <= 3: at most NR * 3 sixteen-byte loads per lane in flight (8 waves -> >= 48 KiB per CU).
template <int U, int NR, int NJ>
__device__ __forceinline__ void stream_accum(const float* __restrict__ W, int ldw, int col0, const float* xs, int n, int xoff,
                                             int wg, int wave, int lane, float (&acc)[NR][PD_NB]) {
    float4 w[NR][NJ];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int r = wave + 8 * i;
        const bool valid = r < 4 * U;                                   // wave-uniform
        const float* rp = W + (size_t)cell_row<U>(valid ? r : 0, wg) * ldw + col0 + 4 * lane;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            w[i][j] = valid ? *reinterpret_cast<const float4*>(rp + 256 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        float4 x[PD_NB];
#pragma unroll
        for (int b = 0; b < PD_NB; ++b) x[b] = *reinterpret_cast<const float4*>(xs + b * n + xoff + 4 * (lane + 64 * j));
#pragma unroll
        for (int i = 0; i < NR; ++i)
#pragma unroll
            for (int b = 0; b < PD_NB; ++b)
                acc[i][b] += w[i][j].x * x[b].x + w[i][j].y * x[b].y + w[i][j].z * x[b].z + w[i][j].w * x[b].w;
    }
}

template <int U, int NR, int NJ>
__device__ __forceinline__ void stream_rows(const float* __restrict__ W, int ldw, int col0, const float* xs, int n, int wg,
                                            int wave, int lane, float (&acc)[NR][PD_NB]) {
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
        for (int b = 0; b < PD_NB; ++b) acc[i][b] = 0.f;
    if constexpr (NJ <= 3) {
        stream_accum<U, NR, NJ>(W, ldw, col0, xs, n, 0, wg, wave, lane, acc);
    } else {
        stream_accum<U, NR, 3>(W, ldw, col0, xs, n, 0, wg, wave, lane, acc);
        stream_accum<U, NR, NJ - 3>(W, ldw, col0 + 768, xs, n, 768, wg, wave, lane, acc);
    }
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
        for (int b = 0; b < PD_NB; ++b) acc[i][b] = wave_total(acc[i][b]);
}

// lanes 0..3 of the owning wave add the part into the partial pre-activations gp[local row][b]
template <int U, int NR>
__device__ __forceinline__ void add_partial(float (*gp)[PD_NB], const float (&acc)[NR][PD_NB], int wave, int lane) {
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int r = wave + 8 * i;
        if (r < 4 * U && lane < PD_NB) {
            const float v = lane == 0 ? acc[i][0] : lane == 1 ? acc[i][1] : lane == 2 ? acc[i][2] : acc[i][3];
            gp[r][lane] += v;
        }
    }
}

template <int U, int NR, int NJ>
__device__ __forceinline__ void early_part(const float* __restrict__ W, int ldw, int col0, const float* xs, int n,
                                           float (*gp)[PD_NB], int wg, int wave, int lane) {
    float acc[NR][PD_NB];
    stream_rows<U, NR, NJ>(W, ldw, col0, xs, n, wg, wave, lane, acc);
    add_partial<U, NR>(gp, acc, wave, lane);
}

// fresh columns out of registers: gates[r][b] = gp[r][b] + bias[r] + sum_c wf[i][j] . x[b][c]
template <int U, int NR, int NJ>
__device__ __forceinline__ void fresh_part(const float4 (&wf)[NR][NJ], const float (&bias)[NR], const float* xs, int n,
                                           const float (*gp)[PD_NB], float (*gates)[PD_NB], int wave, int lane) {
    float acc[NR][PD_NB];
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
        for (int b = 0; b < PD_NB; ++b) acc[i][b] = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        float4 x[PD_NB];
#pragma unroll
        for (int b = 0; b < PD_NB; ++b) x[b] = *reinterpret_cast<const float4*>(xs + b * n + 4 * (lane + 64 * j));
#pragma unroll
        for (int i = 0; i < NR; ++i)
#pragma unroll
            for (int b = 0; b < PD_NB; ++b)
                acc[i][b] += wf[i][j].x * x[b].x + wf[i][j].y * x[b].y + wf[i][j].z * x[b].z + wf[i][j].w * x[b].w;
    }
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int r = wave + 8 * i;
#pragma unroll
        for (int b = 0; b < PD_NB; ++b) acc[i][b] = wave_total(acc[i][b]);
        if (r < 4 * U && lane < PD_NB) {
            const float v = lane == 0 ? acc[i][0] : lane == 1 ? acc[i][1] : lane == 2 ? acc[i][2] : acc[i][3];
            gates[r][lane] = (gp[r][lane] + v) + bias[i];
        }
    }
}

// LSTM cell update of the workgroup's U units (layers.py:308-372, gate order i, f, g, o); publishes h'.
template <int U>
__device__ __forceinline__ void cell_update(const float (*gates)[PD_NB], float (*cst)[PD_NB], float (*hown)[PD_NB], u64* g,
                                            unsigned epoch, int wg, int t) {
    if (t < U * PD_NB) {
        const int u = t / PD_NB, b = t % PD_NB;
        const float gi = pd_sigmoid(gates[0 * U + u][b]), gf = pd_sigmoid(gates[1 * U + u][b]);
        const float gg = tanhf(gates[2 * U + u][b]), go = pd_sigmoid(gates[3 * U + u][b]);
        const float c = gf * cst[u][b] + gi * gg;
        const float h = go * tanhf(c);
        cst[u][b] = c;
        hown[u][b] = h;
        publish(g, b * (U * PD_WG) + wg * U + u, epoch, h);
    }
}

struct AttnLds {
    float pmw[PD_W * PD_AMAX];
    float wloc[PD_FMAX * 2 * PD_KMAX];
    float loc[PD_W][PD_FMAX + 1];
    float wcat[2][PD_W + PD_KMAX - 1];
    float q[PD_AMAX];
    float en[64], wts[64];
    float w[PD_TMAX], cum[PD_TMAX];
    float pos;
};

// Windowed location-sensitive attention of utterance b (model.py:93-161, 49-65), the arithmetic of
// attention_window_body (tacotron_decoder.hip) on 512 threads: previous / cumulative weights and the position live in
// LDS across steps, the memory window is read from L2.  Publishes the context granules of b.
__device__ __forceinline__ void pd_attention(const PdArgs& a, AttnLds& s, int b, unsigned epoch, int step, u64* g_ctx) {
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
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
        for (int i = t; i < 2 * (W + a.K - 1); i += PD_T) {
            const int c = i / (W + a.K - 1), j = i % (W + a.K - 1);
            const int pos = s0 - padk + j;
            const float* src = c == 0 ? s.w : s.cum;
            s.wcat[c][j] = (pos >= 0 && pos < a.T) ? src[pos] : 0.f;
        }
    }
    __syncthreads();
    for (int i = t; i < W * a.F; i += PD_T) {
        const int tt = i / a.F, f = i % a.F;
        float acc = 0.f;
        for (int c = 0; c < 2; ++c)
            for (int j = 0; j < a.K; ++j) acc = fmaf(s.wloc[(f * 2 + c) * a.K + j], s.wcat[c][tt + j], acc);
        s.loc[tt][f] = acc;
    }
    __syncthreads();
    {
        constexpr int MAXP = 5;                  // ceil(33 / 8) window positions per wave
        float epart[MAXP];
#pragma unroll
        for (int i = 0; i < MAXP; ++i) epart[i] = 0.f;
        for (int ad = lane; ad < a.A; ad += 64) {
            float wd[PD_FMAX];
#pragma unroll
            for (int f = 0; f < PD_FMAX; ++f) wd[f] = (a.blob + a.Wd)[(size_t)min(f, a.F - 1) * a.A + ad];
            const float qa = s.q[ad], va = (a.blob + a.v)[ad];
#pragma unroll
            for (int i = 0; i < MAXP; ++i) {
                const int tt = min(wv + 8 * i, W - 1);
                float acc = 0.f;
#pragma unroll
                for (int f = 0; f < PD_FMAX; ++f) acc = fmaf(f < a.F ? wd[f] : 0.f, s.loc[tt][f], acc);
                acc += qa;
                acc += s.pmw[tt * a.A + ad];
                const float th = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc * 2.8853900817779268f));
                epart[i] = fmaf(va, th, epart[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            const int tt = wv + 8 * i;
            const float e = wave_total(epart[i]);
            if (tt < W && lane == 0) {
                const int pos = s0 + tt;
                s.en[tt] = (pos < len && pos < a.T) ? e : -INFINITY;
            }
        }
    }
    __syncthreads();
    if (wv == 0) {
        const float e = lane < W ? s.en[lane] : -INFINITY;
        float m = e;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        const float pexp = lane < W ? expf(e - m) : 0.f;
        const float sum = wave_total(pexp);
        const float wgt = pexp / sum;
        if (lane < W) s.wts[lane] = wgt;
        const float np = wave_total(lane < W ? wgt * (float)(s0 + lane) : 0.f);
        if (lane == 0) {
            const float sf = pd_sigmoid((a.blob + a.scalars)[1]);
            s.pos = s.pos * sf + np * (1.0f - sf);
        }
    }
    __syncthreads();
    for (int d = t; d < PD_DM; d += PD_T) {
        const float* mp = (a.ws + a.memory) + (size_t)b * a.T * PD_DM + d;
        float acc = 0.f;
#pragma unroll 11
        for (int tt = 0; tt < W; ++tt) {
            const int pos = min(s0 + tt, a.T - 1);
            acc = fmaf(s0 + tt < a.T ? s.wts[tt] : 0.f, mp[(size_t)pos * PD_DM], acc);
        }
        publish(g_ctx, b * PD_DM + d, epoch, acc);
    }
    for (int p = t; p < a.T; p += PD_T) {
        const float wgt = (p >= s0 && p < s0 + W) ? s.wts[p - s0] : 0.f;
        s.w[p] = wgt;
        s.cum[p] += wgt;
        a.align_out[((size_t)b * a.max_steps + step) * a.T + p] = wgt;
    }
    __syncthreads();
}

__global__ __launch_bounds__(PD_T, 2) void taco_persistent_kernel(const PdArgs a) {
    __shared__ __attribute__((aligned(16))) float X[X_FLOATS];
    __shared__ float gpA[2][4 * PD_UA][PD_NB], gpD[2][4 * PD_UD][PD_NB], gp2[2][4 * PD_UD][PD_NB];
    __shared__ float gates[4 * PD_UA][PD_NB];
    __shared__ float cA[PD_UA][PD_NB], cD[PD_UD][PD_NB], c2[PD_UD][PD_NB];
    __shared__ float hA[PD_UA][PD_NB], hD[PD_UD][PD_NB], h2[PD_UD][PD_NB];
    __shared__ float pown[PD_NB];
    __shared__ AttnLds att;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wg = blockIdx.x;
    const int I_att = PD_P + PD_DM + PD_RD, I_dec = PD_RA + PD_DM, Dp = PD_RD + PD_DM;
    const bool is_attn = wg < PD_NB;
    if (__hip_atomic_load((gu32*)reinterpret_cast<unsigned*>(a.xb + a.ctl), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)
        return;                                  // an earlier launch on this exchange buffer gave up: stay down

    // ---- weight-stationary part: the fresh columns and the small rows, in registers for the whole launch ----
    float4 wfA[3][1], wfD[2][2], wf2[2][3], wsm[5];
    float bA[3], bD[2], b2[2], bsm = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int r = wave + 8 * i;
        const bool valid = r < 4 * PD_UA;
        const int row = cell_row<PD_UA>(valid ? r : 0, wg);
        wfA[i][0] = valid ? *reinterpret_cast<const float4*>((a.blob + a.att_wih) + (size_t)row * I_att + 4 * lane) : make_float4(0.f, 0.f, 0.f, 0.f);
        bA[i] = valid ? (a.blob + a.att_bih)[row] + (a.blob + a.att_bhh)[row] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = wave + 8 * i;
        const bool valid = r < 4 * PD_UD;
        const int row = cell_row<PD_UD>(valid ? r : 0, wg);
#pragma unroll
        for (int j = 0; j < 2; ++j)
            wfD[i][j] = valid ? *reinterpret_cast<const float4*>((a.blob + a.dec_wih) + (size_t)row * I_dec + PD_RA + 4 * (lane + 64 * j))
                              : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < 3; ++j)
            wf2[i][j] = valid ? *reinterpret_cast<const float4*>((a.blob + a.d2_wih) + (size_t)row * PD_RD + 4 * (lane + 64 * j))
                              : make_float4(0.f, 0.f, 0.f, 0.f);
        bD[i] = valid ? (a.blob + a.dec_bih)[row] + (a.blob + a.dec_bhh)[row] : 0.f;
        b2[i] = valid ? (a.blob + a.d2_bih)[row] + (a.blob + a.d2_bhh)[row] : 0.f;
    }
    // small rows: wave 0 = query row wg; wave 1 = projection row wg; wave 2 = projection row 256 + wg; wave 3 = W2 row wg
    int sm_row = -1;
#pragma unroll
    for (int j = 0; j < 5; ++j) wsm[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (wave == 0 && wg < a.A) {
        sm_row = wg;
#pragma unroll
        for (int j = 0; j < 5; ++j) wsm[j] = *reinterpret_cast<const float4*>((a.blob + a.Wq) + (size_t)wg * PD_RA + 4 * (lane + 64 * j));
    } else if (wave == 1 || wave == 2) {
        const int row = wave == 1 ? wg : PD_WG + wg;
        if (row < a.pd_rows) {
            sm_row = row;
#pragma unroll
            for (int j = 0; j < 5; ++j) wsm[j] = *reinterpret_cast<const float4*>((a.blob + a.Wproj) + (size_t)row * Dp + 4 * (lane + 64 * j));
            bsm = (a.blob + a.bproj)[row];
        }
    } else if (wave == 3) {
        sm_row = wg;
        wsm[0] = *reinterpret_cast<const float4*>((a.blob + a.W2) + (size_t)wg * PD_P + 4 * lane);
    }

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
    if (t < PD_UA * PD_NB) {
        const int u = t / PD_NB, b = t % PD_NB;
        cA[u][b] = b < nbc ? (a.ws + a.att_c)[b * PD_RA + wg * PD_UA + u] : 0.f;
        hA[u][b] = b < nbc ? (a.ws + a.att_h_in)[b * PD_RA + wg * PD_UA + u] : 0.f;
    }
    if (t < PD_UD * PD_NB) {
        const int u = t / PD_NB, b = t % PD_NB;
        cD[u][b] = b < nbc ? (a.ws + a.dec_c)[b * PD_RD + wg * PD_UD + u] : 0.f;
        hD[u][b] = b < nbc ? (a.ws + a.dec_h_in)[b * PD_RD + wg * PD_UD + u] : 0.f;
        c2[u][b] = b < nbc ? (a.ws + a.d2_c)[b * PD_RD + wg * PD_UD + u] : 0.f;
        h2[u][b] = b < nbc ? (a.ws + a.d2_h_in)[b * PD_RD + wg * PD_UD + u] : 0.f;
    }
    if (t < PD_NB) pown[t] = t < nbc ? (a.ws + a.prenet)[t * PD_P + wg] : 0.f;
    for (int i = t; i < 2 * 4 * PD_UA * PD_NB; i += PD_T) (&gpA[0][0][0])[i] = 0.f;
    for (int i = t; i < 2 * 4 * PD_UD * PD_NB; i += PD_T) { (&gpD[0][0][0])[i] = 0.f; (&gp2[0][0][0])[i] = 0.f; }
    if (is_attn) {
        const int b = wg;
        const bool real = b < a.batch;
        for (int p = t; p < a.T; p += PD_T) {
            att.w[p] = real ? (a.ws + a.w)[(size_t)b * a.T + p] : 0.f;
            att.cum[p] = real ? (a.ws + a.cum)[(size_t)b * a.T + p] : 0.f;
        }
        for (int i = t; i < a.F * 2 * a.K; i += PD_T) att.wloc[i] = (a.blob + a.Wloc)[i];
        if (t == 0) att.pos = real ? (a.ws + a.pos)[b] : 0.f;
    }
    __syncthreads();
    int cur = 0;
    // early parts of step0, in the order the loop accumulates them
    early_part<PD_UA, 3, 2>((a.blob + a.att_wih), I_att, PD_P, X + XCTX, PD_DM, gpA[cur], wg, wave, lane);
    early_part<PD_UA, 3, 3>((a.blob + a.att_wih), I_att, PD_P + PD_DM, X + XDEC, PD_RD, gpA[cur], wg, wave, lane);
    early_part<PD_UA, 3, 5>((a.blob + a.att_whh), PD_RA, 0, X + XATT, PD_RA, gpA[cur], wg, wave, lane);
    early_part<PD_UD, 2, 3>((a.blob + a.dec_whh), PD_RD, 0, X + XDEC, PD_RD, gpD[cur], wg, wave, lane);
    early_part<PD_UD, 2, 3>((a.blob + a.d2_whh), PD_RD, 0, X + XD2, PD_RD, gp2[cur], wg, wave, lane);
    __syncthreads();

    bool fail = false;
#define PD_GATHER(NPT, buf, count, dst, phase)                                                                       \
    do {                                                                                                               \
        const bool ok_ = gather<NPT>((buf) + (size_t)par * (count), (count), (dst), epoch, reinterpret_cast<unsigned*>(a.xb + a.ctl), t, (phase), (unsigned)step); \
        if (__syncthreads_or(ok_ ? 0 : 1)) { fail = true; }                                                            \
    } while (0)

    for (int step = a.step0; step < a.step0 + a.n_steps && !fail; ++step) {
        const unsigned epoch = (unsigned)step + 1u;
        const int par = step & 1, nxt = cur ^ 1;
        // ---- A: attention RNN on the fresh prenet columns (model.py:707-717)
        fresh_part<PD_UA, 3, 1>(wfA, bA, X + XP, PD_P, gpA[cur], gates, wave, lane);
        __syncthreads();
        cell_update<PD_UA>(gates, cA, hA, (a.xb + a.g_atth) + (size_t)par * PD_NB * PD_RA, epoch, wg, t);
        // reset the partial sums of step + 1 while waiting (own rows only, written by lanes 0..3 of the owning wave later)
        for (int i = t; i < 4 * PD_UA * PD_NB; i += PD_T) (&gpA[nxt][0][0])[i] = 0.f;
        for (int i = t; i < 4 * PD_UD * PD_NB; i += PD_T) { (&gpD[nxt][0][0])[i] = 0.f; (&gp2[nxt][0][0])[i] = 0.f; }
        PD_GATHER(10, (a.xb + a.g_atth), PD_NB * PD_RA, X + XATT, 1u);
        if (fail) break;
        // ---- B: query row (model.py:126 query_layer), then the decoder RNN's att_h columns as an early part
        if (wave == 0 && sm_row >= 0) {
            float q[PD_NB];
#pragma unroll
            for (int b = 0; b < PD_NB; ++b) {
                float acc = 0.f;
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const float4 x = *reinterpret_cast<const float4*>(X + XATT + b * PD_RA + 4 * (lane + 64 * j));
                    acc += wsm[j].x * x.x + wsm[j].y * x.y + wsm[j].z * x.z + wsm[j].w * x.w;
                }
                q[b] = wave_total(acc);
            }
            if (lane < PD_NB)
                publish((a.xb + a.g_q) + (size_t)par * PD_NB * a.A, lane * a.A + sm_row, epoch,
                        lane == 0 ? q[0] : lane == 1 ? q[1] : lane == 2 ? q[2] : q[3]);
        }
        early_part<PD_UD, 2, 5>((a.blob + a.dec_wih), I_dec, 0, X + XATT, PD_RA, gpD[cur], wg, wave, lane);
        if (is_attn) {
            const int b = wg;
            const bool ok_ = gather<1>((a.xb + a.g_q) + (size_t)par * PD_NB * a.A + (size_t)b * a.A, a.A, att.q, epoch, reinterpret_cast<unsigned*>(a.xb + a.ctl), t, 2u, (unsigned)step);
            if (__syncthreads_or(ok_ ? 0 : 1)) { fail = true; break; }
            if (b < a.batch) {
                pd_attention(a, att, b, epoch, step, (a.xb + a.g_ctx) + (size_t)par * PD_NB * PD_DM);
            } else {
                for (int d = t; d < PD_DM; d += PD_T) publish((a.xb + a.g_ctx) + (size_t)par * PD_NB * PD_DM, b * PD_DM + d, epoch, 0.f);
            }
        }
        PD_GATHER(4, (a.xb + a.g_ctx), PD_NB * PD_DM, X + XCTX, 3u);
        if (fail) break;
        // ---- C: decoder RNN on the fresh context columns (model.py:741-747)
        fresh_part<PD_UD, 2, 2>(wfD, bD, X + XCTX, PD_DM, gpD[cur], gates, wave, lane);
        __syncthreads();
        cell_update<PD_UD>(gates, cD, hD, (a.xb + a.g_dech) + (size_t)par * PD_NB * PD_RD, epoch, wg, t);
        early_part<PD_UA, 3, 2>((a.blob + a.att_wih), I_att, PD_P, X + XCTX, PD_DM, gpA[nxt], wg, wave, lane);
        PD_GATHER(6, (a.xb + a.g_dech), PD_NB * PD_RD, X + XDEC, 4u);
        if (fail) break;
        // ---- D: second decoder RNN on the fresh decoder-hidden columns (model.py:749-755)
        fresh_part<PD_UD, 2, 3>(wf2, b2, X + XDEC, PD_RD, gp2[cur], gates, wave, lane);
        __syncthreads();
        cell_update<PD_UD>(gates, c2, h2, (a.xb + a.g_d2h) + (size_t)par * PD_NB * PD_RD, epoch, wg, t);
        early_part<PD_UA, 3, 3>((a.blob + a.att_wih), I_att, PD_P + PD_DM, X + XDEC, PD_RD, gpA[nxt], wg, wave, lane);
        early_part<PD_UD, 2, 3>((a.blob + a.dec_whh), PD_RD, 0, X + XDEC, PD_RD, gpD[nxt], wg, wave, lane);
        PD_GATHER(6, (a.xb + a.g_d2h), PD_NB * PD_RD, X + XD2, 5u);
        if (fail) break;
        // ---- E: projection row set on [dec_h + d2_h | ctx] (model.py:757-765; rows: mel, gate, folded prenet layer 1)
        const bool have_next = step + 1 < a.max_steps;
        if ((wave == 1 || wave == 2) && sm_row >= 0) {
            float o[PD_NB];
#pragma unroll
            for (int b = 0; b < PD_NB; ++b) {
                float acc = 0.f;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const float4 x1 = *reinterpret_cast<const float4*>(X + XDEC + b * PD_RD + 4 * (lane + 64 * j));
                    const float4 x2 = *reinterpret_cast<const float4*>(X + XD2 + b * PD_RD + 4 * (lane + 64 * j));
                    acc += wsm[j].x * (x1.x + x2.x) + wsm[j].y * (x1.y + x2.y) + wsm[j].z * (x1.z + x2.z) + wsm[j].w * (x1.w + x2.w);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float4 x = *reinterpret_cast<const float4*>(X + XCTX + b * PD_DM + 4 * (lane + 64 * j));
                    acc += wsm[3 + j].x * x.x + wsm[3 + j].y * x.y + wsm[3 + j].z * x.z + wsm[3 + j].w * x.w;
                }
                o[b] = wave_total(acc) + bsm;
            }
            if (lane < PD_NB) {
                const int b = lane;
                const float val = lane == 0 ? o[0] : lane == 1 ? o[1] : lane == 2 ? o[2] : o[3];
                if (sm_row < a.n_mel) {
                    if (b < a.batch) a.mel_out[((size_t)b * a.n_mel + sm_row) * a.max_steps + step] = val;
                } else if (sm_row == a.n_mel) {
                    if (b < a.batch) a.gate_out[(size_t)b * a.max_steps + step] = val;
                } else if (have_next) {     // first prenet layer of step + 1: relu, always-on dropout (model.py:187-190)
                    const int j = sm_row - a.n_mel - 1;
                    const bool kp = b < a.batch && a.keep[(((size_t)(step + 1) * 2 + 0) * a.batch + b) * PD_P + j] != 0;
                    publish((a.xb + a.g_h1) + (size_t)par * PD_NB * PD_P, b * PD_P + j, epoch, kp ? fmaxf(val, 0.f) * 2.0f : 0.0f);
                }
            }
        }
        early_part<PD_UD, 2, 3>((a.blob + a.d2_whh), PD_RD, 0, X + XD2, PD_RD, gp2[nxt], wg, wave, lane);
        if (have_next) {
            PD_GATHER(2, (a.xb + a.g_h1), PD_NB * PD_P, X + XH1, 6u);
            if (fail) break;
            // ---- F: second prenet layer row
            if (wave == 3) {
                float o[PD_NB];
#pragma unroll
                for (int b = 0; b < PD_NB; ++b) {
                    const float4 x = *reinterpret_cast<const float4*>(X + XH1 + b * PD_P + 4 * lane);
                    o[b] = wave_total(wsm[0].x * x.x + wsm[0].y * x.y + wsm[0].z * x.z + wsm[0].w * x.w);
                }
                if (lane < PD_NB) {
                    const int b = lane;
                    const float val = lane == 0 ? o[0] : lane == 1 ? o[1] : lane == 2 ? o[2] : o[3];
                    const bool kp = b < a.batch && a.keep[(((size_t)(step + 1) * 2 + 1) * a.batch + b) * PD_P + sm_row] != 0;
                    const float pv = kp ? fmaxf(val, 0.f) * 2.0f : 0.0f;
                    pown[b] = pv;
                    publish((a.xb + a.g_p) + (size_t)par * PD_NB * PD_P, b * PD_P + sm_row, epoch, pv);
                }
            }
            early_part<PD_UA, 3, 5>((a.blob + a.att_whh), PD_RA, 0, X + XATT, PD_RA, gpA[nxt], wg, wave, lane);
            PD_GATHER(2, (a.xb + a.g_p), PD_NB * PD_P, X + XP, 7u);
            if (fail) break;
        }
        cur = nxt;
    }
#undef PD_GATHER
    if (fail) return;
    // ---- exit: this workgroup's slices of the state for the next launch
    __syncthreads();
    if (t < PD_UA * PD_NB) {
        const int u = t / PD_NB, b = t % PD_NB;
        if (b < nbc) {
            (a.ws + a.att_h_out)[b * PD_RA + wg * PD_UA + u] = hA[u][b];
            (a.ws + a.att_c)[b * PD_RA + wg * PD_UA + u] = cA[u][b];
        }
    }
    if (t < PD_UD * PD_NB) {
        const int u = t / PD_NB, b = t % PD_NB;
        if (b < nbc) {
            (a.ws + a.dec_h_out)[b * PD_RD + wg * PD_UD + u] = hD[u][b];
            (a.ws + a.dec_c)[b * PD_RD + wg * PD_UD + u] = cD[u][b];
            (a.ws + a.d2_h_out)[b * PD_RD + wg * PD_UD + u] = h2[u][b];
            (a.ws + a.d2_c)[b * PD_RD + wg * PD_UD + u] = c2[u][b];
        }
    }
    if (t < nbc) (a.ws + a.prenet)[t * PD_P + wg] = pown[t];
    if (is_attn && wg < a.batch) {
        const int b = wg;
        for (int p = t; p < a.T; p += PD_T) { (a.ws + a.w)[(size_t)b * a.T + p] = att.w[p]; (a.ws + a.cum)[(size_t)b * a.T + p] = att.cum[p]; }
        if (t == 0) (a.ws + a.pos)[b] = att.pos;
        for (int d = t; d < PD_DM; d += PD_T) (a.ws + a.ctx)[b * PD_DM + d] = X[XCTX + b * PD_DM + d];
    }
}

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
    x.ctl = take(8);          // control words LAST: sticky across launches (zeroed by the owner of the buffer, once)
    x.total = o;
}

bool pd_supported(const DecPlan& p, int batch, int T) {
    const auto& c = p.c;
    return c.attention_rnn_dim == PD_RA && c.decoder_rnn_dim == PD_RD && c.second_decoder_rnn_dim == PD_RD &&
           c.prenet_dim == PD_P && c.memory_dim == PD_DM && c.attention_dim <= PD_AMAX && c.attention_dim % 4 == 0 &&
           c.attention_dim <= PD_WG && c.location_n_filters <= PD_FMAX && c.location_kernel_size <= PD_KMAX &&
           c.window_range == 16 && p.pd_rows <= 2 * PD_WG && batch >= 1 && batch <= PD_NB && T >= 1 && T <= PD_TMAX;
}

}  // namespace
}  // namespace ctts

using namespace ctts;

extern "C" {

size_t ctts_taco_decoder_persistent_bytes(const ctts_taco_decoder_config* cfg, int32_t batch, int32_t text_len) {
    DecPlan p;
    if (make_dec_plan(cfg, p) || !pd_supported(p, batch, text_len)) return 0;
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
    hipLaunchKernelGGL(taco_persistent_kernel, dim3(PD_WG), dim3(PD_T), 0, s, a);
    CTTS_CHECK_LAUNCH("taco_persistent");
    return CTTS_OK;
}

}  // extern "C"
