// Tacotron2-TM decoder loop (model.py:668-767, 851-916) for gfx950.
//
// One mel frame per step, strictly sequential, batch of a few utterances: every step re-reads the
// 27 M LSTM/projection weights (108 MB fp32), so the step is bound by weight streaming (HBM /
// Infinity Cache), not by arithmetic.  Round-1 structure: five dependent launches per step
//   lstm_step (attention RNN) -> attention_step -> lstm_step (decoder RNN) -> lstm_step (2nd decoder RNN)
//   -> project_prenet (mel + gate projection, next step's prenet)
// each sized so its weight stream is spread over ~256 workgroups (one per CU):
//   lstm_step: a workgroup owns R hidden units; wave g streams the R rows of gate g (i,f,g,o) of W_ih|W_hh
//   with 16-byte loads against the batch's input vectors staged in LDS, 64-lane shuffle reductions, then
//   the cell update for its units - the gate pre-activations never leave the CU.
//   attention_step: one workgroup per utterance; windowed attention (+-16 tokens) means only 33
//   energies are finite, so location conv, energies, softmax (wave-level reductions), context and the
//   expected position are computed for the window only; weights outside it are exact zeros.
#include "tacotron_plan.h"
#include "taco_math.h"
#include "tuning.h"

namespace ctts {
namespace {

using namespace taco;

typedef float vfloat4 __attribute__((ext_vector_type(4)));
// streaming (non-temporal) 16-byte load: the LSTM weight streams are read once per step and must not
// evict the small re-used matrices (query / projection / prenet) from L2
__device__ __forceinline__ float4 load4_nt(const float* p) {
    const vfloat4 v = __builtin_nontemporal_load(reinterpret_cast<const vfloat4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}

// y[r] = dot(W[r][:], x) for rows r0..r0+RB-1 at once: all RB*ceil(K/256) 16-byte weight loads of the
// wave are issued before the first use, so the batch costs ~one memory latency instead of RB.
// x lives in LDS.  K % 4 == 0.  Result valid in every lane.
template <int RB, int KMAX>
__device__ __forceinline__ void wave_dots(const float* __restrict__ W, size_t ldw, int r0, int rows, int K,
                                          const float* xs, int lane, float (&out)[RB]) {
    constexpr int NK = (KMAX + 255) / 256;
    float4 w[RB][NK];
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < NK; ++j) {
            const int k = lane * 4 + j * 256;
            w[i][j] = (r0 + i < rows && k < K) ? *reinterpret_cast<const float4*>(W + (size_t)(r0 + i) * ldw + k)
                                               : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < NK; ++j) {
            const int k = lane * 4 + j * 256;
            if (k < K) {
                const float4 x = *reinterpret_cast<const float4*>(xs + k);
                acc += w[i][j].x * x.x + w[i][j].y * x.y + w[i][j].z * x.z + w[i][j].w * x.w;
            }
        }
        out[i] = acc;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int i = 0; i < RB; ++i) out[i] += __shfl_xor(out[i], off);
}

// dst[k][o] = src[o][k]
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int O, int K) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)O * K) return;
    const int k = i / O, o = i % O;
    dst[i] = src[(size_t)o * K + k];
}

// Persistent-decoder pack: the first prenet layer (no bias, model.py:187) applied to the mel projection is linear in the
// projection input, so h1_pre = W1 (Wp v + bp) = (W1 Wp) v + W1 bp is one row set of the same matvec as mel and gate:
//   dst_w[j][k] = sum_m W1[j][m] * Wp[m][k],   dst_b[j] = sum_m W1[j][m] * bp[m]      (W1 [P][n_mel], Wp [n_mel][D])
__global__ __launch_bounds__(256) void fold_prenet_proj_kernel(const float* __restrict__ W1, const float* __restrict__ Wp,
                                                               const float* __restrict__ bp, float* __restrict__ dst_w,
                                                               float* __restrict__ dst_b, int P, int n_mel, int D) {
    const int j = blockIdx.x;
    for (int k = threadIdx.x; k < D; k += 256) {
        float acc = 0.f;
        for (int m = 0; m < n_mel; ++m) acc = fmaf(W1[(size_t)j * n_mel + m], Wp[(size_t)m * D + k], acc);
        dst_w[(size_t)j * D + k] = acc;
    }
    if (threadIdx.x == 0) {
        float acc = 0.f;
        for (int m = 0; m < n_mel; ++m) acc = fmaf(W1[(size_t)j * n_mel + m], bp[m], acc);
        dst_b[j] = acc;
    }
}

// out[r][o] = sum_k in[r][k] * WT[k][o]  (4 rows per workgroup share every weight load)
__global__ __launch_bounds__(256) void linear_rows_kernel(const float* __restrict__ in, const float* __restrict__ WT,
                                                          float* __restrict__ out, int rows, int K, int O) {
    extern __shared__ __attribute__((aligned(16))) float xs[];   // [4][K]
    const int r0 = blockIdx.x * 4;
    for (int i = threadIdx.x; i < 4 * K; i += 256) {
        const int r = i / K, k = i % K;
        xs[i] = (r0 + r < rows) ? in[(size_t)(r0 + r) * K + k] : 0.f;
    }
    __syncthreads();
    for (int o = threadIdx.x; o < O; o += 256) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        for (int k = 0; k < K; ++k) {
            const float w = WT[(size_t)k * O + o];
            a0 = fmaf(xs[k], w, a0); a1 = fmaf(xs[K + k], w, a1);
            a2 = fmaf(xs[2 * K + k], w, a2); a3 = fmaf(xs[3 * K + k], w, a3);
        }
        if (r0 + 0 < rows) out[(size_t)(r0 + 0) * O + o] = a0;
        if (r0 + 1 < rows) out[(size_t)(r0 + 1) * O + o] = a1;
        if (r0 + 2 < rows) out[(size_t)(r0 + 2) * O + o] = a2;
        if (r0 + 3 < rows) out[(size_t)(r0 + 3) * O + o] = a3;
    }
}

// ---- LSTM cell step ------------------------------------------------------------------
// Sequence mode (packed-sequence nn.LSTM, one direction): the input projection W_ih x_t + b was computed for
// all t by one GEMM (gadd), item b is active while step < lengths[b] and reads/writes time index
// t_b = step (forward) or lengths[b]-1-step (reverse); inactive items keep their state.
struct LstmSeq {
    const float* gadd;        // [B][4H][ld] padded layout, NULL = decoder mode
    long long ga_bstride; int ga_ld, ga_pad;
    const int* lengths; int step, reverse;
    float* out; long long out_bstride; int out_tstride, out_col;
    float* hn; int hn_stride, hn_col;
};

// Column window of one launch.  The decoder splits every cell's matvec in two: the columns whose inputs are known
// early (recurrent state, previous context) are summed by a `gout` launch on a side stream while the few-workgroup
// attention / projection kernels hold the critical path, and the cell launch streams only the fresh columns and adds
// those partial pre-activations (`gin`).  Default {nullptr, nullptr, 0, I, 1} = the whole cell in one launch.
struct LstmPart {
    const float* gin;   // [NB][4H] partial pre-activations to add, or NULL
    float* gout;        // if set: write this launch's raw partial sums [NB][4H] and stop (no bias, no cell update)
    int ih_k0, ih_k1;   // W_ih columns [ih_k0, ih_k1) (multiples of 4)
    int do_hh;          // W_hh . h_old
};

// Everything one LSTM-step workgroup needs (the cell input is the concatenation in0|in1|in2 of [NB][n_i] pieces).
struct LstmCall {
    const float *Wih, *Whh, *bih, *bhh;
    const float *in0; int n0;
    const float *in1; int n1;
    const float *in2; int n2;
    const float* h_old; float* h_new; float* c;
    LstmSeq sq; LstmPart pt;
    int I, H, batch;
};

// One workgroup = R hidden units x 4 gates x NB items; `blk` = which R units; smem >= NB*(I+H) + 4*R*NB floats.
template <int NB, int R>
__device__ __forceinline__ void lstm_body(const LstmCall& q, float* __restrict__ smem, int blk) {
    const float* __restrict__ Wih = q.Wih; const float* __restrict__ Whh = q.Whh;
    const float* __restrict__ bih = q.bih; const float* __restrict__ bhh = q.bhh;
    const float* __restrict__ in0 = q.in0; const float* __restrict__ in1 = q.in1; const float* __restrict__ in2 = q.in2;
    const int n0 = q.n0, n1 = q.n1, n2 = q.n2;
    const float* __restrict__ h_old = q.h_old; float* __restrict__ h_new = q.h_new; float* __restrict__ c = q.c;
    const LstmSeq& sq = q.sq; const LstmPart& pt = q.pt;
    const int I = q.I, H = q.H, batch = q.batch;
    const int K = I + H;
    float* xs = smem;                       // [NB][K]   cell input | previous hidden
    float* gates = smem + NB * K;           // [4][R][NB]
    const int t = threadIdx.x, lane = t & 63, g = t >> 6;
    const int u0 = blk * R;
    // first weight slab group: issued now, consumed after the staging barrier
    constexpr int CH = 4;
    float4 pre[CH][R];
    const bool use_ih = pt.ih_k1 > pt.ih_k0;
    const float* pre_base = use_ih ? Wih + (size_t)(g * H + u0) * I + pt.ih_k0 : Whh + (size_t)(g * H + u0) * H;
    const int pre_ld = use_ih ? I : H;
    const int pre_len = use_ih ? pt.ih_k1 - pt.ih_k0 : (pt.do_hh ? H : 0), pre_xoff = use_ih ? pt.ih_k0 : I;
#pragma unroll
    for (int j = 0; j < CH; ++j) {
        const int k = lane * 4 + j * 256;
        const int kc = k < pre_len ? k : 0;
#pragma unroll
        for (int r = 0; r < R; ++r) pre[j][r] = load4_nt(pre_base + (size_t)r * pre_ld + kc);
    }
    // stage the needed columns of [cell input | previous hidden] for all NB items: 16-byte loads, SB of them in
    // flight per thread (piece boundaries n0, n0+n1, I and the window bounds are multiples of 4).  Only the launch's
    // column window is walked: a "fresh columns" launch of the decoder touches 256 of 2816 columns.
    {
        constexpr int SB = 6;
        const int nih4 = use_ih ? (pt.ih_k1 - pt.ih_k0) / 4 : 0;
        const int per4 = nih4 + (pt.do_hh ? H / 4 : 0);      // 16-byte units per item
        const int total4 = NB * per4;
        for (int base = t; base < total4; base += 256 * SB) {
            float4 v[SB];
            int kk[SB];
#pragma unroll
            for (int j = 0; j < SB; ++j) {
                const int i4 = base + j * 256;
                const int ic = i4 < total4 ? i4 : 0;
                const int b = ic / per4, u = ic % per4;
                const int k = u < nih4 ? pt.ih_k0 + 4 * u : I + 4 * (u - nih4);
                const float* src;
                if (k < n0) src = in0 + b * n0 + k;
                else if (k < n0 + n1) src = in1 + b * n1 + (k - n0);
                else if (k < I) src = in2 + b * n2 + (k - n0 - n1);
                else src = h_old + b * H + (k - I);
                v[j] = *reinterpret_cast<const float4*>(src);
                kk[j] = b * K + k;
            }
#pragma unroll
            for (int j = 0; j < SB; ++j)
                if (base + j * 256 < total4) *reinterpret_cast<float4*>(xs + kk[j]) = v[j];
        }
    }
    __syncthreads();
    float acc[R][NB];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[r][b] = 0.f;
    // Weight streaming: CH k-slabs x R rows of 16-byte non-temporal loads are issued back to back before
    // the first FMA (CH*R*16 B = 256-320 B per lane, ~80 KB per CU in flight), which is what it takes to
    // cover HBM latency with one workgroup per CU.  The first slab group was issued before the input
    // staging (`pre`), so the weight stream is already in flight while x is being staged.
#define CTTS_CONSUME(WV, K0, KLEN, XOFF)                                                                    \
    _Pragma("unroll") for (int j = 0; j < CH; ++j) {                                                        \
        const int k = (K0) + j * 256;                                                                       \
        if (k < (KLEN)) {                                                                                   \
            float4 xv[NB];                                                                                  \
            _Pragma("unroll") for (int b = 0; b < NB; ++b)                                                  \
                xv[b] = *reinterpret_cast<const float4*>(xs + b * K + (XOFF) + k);                          \
            _Pragma("unroll") for (int r = 0; r < R; ++r)                                                   \
                _Pragma("unroll") for (int b = 0; b < NB; ++b)                                              \
                    acc[r][b] += WV[j][r].x * xv[b].x + WV[j][r].y * xv[b].y + WV[j][r].z * xv[b].z +       \
                                 WV[j][r].w * xv[b].w;                                                      \
        }                                                                                                   \
    }
    auto stream = [&](const float* wbase, int ldw, int klen, int xoff, int kstart) {
        for (int k0 = kstart + lane * 4; k0 < klen; k0 += 256 * CH) {
            float4 wv[CH][R];
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const int k = k0 + j * 256;
                const int kc = k < klen ? k : 0;                 // clamped: the tail slab re-reads slab 0
#pragma unroll
                for (int r = 0; r < R; ++r) wv[j][r] = load4_nt(wbase + (size_t)r * ldw + kc);
            }
            CTTS_CONSUME(wv, k0, klen, xoff)
        }
    };
    CTTS_CONSUME(pre, lane * 4, pre_len, pre_xoff)
    if (use_ih) {
        stream(pre_base, I, pre_len, pt.ih_k0, 256 * CH);
        if (pt.do_hh) stream(Whh + (size_t)(g * H + u0) * H, H, H, I, 0);
    } else if (pt.do_hh) {
        stream(pre_base, H, H, I, 256 * CH);
    }
    if (pt.gout) {      // partial launch: raw sums only
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const float s = wave_sum(acc[r][b]);
                if (lane == 0) pt.gout[(size_t)b * 4 * H + g * H + u0 + r] = s;
            }
        return;
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const float s = wave_sum(acc[r][b]);
            if (lane == 0) gates[(g * R + r) * NB + b] = s + (bih ? bih[g * H + u0 + r] + bhh[g * H + u0 + r] : 0.f);
        }
    __syncthreads();
    if (t < R * NB) {
        const int r = t / NB, b = t % NB;
        const int idx = b * H + u0 + r;
        float pre[4];
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) pre[gg] = gates[(gg * R + r) * NB + b];
        if (pt.gin) {
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) pre[gg] += pt.gin[(size_t)b * 4 * H + gg * H + u0 + r];
        }
        bool active = true;
        int tb = 0;
        if (sq.gadd) {
            const int len = b < batch ? sq.lengths[b] : 0;
            active = sq.step < len;
            tb = sq.reverse ? len - 1 - sq.step : sq.step;
            if (active) {
#pragma unroll
                for (int gg = 0; gg < 4; ++gg)
                    pre[gg] += sq.gadd[(size_t)b * sq.ga_bstride + (size_t)(gg * H + u0 + r) * sq.ga_ld + sq.ga_pad + tb];
            }
        }
        if (active) {
            const float ig = sigmoidf_(pre[0]), fg = sigmoidf_(pre[1]), gg = tanhf(pre[2]), og = sigmoidf_(pre[3]);
            const float cy = fg * c[idx] + ig * gg;
            const float hy = og * tanhf(cy);
            c[idx] = cy;
            h_new[idx] = hy;
            if (sq.gadd) {
                sq.out[(size_t)b * sq.out_bstride + (size_t)tb * sq.out_tstride + sq.out_col + u0 + r] = hy;
                if (sq.step == sq.lengths[b] - 1) sq.hn[(size_t)b * sq.hn_stride + sq.hn_col + u0 + r] = hy;
            }
        } else {
            h_new[idx] = h_old[idx];
        }
    }
}

template <int NB, int R>
__global__ __launch_bounds__(256) void lstm_step_kernel(const LstmCall q) {
    extern __shared__ __attribute__((aligned(16))) float lstm_smem[];
    lstm_body<NB, R>(q, lstm_smem, blockIdx.x);
}

// Both directions of a bidirectional layer advance in ONE launch per time step (blockIdx.y = direction): the encoder's
// BiLSTM is 2 x T dependent launches of ~6.7 us otherwise, i.e. launch-bound.
template <int NB, int R>
__global__ __launch_bounds__(256) void lstm_step2_kernel(const LstmCall q0, const LstmCall q1) {
    extern __shared__ __attribute__((aligned(16))) float lstm_smem[];
    if (blockIdx.y == 0) lstm_body<NB, R>(q0, lstm_smem, blockIdx.x);
    else lstm_body<NB, R>(q1, lstm_smem, blockIdx.x);
}

// ---- attention step --------------------------------------------------------------------
struct AttnArgs {
    const float *Wq, *v, *Wloc, *Wd, *scalars;
    const float* G;                        // batched form: the location conv folded into the location-dense layer, [2][K][A]
    const float *att_h, *memory, *pm;
    float *w, *cum, *ctx, *pos, *align_out;
    const int* lengths;
    int T, A, Ra, Dm, F, K, R, step, max_steps;
};

template <int FMAX>
__global__ __launch_bounds__(256) void attention_step_kernel(const AttnArgs a) {
    __shared__ float q[256];
    __shared__ __attribute__((aligned(16))) float hs[1536];
    __shared__ float wcat[2][128];
    __shared__ float wloc[64 * 2 * 63];
    __shared__ float loc[64][65];
    __shared__ float en[64];
    __shared__ float wts[64];
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int W = 2 * a.R + 1, padk = (a.K - 1) / 2;
    // window start (model.py:131-140); round = half-to-even
    const int len = a.lengths[b];
    float cur = a.pos[b];
    const float off = a.scalars[0];
    if (off != 0.f) cur += off;
    cur = fminf(fmaxf(cur, (float)a.R), (float)(len - 1 - a.R));
    const int s = (int)rintf(fmaxf(cur - (float)a.R, 0.f));
    // query projection: each wave owns rows wv*RQ.. in batches of 8 (att_h staged in LDS)
    for (int i = t; i < a.Ra; i += 256) hs[i] = a.att_h[(size_t)b * a.Ra + i];
    __syncthreads();
    {
        const int per = (a.A + 3) / 4;
        for (int r0 = wv * per; r0 < (wv + 1) * per && r0 < a.A; r0 += 8) {
            float o[8];
            const int lim = min(a.A, (wv + 1) * per);
            wave_dots<8, 1536>(a.Wq, a.Ra, r0, lim, a.Ra, hs, lane, o);
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (lane == i && r0 + i < lim) q[r0 + i] = o[i];
        }
    }
    // previous / cumulative weights around the window, location conv filters
    for (int i = t; i < 2 * (W + a.K - 1); i += 256) {
        const int c = i / (W + a.K - 1), j = i % (W + a.K - 1);
        const int pos = s - padk + j;
        const float* src = c == 0 ? a.w : a.cum;
        wcat[c][j] = (pos >= 0 && pos < a.T) ? src[(size_t)b * a.T + pos] : 0.f;
    }
    for (int i = t; i < a.F * 2 * a.K; i += 256) wloc[i] = a.Wloc[i];
    __syncthreads();
    for (int i = t; i < W * a.F; i += 256) {
        const int tt = i / a.F, f = i % a.F;
        float acc = 0.f;
        for (int c = 0; c < 2; ++c)
            for (int j = 0; j < a.K; ++j) acc = fmaf(wloc[(f * 2 + c) * a.K + j], wcat[c][tt + j], acc);
        loc[tt][f] = acc;
    }
    __syncthreads();
    // energies: wave wv owns window positions wv, wv+4, ...; lane owns attention dims lane, lane+64, ...
    // The lane's rows of the location-dense weight (coalesced, from the transposed copy WdT[f][a]) and the
    // processed-memory values of its positions are loaded up front, unconditionally (positions are clamped,
    // masked ones are overwritten with -inf afterwards), so all global loads of a pass are in flight together.
    {
        constexpr int MAXP = 16;                 // ceil(63 / 4) window positions per wave
        float epart[MAXP];
#pragma unroll
        for (int i = 0; i < MAXP; ++i) epart[i] = 0.f;
        for (int ad = lane; ad < a.A; ad += 64) {
            float wd[FMAX], pmv[MAXP];
#pragma unroll
            for (int f = 0; f < FMAX; ++f) wd[f] = a.Wd[(size_t)min(f, a.F - 1) * a.A + ad];
#pragma unroll
            for (int i = 0; i < MAXP; ++i) {
                const int pos = min(s + min(wv + 4 * i, W - 1), a.T - 1);
                pmv[i] = a.pm[((size_t)b * a.T + pos) * a.A + ad];
            }
            const float qa = q[ad], va = a.v[ad];
#pragma unroll
            for (int i = 0; i < MAXP; ++i) {
                const int tt = min(wv + 4 * i, W - 1);
                float acc = 0.f;
#pragma unroll
                for (int f = 0; f < FMAX; ++f) acc = fmaf(f < a.F ? wd[f] : 0.f, loc[tt][f], acc);
                acc += qa;
                acc += pmv[i];
                epart[i] = fmaf(va, tanhf(acc), epart[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            const int tt = wv + 4 * i;
            const float e = wave_sum(epart[i]);
            if (tt < W && lane == 0) {
                const int pos = s + tt;
                en[tt] = (pos < len && pos < a.T) ? e : -INFINITY;
            }
        }
    }
    __syncthreads();
    if (wv == 0) {     // softmax over the window with wave-level reductions
        const float e = lane < W ? en[lane] : -INFINITY;
        const float m = wave_max(e);
        const float pexp = lane < W ? expf(e - m) : 0.f;
        const float sum = wave_sum(pexp);
        const float wgt = pexp / sum;
        if (lane < W) wts[lane] = wgt;
        const float np = wave_sum(lane < W ? wgt * (float)(s + lane) : 0.f);
        if (lane == 0) {
            const float sf = sigmoidf_(a.scalars[1]);
            a.pos[b] = a.pos[b] * sf + np * (1.0f - sf);
        }
    }
    __syncthreads();
    // context = sum_t w[t] * memory[t]: positions clamped (their weight is exactly 0 when masked) so the
    // loads are unconditional and an unrolled slab of them is in flight at once
    for (int d = t; d < a.Dm; d += 256) {
        const float* mp = a.memory + (size_t)b * a.T * a.Dm + d;
        float acc = 0.f;
#pragma unroll 11
        for (int tt = 0; tt < W; ++tt) {
            const int pos = min(s + tt, a.T - 1);
            acc = fmaf(s + tt < a.T ? wts[tt] : 0.f, mp[(size_t)pos * a.Dm], acc);
        }
        a.ctx[(size_t)b * a.Dm + d] = acc;
    }
    for (int p = t; p < a.T; p += 256) {
        const float wgt = (p >= s && p < s + W) ? wts[p - s] : 0.f;
        a.w[(size_t)b * a.T + p] = wgt;
        a.cum[(size_t)b * a.T + p] += wgt;
        a.align_out[((size_t)b * a.max_steps + a.step) * a.T + p] = wgt;
    }
}

// ---- multi-workgroup GEMV: y[b][r] = sum_k W[r][k] x[b][k] for all NB items at once (query projection) ------
template <int NB>
__global__ __launch_bounds__(256) void gemv_rows_kernel(const float* __restrict__ W, const float* __restrict__ x,
                                                        float* __restrict__ y, int rows, int K) {
    __shared__ __attribute__((aligned(16))) float xs[NB * 1536];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    for (int i = t * 4; i < NB * K; i += 1024) *reinterpret_cast<float4*>(xs + i) = *reinterpret_cast<const float4*>(x + i);
    __syncthreads();
    const int r0 = (blockIdx.x * 4 + wv) * 2;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        float o[2];
        wave_dots<2, 1536>(W, K, r0, rows, K, xs + b * K, lane, o);
        if (lane < 2 && r0 + lane < rows) y[(size_t)b * rows + r0 + lane] = lane == 0 ? o[0] : o[1];
    }
}

// ---- attention step, window fast path -------------------------------------------------------------------
// Same math as attention_step_kernel, for window_range <= 16, memory_dim <= 512, attention_dim <= 256,
// <= 32 location filters of <= 31 taps.  The window start depends only on the PREVIOUS step's position, so every
// global operand (query from gemv_rows_kernel, previous/cumulative weights, the 33-row windows of the processed
// memory and of the memory itself) is fetched up front into LDS in one burst; the rest runs out of LDS.
constexpr int AW = 33, ADM = 512, AAD = 256, AF = 32, AK = 31;

constexpr int ATTN_SMEM_FLOATS = AW * ADM + AW * AAD + AF * 2 * AK + AW * (AF + 1) + 2 * (AW + AK - 1) + 2 + AAD + 64 + 64 + AF * AAD;

__device__ __forceinline__ void attention_window_body(const AttnArgs& a, const float* __restrict__ qbuf,
                                                      float* __restrict__ smem, int b) {
    float* memw = smem;                                   // [AW][ADM]   (16-byte aligned)
    float* pmw = memw + AW * ADM;                         // [AW][AAD]
    float* wloc = pmw + AW * AAD;                         // [AF*2*AK]
    float (*loc)[AF + 1] = reinterpret_cast<float (*)[AF + 1]>(wloc + AF * 2 * AK);       // [AW][AF+1]
    float (*wcat)[AW + AK - 1] = reinterpret_cast<float (*)[AW + AK - 1]>(&loc[AW][0]);   // [2][AW+AK-1]
    float* q = &wcat[2][0] + 2;                           // [AAD]
    float* en = q + AAD;                                  // [64]
    float* wts = en + 64;                                 // [64]
    float* wds = wts + 64;                                // [AF*AAD] location-dense weight [F][A]
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int W = 2 * a.R + 1, padk = (a.K - 1) / 2;
    const int len = a.lengths[b];
    float cur = a.pos[b];
    const float off = a.scalars[0];
    if (off != 0.f) cur += off;
    cur = fminf(fmaxf(cur, (float)a.R), (float)(len - 1 - a.R));
    const int s = (int)rintf(fmaxf(cur - (float)a.R, 0.f));
    // ---- one burst of loads (rows clamped to T-1: their weights are exactly 0 when masked)
    {
        const int dm4 = a.Dm / 4, a4 = a.A / 4;
        for (int i = t; i < W * dm4; i += 256) {
            const int tt = i / dm4, d4 = i % dm4;
            const int pos = min(s + tt, a.T - 1);
            *reinterpret_cast<float4*>(memw + tt * a.Dm + d4 * 4) =
                *reinterpret_cast<const float4*>(a.memory + ((size_t)b * a.T + pos) * a.Dm + d4 * 4);
        }
        for (int i = t; i < W * a4; i += 256) {
            const int tt = i / a4, c4 = i % a4;
            const int pos = min(s + tt, a.T - 1);
            *reinterpret_cast<float4*>(pmw + tt * a.A + c4 * 4) =
                *reinterpret_cast<const float4*>(a.pm + ((size_t)b * a.T + pos) * a.A + c4 * 4);
        }
        for (int i = t; i < 2 * (W + a.K - 1); i += 256) {
            const int c = i / (W + a.K - 1), j = i % (W + a.K - 1);
            const int pos = s - padk + j;
            const float* src = c == 0 ? a.w : a.cum;
            wcat[c][j] = (pos >= 0 && pos < a.T) ? src[(size_t)b * a.T + pos] : 0.f;
        }
        for (int i = t; i < a.F * 2 * a.K; i += 256) wloc[i] = a.Wloc[i];
        for (int i = t; i < a.F * a.A; i += 256) wds[i] = a.Wd[i];
        if (t < a.A) q[t] = qbuf[(size_t)b * a.A + t];
    }
    __syncthreads();
    for (int i = t; i < W * a.F; i += 256) {
        const int tt = i / a.F, f = i % a.F;
        float acc = 0.f;
        for (int c = 0; c < 2; ++c)
            for (int j = 0; j < a.K; ++j) acc = fmaf(wloc[(f * 2 + c) * a.K + j], wcat[c][tt + j], acc);
        loc[tt][f] = acc;
    }
    __syncthreads();
    {
        constexpr int MAXP = 9;                  // ceil(33 / 4) window positions per wave
        float epart[MAXP];
#pragma unroll
        for (int i = 0; i < MAXP; ++i) epart[i] = 0.f;
        for (int ad = lane; ad < a.A; ad += 64) {
            float wd[AF];
#pragma unroll
            for (int f = 0; f < AF; ++f) wd[f] = wds[min(f, a.F - 1) * a.A + ad];
            const float qa = q[ad], va = a.v[ad];
#pragma unroll
            for (int i = 0; i < MAXP; ++i) {
                const int tt = min(wv + 4 * i, W - 1);
                float acc = 0.f;
#pragma unroll
                for (int f = 0; f < AF; ++f) acc = fmaf(f < a.F ? wd[f] : 0.f, loc[tt][f], acc);
                acc += qa;
                acc += pmw[tt * a.A + ad];
                epart[i] = fmaf(va, tanhf(acc), epart[i]);      // libm: the exp2 / rcp form's 1e-7 absolute error is amplified by sharp attention
            }
        }
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            const int tt = wv + 4 * i;
            const float e = wave_sum(epart[i]);
            if (tt < W && lane == 0) {
                const int pos = s + tt;
                en[tt] = (pos < len && pos < a.T) ? e : -INFINITY;
            }
        }
    }
    __syncthreads();
    if (wv == 0) {
        const float e = lane < W ? en[lane] : -INFINITY;
        const float m = wave_max(e);
        const float pexp = lane < W ? expf(e - m) : 0.f;
        const float sum = wave_sum(pexp);
        const float wgt = pexp / sum;
        if (lane < W) wts[lane] = wgt;
        const float np = wave_sum(lane < W ? wgt * (float)(s + lane) : 0.f);
        if (lane == 0) {
            const float sf = sigmoidf_(a.scalars[1]);
            a.pos[b] = a.pos[b] * sf + np * (1.0f - sf);
        }
    }
    __syncthreads();
    for (int d = t; d < a.Dm; d += 256) {
        float acc = 0.f;
#pragma unroll 11
        for (int tt = 0; tt < W; ++tt) acc = fmaf(s + tt < a.T ? wts[tt] : 0.f, memw[tt * a.Dm + d], acc);
        a.ctx[(size_t)b * a.Dm + d] = acc;
    }
    for (int p = t; p < a.T; p += 256) {
        const float wgt = (p >= s && p < s + W) ? wts[p - s] : 0.f;
        a.w[(size_t)b * a.T + p] = wgt;
        a.cum[(size_t)b * a.T + p] += wgt;
        a.align_out[((size_t)b * a.max_steps + a.step) * a.T + p] = wgt;
    }
}

__global__ __launch_bounds__(256) void attention_window_kernel(const AttnArgs a, const float* __restrict__ qbuf) {
    __shared__ __attribute__((aligned(16))) float smem[ATTN_SMEM_FLOATS];
    attention_window_body(a, qbuf, smem, blockIdx.x);
}

// ---- projection + next prenet ------------------------------------------------------------
struct ProjArgs {
    const float *Wp, *bp, *W1T, *W2T;
    const float *dec_h, *d2_h, *ctx;
    const unsigned char* keep;       // masks of the NEXT step: [2][B][P], or NULL on the last step
    float *mel_out, *gate_out, *prenet_out;
    int n_mel, Rd, Dm, P, B, step, max_steps;
};

constexpr int PROJ_SMEM_FLOATS = 2048 + 256 + 256;

__device__ __forceinline__ void project_prenet_body(const ProjArgs& a, float* __restrict__ smem, int b) {
    float* v = smem;            // [2048] (16-byte aligned)
    float* mel = v + 2048;      // [256]
    float* a1 = mel + 256;      // [256]
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int D = a.Rd + a.Dm;
    for (int i = t; i < D; i += 256)
        v[i] = i < a.Rd ? a.dec_h[(size_t)b * a.Rd + i] + a.d2_h[(size_t)b * a.Rd + i] : a.ctx[(size_t)b * a.Dm + (i - a.Rd)];
    __syncthreads();
    {   // rows 0..n_mel-1 = mel, row n_mel = gate; each wave owns a contiguous quarter, 8 rows per batch
        const int rows = a.n_mel + 1, per = (rows + 3) / 4, lim = min(rows, (wv + 1) * per);
        for (int r0 = wv * per; r0 < lim; r0 += 8) {
            float o[8];
            wave_dots<8, 1536>(a.Wp, D, r0, lim, D, v, lane, o);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = r0 + i;
                if (lane == i && r < lim) {
                    const float val = o[i] + a.bp[r];
                    if (r < a.n_mel) {
                        mel[r] = val;
                        a.mel_out[((size_t)b * a.n_mel + r) * a.max_steps + a.step] = val;
                    } else {
                        a.gate_out[(size_t)b * a.max_steps + a.step] = val;
                    }
                }
            }
        }
    }
    __syncthreads();
    if (a.keep == nullptr) return;
    // prenet of the next step: relu(W . x) * keep * 2 twice (model.py:187-190).  Thread j owns output j;
    // weights are stored transposed ([in][out]) so every load is coalesced, and the loads of a 32-deep
    // slab are independent of the running sums (4 accumulators) so they are all in flight together.
    if (t < a.P) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int k = 0; k < a.n_mel; k += 4) {
            acc[0] = fmaf(a.W1T[(size_t)(k + 0) * a.P + t], mel[k + 0], acc[0]);
            acc[1] = fmaf(a.W1T[(size_t)(k + 1) * a.P + t], mel[k + 1], acc[1]);
            acc[2] = fmaf(a.W1T[(size_t)(k + 2) * a.P + t], mel[k + 2], acc[2]);
            acc[3] = fmaf(a.W1T[(size_t)(k + 3) * a.P + t], mel[k + 3], acc[3]);
        }
        a1[t] = fmaxf((acc[0] + acc[1]) + (acc[2] + acc[3]), 0.f) * (a.keep[(size_t)b * a.P + t] ? 2.0f : 0.0f);
    }
    __syncthreads();
    if (t < a.P) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
        for (int k = 0; k < a.P; k += 4) {
            acc[0] = fmaf(a.W2T[(size_t)(k + 0) * a.P + t], a1[k + 0], acc[0]);
            acc[1] = fmaf(a.W2T[(size_t)(k + 1) * a.P + t], a1[k + 1], acc[1]);
            acc[2] = fmaf(a.W2T[(size_t)(k + 2) * a.P + t], a1[k + 2], acc[2]);
            acc[3] = fmaf(a.W2T[(size_t)(k + 3) * a.P + t], a1[k + 3], acc[3]);
        }
        a.prenet_out[(size_t)b * a.P + t] =
            fmaxf((acc[0] + acc[1]) + (acc[2] + acc[3]), 0.f) * (a.keep[((size_t)a.B + b) * a.P + t] ? 2.0f : 0.0f);
    }
}

__global__ __launch_bounds__(256) void project_prenet_kernel(const ProjArgs a) {
    __shared__ __attribute__((aligned(16))) float smem[PROJ_SMEM_FLOATS];
    project_prenet_body(a, smem, blockIdx.x);
}

// ---- heterogeneous launches: critical-path stage + early partial sums of the next cells ----------------------------
// The window attention and the projection/prenet stages are one workgroup per utterance (4 of 256 CUs busy for
// ~25 us each).  The other CUs use that time to stream the part of the NEXT cells' weights whose inputs are already
// known (LstmPart), so the cell launches that follow only touch the fresh columns.  One launch, one stream: workgroups
// [0, n_head) run the head stage, the rest the partial matvecs.  (The same split over two streams with events was
// measured slower than no split: 121 vs 100 us per step.)
template <int NB, int R1, int R2>
__global__ __launch_bounds__(256) void attention_partials_kernel(const AttnArgs a, const float* __restrict__ qbuf,
                                                                 int n_head, const LstmCall p1, int n1, const LstmCall p2) {
    __shared__ __attribute__((aligned(16))) float smem[ATTN_SMEM_FLOATS];
    const int blk = blockIdx.x;
    if (blk < n_head) attention_window_body(a, qbuf, smem, blk);
    else if (blk < n_head + n1) lstm_body<NB, R1>(p1, smem, blk - n_head);
    else lstm_body<NB, R2>(p2, smem, blk - n_head - n1);
}

constexpr int PROJ_FUSED_SMEM_FLOATS = 4 * (1536 + 1280) + 4 * 8 * 4;   // the attention RNN's staging: NB * (I + H) + gates

template <int NB, int R>
__global__ __launch_bounds__(256) void project_partial_kernel(const ProjArgs a, int n_head, const LstmCall p1) {
    __shared__ __attribute__((aligned(16))) float smem[PROJ_FUSED_SMEM_FLOATS];
    static_assert(PROJ_FUSED_SMEM_FLOATS >= PROJ_SMEM_FLOATS, "projection stage does not fit");
    const int blk = blockIdx.x;
    if (blk < n_head) project_prenet_body(a, smem, blk);
    else lstm_body<NB, R>(p1, smem, blk - n_head);
}

template <int NB>
int launch_lstm(const float* wih, const float* whh, const float* bih, const float* bhh, const float* in0, int n0,
                const float* in1, int n1, const float* in2, int n2, const float* h_old, float* h_new, float* c, int I,
                int H, int batch, const LstmSeq& sq, const LstmPart& pt, hipStream_t s) {
    const size_t smem_base = (size_t)NB * (I + H) * sizeof(float);
#define CTTS_LSTM_CASE(RR)                                                                                         \
    if (H % RR == 0 && H / RR <= 256) {                                                                           \
        const size_t smem = smem_base + 4 * RR * NB * sizeof(float);                                              \
        const LstmCall q{wih, whh, bih, bhh, in0, n0, in1, n1, in2, n2, h_old, h_new, c, sq, pt, I, H, batch};    \
        hipLaunchKernelGGL((lstm_step_kernel<NB, RR>), dim3(H / RR), dim3(256), smem, s, q);                      \
        CTTS_CHECK_LAUNCH("lstm_step");                                                                            \
        return CTTS_OK;                                                                                            \
    }
    CTTS_LSTM_CASE(1) CTTS_LSTM_CASE(2) CTTS_LSTM_CASE(3) CTTS_LSTM_CASE(4) CTTS_LSTM_CASE(5) CTTS_LSTM_CASE(6)
    CTTS_LSTM_CASE(8)
#undef CTTS_LSTM_CASE
    set_error("lstm_step: no tiling for hidden size %d", H);
    return CTTS_E_ARG;
}

template <int NB>
int launch_lstm2(const LstmCall& q0, const LstmCall& q1, hipStream_t s) {
    const int I = q0.I, H = q0.H;
    const size_t smem_base = (size_t)NB * (I + H) * sizeof(float);
#define CTTS_LSTM2_CASE(RR)                                                                                        \
    if (H % RR == 0 && H / RR <= 256) {                                                                           \
        const size_t smem = smem_base + 4 * RR * NB * sizeof(float);                                              \
        hipLaunchKernelGGL((lstm_step2_kernel<NB, RR>), dim3(H / RR, 2), dim3(256), smem, s, q0, q1);             \
        CTTS_CHECK_LAUNCH("lstm_step2");                                                                           \
        return CTTS_OK;                                                                                            \
    }
    CTTS_LSTM2_CASE(1) CTTS_LSTM2_CASE(2) CTTS_LSTM2_CASE(3) CTTS_LSTM2_CASE(4) CTTS_LSTM2_CASE(5) CTTS_LSTM2_CASE(6)
    CTTS_LSTM2_CASE(8)
#undef CTTS_LSTM2_CASE
    set_error("lstm_step2: no tiling for hidden size %d", H);
    return CTTS_E_ARG;
}

int launch_lstm_raw(int NB, const float* wih, const float* whh, const float* bih, const float* bhh, const float* in0,
                    int n0, const float* in1, int n1, const float* in2, int n2, const float* h_old, float* h_new, float* c,
                    int I, int H, int batch, const LstmSeq& sq, hipStream_t s, const LstmPart* part = nullptr) {
    const LstmPart whole{nullptr, nullptr, 0, I, 1};
    const LstmPart& pt = part ? *part : whole;
    switch (NB) {
        case 1: return launch_lstm<1>(wih, whh, bih, bhh, in0, n0, in1, n1, in2, n2, h_old, h_new, c, I, H, batch, sq, pt, s);
        case 2: return launch_lstm<2>(wih, whh, bih, bhh, in0, n0, in1, n1, in2, n2, h_old, h_new, c, I, H, batch, sq, pt, s);
        default: return launch_lstm<4>(wih, whh, bih, bhh, in0, n0, in1, n1, in2, n2, h_old, h_new, c, I, H, batch, sq, pt, s);
    }
}

int launch_lstm_nb(int NB, const float* blob, const size_t* off, const float* in0, int n0, const float* in1, int n1,
                   const float* in2, int n2, const float* h_old, float* h_new, float* c, int I, int H, hipStream_t s,
                   const LstmPart* part = nullptr) {
    LstmSeq none{};
    return launch_lstm_raw(NB, blob + off[0], blob + off[1], blob + off[2], blob + off[3], in0, n0, in1, n1, in2, n2, h_old,
                           h_new, c, I, H, NB, none, s, part);
}

LstmCall make_call(const float* blob, const size_t* off, const float* in0, int n0, const float* in1, int n1, const float* in2,
                   int n2, const float* h_old, int I, int H, int NB, const LstmPart& pt) {
    LstmSeq none{};
    return LstmCall{blob + off[0], blob + off[1], blob + off[2], blob + off[3], in0, n0, in1, n1, in2, n2, h_old, nullptr,
                    nullptr, none, pt, I, H, NB};
}

// Decoder.inference(return_hidden_state=True) (model.py:762, 888-889): the vector the gate and the mel projection read,
// hidden[b][k][step] = k < Rd ? dec_h + d2_h (the residual sum, model.py:755-759) : attention context
__global__ __launch_bounds__(256) void hidden_state_kernel(const float* __restrict__ dec_h, const float* __restrict__ d2_h,
                                                           const float* __restrict__ ctx, float* __restrict__ hidden, int Rd, int Dm,
                                                           int step, int max_steps) {
    const int b = blockIdx.x, D = Rd + Dm;
    for (int k = threadIdx.x; k < D; k += 256)
        hidden[((size_t)b * D + k) * max_steps + step] = k < Rd ? dec_h[(size_t)b * Rd + k] + d2_h[(size_t)b * Rd + k] : ctx[(size_t)b * Dm + k - Rd];
}

#include "tacotron_batched.h"

// One decoder step of the batched form (4 < batch <= MAX_BATCH, and batch <= 4 where the persistent kernel is not used): seven
// dependent launches.  Plain schedule (more than 64 items, or CTTS_TACO_BG_NO_PIPE):
//   attention RNN (bg CELL) + the attention's part 1 as extra workgroups -> query rows (bg LINEAR) -> attention part 2 (one
//   workgroup per item) -> decoder RNN -> second decoder RNN -> projection row set [mel | gate | first prenet layer] (bg PROJ)
//   -> second prenet layer (bg PRENET2)
// Pipelined schedule (<= 64 items): the same seven launches, but the K columns of a cell whose inputs exist EARLY are summed in
// the launches of the latency-bound stages (independent roles of one launch, tacotron_batched.h "heterogeneous launches"), and the
// cell's own launch only adds the columns that arrive last:
//   1  attention RNN FINAL: prenet columns + [ctx | dec_h] and [att_h] sums of the previous step's launches 6 / 7; attention part 1
//   2  query rows                 + decoder RNN EARLY on att_h(step)
//   3  attention part 2           + decoder RNN EARLY on dec_h(step - 1) + second decoder RNN EARLY on d2_h(step - 1)
//   4  decoder RNN FINAL: context columns + both sums          5  second decoder RNN FINAL: dec_h(step) columns + its sum
//   6  projection row set         + next step's attention RNN EARLY on [ctx(step) | dec_h(step)]
//   7  second prenet layer        + next step's attention RNN EARLY on att_h(step)
int batched_steps(const DecPlan& p, const DecWs& w, const float* blob, const uint8_t* keep_masks, float* mel_out, float* gate_out,
                  float* align_out, float* hidden_out, int batch, int text_len, int step0, int n_steps, int max_steps, hipStream_t s) {
    const auto& c = p.c;
    const int NB = ws_rows(p, batch);
    const int Pn = c.prenet_dim, Ra = c.attention_rnn_dim, Rd = c.decoder_rnn_dim, Rd2 = c.second_decoder_rnn_dim, Dm = c.memory_dim;
    const int shape = tuning().taco_bg_shape;
    const bool pipe = NB <= 64 && shape == 0 && !tuning().taco_bg_no_pipe;
    const int nyb = NB / 16;
    int rc;
    // the three cells' GEMMs for a given step parity (cur = step & 1): X pieces and state, K range / partial mode set by the caller
    auto att_args = [&](int cur) {
        BgArgs a{};
        a.W = blob + p.bg_att.off; a.rows = p.bg_att.rows; a.batch = batch;
        bg_set_x(a, p.bg_att, w.prenet, Pn, w.ctx, Dm, w.dec_h[cur], Rd, w.att_h[cur], Ra);
        a.bih = blob + p.att[2]; a.bhh = blob + p.att[3]; a.c = w.att_c; a.h_new = w.att_h[cur ^ 1]; a.H = Ra;
        a.part = w.part_att; a.nt_total = nyb;
        return a;
    };
    auto dec_args = [&](int cur) {
        BgArgs a{};
        a.W = blob + p.bg_dec.off; a.rows = p.bg_dec.rows; a.batch = batch;
        bg_set_x(a, p.bg_dec, w.att_h[cur ^ 1], Ra, w.ctx, Dm, w.dec_h[cur], Rd, nullptr, 0);
        a.bih = blob + p.dec[2]; a.bhh = blob + p.dec[3]; a.c = w.dec_c; a.h_new = w.dec_h[cur ^ 1]; a.H = Rd;
        a.part = w.part_dec; a.nt_total = nyb;
        return a;
    };
    auto d2_args = [&](int cur) {
        BgArgs a{};
        a.W = blob + p.bg_d2.off; a.rows = p.bg_d2.rows; a.batch = batch;
        bg_set_x(a, p.bg_d2, w.dec_h[cur ^ 1], Rd, w.d2_h[cur], Rd2, nullptr, 0, nullptr, 0);
        a.bih = blob + p.d2[2]; a.bhh = blob + p.d2[3]; a.c = w.d2_c; a.h_new = w.d2_h[cur ^ 1]; a.H = Rd2;
        a.hsum = w.hsum; a.hres = w.dec_h[cur ^ 1];
        a.part = w.part_d2; a.nt_total = nyb;
        return a;
    };
    auto early = [](BgArgs a, int c0, int c1, int slot) { a.c_begin = c0; a.c_end = c1; a.pmode = 1; a.pslot = slot; return a; };
    auto final_ = [](BgArgs a, int c0, int c1, int nslots) { a.c_begin = c0; a.c_end = c1; a.pmode = 2; a.npart = nslots; return a; };
    // K chunk boundaries of the pieces (bg_set_x order): att [prenet | ctx | dec_h | att_h], dec [att_h | ctx | dec_h], d2 [dec_h | d2_h]
    const int aP = Pn / BG_KC, aD = (Pn + Dm + Rd) / BG_KC, aK = p.bg_att.nchunks;
    const int dA = Ra / BG_KC, dC = (Ra + Dm) / BG_KC, dK = p.bg_dec.nchunks;
    const int sD = Rd / BG_KC, sK = p.bg_d2.nchunks;
    auto attn_args = [&](int step) {
        AttnArgs at{};
        at.Wq = blob + p.query_w; at.v = blob + p.v_w; at.Wloc = blob + p.loc_conv_w; at.Wd = blob + p.loc_dense_w;
        at.G = blob + p.loc_fold;
        at.scalars = blob + p.scalars;
        at.att_h = w.att_h[(step & 1) ^ 1]; at.memory = w.memory; at.pm = w.pm;
        at.w = w.w; at.cum = w.cum; at.ctx = w.ctx; at.pos = w.pos; at.align_out = align_out; at.lengths = w.lengths;
        at.T = text_len; at.A = c.attention_dim; at.Ra = Ra; at.Dm = Dm;
        at.F = c.location_n_filters; at.K = c.location_kernel_size; at.R = c.window_range;
        at.step = step; at.max_steps = max_steps;
        return at;
    };
    if (pipe && n_steps > 0) {      // what the previous call's last step did not leave: the EARLY sums of step0's attention RNN and
        const AttnArgs at0 = attn_args(step0);                  // its attention part 1
        BgRoles m{};
        m.cell[0] = early(att_args(step0 & 1), aP, aD, 0); m.n_cell[0] = 1;
        m.cell[1] = early(att_args(step0 & 1), aD, aK, 1); m.n_cell[1] = 1;
        m.pre = &at0; m.apre = w.apre; m.astart = w.astart; m.n_pre = batch;
        if ((rc = bg_launch_multi8(m, NB, s))) return rc;
    }
    for (int step = step0; step < step0 + n_steps; ++step) {
        const int cur = step & 1, nxt = cur ^ 1;
        const bool more = step + 1 < step0 + n_steps;          // the next step belongs to this call: leave its EARLY sums
        const AttnArgs at = attn_args(step), at_next = attn_args(step + 1);
        // 1: attention RNN on [prenet | context | decoder hidden], recurrent on its own hidden state (model.py:707-717); in the plain
        // schedule the attention's part 1 rides along, in the pipelined one it ran in the previous step's launch 7
        if ((rc = bg_launch_cell(pipe ? final_(att_args(cur), 0, aP, 2) : att_args(cur), NB, pipe ? nullptr : &at, w.apre, w.astart, batch, shape, s))) return rc;
        BgArgs q{};     // 2: query rows (model.py:126)
        q.W = blob + p.bg_q.off; q.rows = p.bg_q.rows; q.batch = batch;
        bg_set_x(q, p.bg_q, w.att_h[nxt], Ra, nullptr, 0, nullptr, 0, nullptr, 0);
        q.y = w.qbuf; q.ldy = c.attention_dim;
        if (pipe) {
            BgRoles m{};
            m.small = q; m.n_small = p.bg_q.tiles * nyb; m.small_epi = BG_EPI_LINEAR;
            m.cell[0] = early(dec_args(cur), 0, dA, 0); m.n_cell[0] = 1;
            if ((rc = bg_launch_multi8(m, NB, s))) return rc;
            // 3: attention part 2 + the EARLY sums on last step's dec_h / d2_h
            if ((rc = bg_launch_post_multi4(at, w.qbuf, w.apre, w.astart, batch, early(dec_args(cur), dC, dK, 1),
                                            early(d2_args(cur), sD, sK, 0), NB, s))) return rc;
            // 4, 5: decoder RNN on [attention hidden | context] (model.py:741-747), second decoder RNN on its output (:749-755)
            if ((rc = bg_launch_cell(final_(dec_args(cur), dA, dC, 2), NB, nullptr, nullptr, nullptr, batch, shape, s))) return rc;
            if ((rc = bg_launch_cell(final_(d2_args(cur), 0, sD, 1), NB, nullptr, nullptr, nullptr, batch, shape, s))) return rc;
        } else {
            if ((rc = bg_launch_small<BG_EPI_LINEAR>(q, NB, s))) return rc;
            if ((rc = BG_ALLOW_LDS(attn_post_kernel, BGA_POST_LDS_BYTES))) return rc;
            hipLaunchKernelGGL(attn_post_kernel, dim3(batch), dim3(256), BGA_POST_LDS_BYTES, s, at, w.qbuf, w.apre, w.astart, tuning().taco_bg_debug);
            CTTS_CHECK_LAUNCH("attn_post");
            if ((rc = bg_launch_cell(dec_args(cur), NB, nullptr, nullptr, nullptr, batch, shape, s))) return rc;
            if ((rc = bg_launch_cell(d2_args(cur), NB, nullptr, nullptr, nullptr, batch, shape, s))) return rc;
        }
        const unsigned char* keep = step + 1 < max_steps ? keep_masks + (size_t)(step + 1) * 2 * batch * Pn : nullptr;
        BgArgs pr{}, w2{};
        // 6: projection row set on [dec_h + d2_h | ctx] (model.py:755-765) + first prenet layer of the next step (:187-190)
        pr.W = blob + p.bg_proj.off; pr.rows = p.bg_proj.rows; pr.batch = batch;
        bg_set_x(pr, p.bg_proj, w.hsum, Rd2, w.ctx, Dm, nullptr, 0, nullptr, 0);
        pr.bias = blob + p.pd_proj_b; pr.keep = keep; pr.mel_out = mel_out; pr.gate_out = gate_out; pr.act_out = w.h1;
        pr.n_mel = c.n_mel_channels; pr.P = Pn; pr.step = step; pr.max_steps = max_steps;
        // 7: second prenet layer
        w2.W = blob + p.bg_w2.off; w2.rows = p.bg_w2.rows; w2.batch = batch;
        bg_set_x(w2, p.bg_w2, w.h1, Pn, nullptr, 0, nullptr, 0, nullptr, 0);
        w2.keep = keep; w2.act_out = w.prenet; w2.P = Pn;
        if (pipe) {
            BgRoles m{};
            m.small = pr; m.n_small = p.bg_proj.tiles * nyb; m.small_epi = BG_EPI_PROJ;
            if (more) { m.cell[0] = early(att_args(nxt), aP, aD, 0); m.n_cell[0] = 1; }
            if ((rc = bg_launch_multi8(m, NB, s))) return rc;
            BgRoles m2{};
            if (keep) { m2.small = w2; m2.n_small = p.bg_w2.tiles * nyb; m2.small_epi = BG_EPI_PRENET2; }
            if (more) {
                m2.cell[0] = early(att_args(nxt), aD, aK, 1); m2.n_cell[0] = 1;
                m2.pre = &at_next; m2.apre = w.apre; m2.astart = w.astart; m2.n_pre = batch;     // (reads what launch 3 left: weights, position)
            }
            if ((rc = bg_launch_multi8(m2, NB, s))) return rc;
        } else {
            if ((rc = bg_launch_small<BG_EPI_PROJ>(pr, NB, s))) return rc;
            if (keep && (rc = bg_launch_small<BG_EPI_PRENET2>(w2, NB, s))) return rc;
        }
        if (hidden_out) {
            hipLaunchKernelGGL(hidden_state_kernel, dim3(batch), dim3(256), 0, s, w.dec_h[nxt], w.d2_h[nxt], w.ctx, hidden_out, Rd, Dm, step, max_steps);
            CTTS_CHECK_LAUNCH("hidden_state");
        }
    }
    return CTTS_OK;
}

// G[c][j][a] = sum_f Wd[f][a] * Wloc[f][c][j]: the location conv (model.py:56-60) folded into the location-dense layer (:61-62)
__global__ __launch_bounds__(256) void location_fold_kernel(const float* __restrict__ Wloc, const float* __restrict__ Wd, float* __restrict__ G,
                                                            int F, int K, int A) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 2 * K * A) return;
    const int ad = i % A, j = (i / A) % K, c = i / (A * K);
    float acc = 0.f;
    for (int f = 0; f < F; ++f) acc = fmaf(Wd[(size_t)f * A + ad], Wloc[(f * 2 + c) * K + j], acc);
    G[i] = acc;
}

int bg_pack(float* dst, const BgMat& m, const BgSeg* segs, int nseg, int interleave_H, hipStream_t s) {
    BgPackArgs a{};
    for (int i = 0; i < nseg; ++i) a.seg[i] = segs[i];
    a.nseg = nseg; a.rows = m.rows; a.tiles = m.tiles; a.nchunks = m.nchunks; a.interleave_H = interleave_H; a.dst = dst + m.off;
    const size_t total = (size_t)m.tiles * m.nchunks * 256;
    hipLaunchKernelGGL(bg_pack_kernel, dim3((unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096)), dim3(256), 0, s, a);
    CTTS_CHECK_LAUNCH("bg_pack");
    return CTTS_OK;
}

static_assert(AW == 33 && ADM == 512 && AAD == 256 && AF == 32 && AK == 31, "make_dec_plan's bg_ok restates these limits");
bool bg_supported(const DecPlan& p) { return p.bg_ok; }

}  // namespace
}  // namespace ctts

using namespace ctts;

extern "C" {

size_t ctts_taco_decoder_packed_bytes(const ctts_taco_decoder_config* cfg) {
    DecPlan p;
    if (make_dec_plan(cfg, p)) return 0;
    return p.total * sizeof(float);
}

int ctts_taco_decoder_pack(const ctts_taco_decoder_config* cfg, const ctts_taco_decoder_weights* w, void* packed,
                           void* stream) {
    DecPlan p;
    int rc = make_dec_plan(cfg, p); if (rc) return rc;
    CTTS_CHECK_ARG(w && packed, "decoder pack: NULL pointer");
    const auto& c = p.c;
    hipStream_t s = as_stream(stream);
    float* blob = static_cast<float*>(packed);
    auto copy = [&](size_t off, const float* src, size_t n) -> int {
        CTTS_CHECK_ARG(src != nullptr, "decoder pack: NULL weight pointer");
        CTTS_CHECK_HIP(hipMemcpyAsync(blob + off, src, n * sizeof(float), hipMemcpyDeviceToDevice, s));
        return CTTS_OK;
    };
    auto transpose = [&](size_t off, const float* src, int O, int K) -> int {
        CTTS_CHECK_ARG(src != nullptr, "decoder pack: NULL weight pointer");
        const size_t n = (size_t)O * K;
        hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, blob + off, O, K);
        CTTS_CHECK_LAUNCH("transpose");
        return CTTS_OK;
    };
    if ((rc = transpose(p.bottleneck_wT, w->bottleneck_w, c.memory_dim, c.memory_in_dim))) return rc;
    if ((rc = transpose(p.memory_wT, w->memory_layer_w, c.attention_dim, c.memory_dim))) return rc;
    if ((rc = copy(p.query_w, w->query_w, (size_t)c.attention_dim * c.attention_rnn_dim))) return rc;
    if ((rc = copy(p.v_w, w->v_w, c.attention_dim))) return rc;
    if ((rc = copy(p.loc_conv_w, w->loc_conv_w, (size_t)c.location_n_filters * 2 * c.location_kernel_size))) return rc;
    if ((rc = transpose(p.loc_dense_w, w->loc_dense_w, c.attention_dim, c.location_n_filters))) return rc;   // -> [F][A]
    if ((rc = transpose(p.prenet_w1, w->prenet_w1, c.prenet_dim, c.n_mel_channels))) return rc;   // -> [n_mel][P]
    if ((rc = transpose(p.prenet_w2, w->prenet_w2, c.prenet_dim, c.prenet_dim))) return rc;       // -> [P][P]^T
    auto lstm = [&](const size_t* off, const ctts_lstm_weights& l, int I, int H) -> int {
        int r;
        if ((r = copy(off[0], l.w_ih, (size_t)4 * H * I))) return r;
        if ((r = copy(off[1], l.w_hh, (size_t)4 * H * H))) return r;
        if ((r = copy(off[2], l.b_ih, 4 * H))) return r;
        return copy(off[3], l.b_hh, 4 * H);
    };
    if ((rc = lstm(p.att, w->att_rnn, p.I_att, c.attention_rnn_dim))) return rc;
    if ((rc = lstm(p.dec, w->dec_rnn, p.I_dec, c.decoder_rnn_dim))) return rc;
    if ((rc = lstm(p.d2, w->dec2_rnn, p.I_d2, c.second_decoder_rnn_dim))) return rc;
    if ((rc = copy(p.proj_w, w->proj_w, (size_t)c.n_mel_channels * p.Dproj))) return rc;
    if ((rc = copy(p.proj_w + (size_t)c.n_mel_channels * p.Dproj, w->gate_w, p.Dproj))) return rc;
    if ((rc = copy(p.proj_b, w->proj_b, c.n_mel_channels))) return rc;
    if ((rc = copy(p.proj_b + c.n_mel_channels, w->gate_b, 1))) return rc;
    // persistent-decoder row set: [mel | gate | W1 . Wp] and the second prenet layer row-major
    if ((rc = copy(p.pd_proj_w, w->proj_w, (size_t)c.n_mel_channels * p.Dproj))) return rc;
    if ((rc = copy(p.pd_proj_w + (size_t)c.n_mel_channels * p.Dproj, w->gate_w, p.Dproj))) return rc;
    if ((rc = copy(p.pd_proj_b, w->proj_b, c.n_mel_channels))) return rc;
    if ((rc = copy(p.pd_proj_b + c.n_mel_channels, w->gate_b, 1))) return rc;
    hipLaunchKernelGGL(fold_prenet_proj_kernel, dim3(c.prenet_dim), dim3(256), 0, s, w->prenet_w1, w->proj_w, w->proj_b,
                       blob + p.pd_proj_w + (size_t)(c.n_mel_channels + 1) * p.Dproj, blob + p.pd_proj_b + c.n_mel_channels + 1,
                       c.prenet_dim, c.n_mel_channels, p.Dproj);
    CTTS_CHECK_LAUNCH("fold_prenet_proj");
    if ((rc = copy(p.pd_w2, w->prenet_w2, (size_t)c.prenet_dim * c.prenet_dim))) return rc;
    if (p.bg_ok) {   // batched form: MFMA tiles (sources: the copies above, already in the blob)
        const int Ra = c.attention_rnn_dim, Rd = c.decoder_rnn_dim, Rd2 = c.second_decoder_rnn_dim;
        const BgSeg att[2] = {{blob + p.att[0], p.I_att, 0, p.I_att}, {blob + p.att[1], Ra, 0, Ra}};
        const BgSeg dec[2] = {{blob + p.dec[0], p.I_dec, 0, p.I_dec}, {blob + p.dec[1], Rd, 0, Rd}};
        const BgSeg d2[2] = {{blob + p.d2[0], p.I_d2, 0, p.I_d2}, {blob + p.d2[1], Rd2, 0, Rd2}};
        const BgSeg q[1] = {{blob + p.query_w, Ra, 0, Ra}};
        const BgSeg pr[1] = {{blob + p.pd_proj_w, p.Dproj, 0, p.Dproj}};
        const BgSeg w2[1] = {{blob + p.pd_w2, c.prenet_dim, 0, c.prenet_dim}};
        if ((rc = bg_pack(blob, p.bg_att, att, 2, Ra, s))) return rc;
        if ((rc = bg_pack(blob, p.bg_dec, dec, 2, Rd, s))) return rc;
        if ((rc = bg_pack(blob, p.bg_d2, d2, 2, Rd2, s))) return rc;
        if ((rc = bg_pack(blob, p.bg_q, q, 1, 0, s))) return rc;
        if ((rc = bg_pack(blob, p.bg_proj, pr, 1, 0, s))) return rc;
        if ((rc = bg_pack(blob, p.bg_w2, w2, 1, 0, s))) return rc;
        const int n = 2 * c.location_kernel_size * c.attention_dim;
        hipLaunchKernelGGL(location_fold_kernel, dim3((n + 255) / 256), dim3(256), 0, s, blob + p.loc_conv_w, blob + p.loc_dense_w,
                           blob + p.loc_fold, c.location_n_filters, c.location_kernel_size, c.attention_dim);
        CTTS_CHECK_LAUNCH("location_fold");
    }
    const float sc[4] = {w->windowed_att_pos_offset, w->exp_smoothing_factor, 0.f, 0.f};
    CTTS_CHECK_HIP(hipMemcpyAsync(blob + p.scalars, sc, sizeof(sc), hipMemcpyHostToDevice, s));
    CTTS_CHECK_HIP(hipStreamSynchronize(s));     // `sc` is a stack temporary
    return CTTS_OK;
}

int32_t ctts_taco_decoder_max_batch(const ctts_taco_decoder_config* cfg) {
    DecPlan p;
    if (make_dec_plan(cfg, p)) return 0;
    return bg_supported(p) ? MAX_BATCH : MAX_NB;
}

size_t ctts_taco_decoder_workspace_bytes(const ctts_taco_decoder_config* cfg, int32_t batch, int32_t text_len) {
    DecPlan p; DecWs w;
    if (make_dec_plan(cfg, p) || batch < 1 || batch > (bg_supported(p) ? MAX_BATCH : MAX_NB) || text_len < 1) return 0;
    dec_carve(p, batch, text_len, nullptr, w);
    return w.total * sizeof(float);
}

int ctts_taco_decoder_init_f32(const ctts_taco_decoder_config* cfg, const void* packed, const float* memory_in,
                               const int32_t* lengths, int32_t batch, int32_t text_len, void* workspace,
                               size_t workspace_bytes, void* stream) {
    DecPlan p; DecWs w;
    int rc = make_dec_plan(cfg, p); if (rc) return rc;
    CTTS_CHECK_ARG(packed && memory_in && lengths && workspace, "decoder init: NULL pointer");
    const int max_batch = bg_supported(p) ? MAX_BATCH : MAX_NB;
    CTTS_CHECK_ARG(batch >= 1 && batch <= max_batch, "decoder init: batch=%d (1..%d built for this shape)", batch, max_batch);
    CTTS_CHECK_ARG(text_len >= 1, "decoder init: text_len=%d", text_len);
    dec_carve(p, batch, text_len, static_cast<float*>(workspace), w);
    if (w.total * sizeof(float) > workspace_bytes) {
        set_error("decoder init: workspace %zu bytes < required %zu", workspace_bytes, w.total * sizeof(float));
        return CTTS_E_WORKSPACE;
    }
    const auto& c = p.c;
    hipStream_t s = as_stream(stream);
    const float* blob = static_cast<const float*>(packed);
    CTTS_CHECK_HIP(hipMemsetAsync(workspace, 0, w.total * sizeof(float), s));
    CTTS_CHECK_HIP(hipMemcpyAsync(w.lengths, lengths, batch * sizeof(int), hipMemcpyDeviceToDevice, s));
    const int rows = batch * text_len;
    CTTS_CHECK_ARG((size_t)4 * c.memory_in_dim * sizeof(float) <= 64 * 1024, "decoder init: memory_in_dim too large");
    hipLaunchKernelGGL(linear_rows_kernel, dim3((rows + 3) / 4), dim3(256), (size_t)4 * c.memory_in_dim * sizeof(float), s,
                       memory_in, blob + p.bottleneck_wT, w.memory, rows, c.memory_in_dim, c.memory_dim);
    CTTS_CHECK_LAUNCH("memory_bottleneck");
    hipLaunchKernelGGL(linear_rows_kernel, dim3((rows + 3) / 4), dim3(256), (size_t)4 * c.memory_dim * sizeof(float), s,
                       w.memory, blob + p.memory_wT, w.pm, rows, c.memory_dim, c.attention_dim);
    CTTS_CHECK_LAUNCH("memory_layer");
    return CTTS_OK;
}

int ctts_taco_decoder_steps_f32(const ctts_taco_decoder_config* cfg, const void* packed, const uint8_t* keep_masks,
                                float* mel_out, float* gate_out, float* align_out, int32_t batch, int32_t text_len,
                                int32_t step0, int32_t n_steps, int32_t max_steps, void* workspace, void* stream) {
    return ctts_taco_decoder_steps_hidden_f32(cfg, packed, keep_masks, mel_out, gate_out, align_out, nullptr, batch, text_len, step0,
                                              n_steps, max_steps, workspace, stream);
}

int ctts_taco_decoder_steps_hidden_f32(const ctts_taco_decoder_config* cfg, const void* packed, const uint8_t* keep_masks,
                                       float* mel_out, float* gate_out, float* align_out, float* hidden_out, int32_t batch,
                                       int32_t text_len, int32_t step0, int32_t n_steps, int32_t max_steps, void* workspace,
                                       void* stream) {
    DecPlan p; DecWs w;
    int rc = make_dec_plan(cfg, p); if (rc) return rc;
    CTTS_CHECK_ARG(packed && keep_masks && mel_out && gate_out && align_out && workspace, "decoder steps: NULL pointer");
    const int max_batch = bg_supported(p) ? MAX_BATCH : MAX_NB;
    CTTS_CHECK_ARG(batch >= 1 && batch <= max_batch && step0 >= 0 && n_steps >= 0 && step0 + n_steps <= max_steps,
                   "decoder steps: batch=%d (1..%d) step0=%d n_steps=%d max_steps=%d", batch, max_batch, step0, n_steps, max_steps);
    dec_carve(p, batch, text_len, static_cast<float*>(workspace), w);
    // the batched MFMA form serves every batch it is built for: at batch <= 4 it is also ahead of the VALU kernels below (default
    // widths 64 against 88 us/step, the non-default golden shape 43 against 63: profiles/r6_05), which stay as the form for shapes
    // it does not take and, through CTTS_TACO_VALU, as an independent cross-check of both other forms
    if (batch > MAX_NB || (bg_supported(p) && !tuning().taco_valu))
        return batched_steps(p, w, static_cast<const float*>(packed), keep_masks, mel_out, gate_out, align_out, hidden_out, batch,
                             text_len, step0, n_steps, max_steps, as_stream(stream));
    const auto& c = p.c;
    const int NB = pad_batch(batch);
    hipStream_t s = as_stream(stream);
    const float* blob = static_cast<const float*>(packed);
    // Early partial sums (see LstmPart and the heterogeneous kernels above).  Column windows of the three cells:
    //   attention RNN  input [prenet | ctx | dec_h]: fresh = prenet, early = ctx, dec_h and W_hh.att_h  (behind project/prenet)
    //   decoder RNN    input [att_h | ctx]:          fresh = ctx,    early = att_h and W_hh.dec_h       (behind the attention)
    //   2nd decoder    input [dec_h]:                fresh = dec_h,  early = W_hh.d2_h                  (behind the attention)
    const int Pn = c.prenet_dim, Ra = c.attention_rnn_dim, Rd = c.decoder_rnn_dim, Rd2 = c.second_decoder_rnn_dim;
    const bool fast = c.window_range <= 16 && c.memory_dim <= ADM && c.attention_dim <= AAD && c.attention_dim % 4 == 0 &&
                      c.location_n_filters <= AF && c.location_kernel_size <= AK && c.attention_rnn_dim <= 1536;
    constexpr int R_DEC = 6, R_D2 = 8, R_ATT = 8;     // units per partial workgroup: 4 + 128 + 96 <= 256 CUs at config 5
    const bool fuse = fast && n_steps > 0 && Rd % R_DEC == 0 && Rd2 % R_D2 == 0 && Ra % R_ATT == 0 &&
                      NB * (p.I_dec + Rd) + 4 * R_DEC * NB <= ATTN_SMEM_FLOATS &&
                      NB * (p.I_d2 + Rd2) + 4 * R_D2 * NB <= ATTN_SMEM_FLOATS &&
                      NB * (p.I_att + Ra) + 4 * R_ATT * NB <= PROJ_FUSED_SMEM_FLOATS && !tuning().taco_no_fuse;
    const LstmPart att_early{nullptr, w.gp_att, Pn, p.I_att, 1}, att_fresh{w.gp_att, nullptr, 0, Pn, 0};
    const LstmPart dec_early{nullptr, w.gp_dec, 0, Ra, 1}, dec_fresh{w.gp_dec, nullptr, Ra, p.I_dec, 0};
    const LstmPart d2_early{nullptr, w.gp_d2, 0, 0, 1}, d2_fresh{w.gp_d2, nullptr, 0, p.I_d2, 0};
    if (fuse) {   // the first step of this call has nothing to hide behind: its early part runs in line
        const int cur = step0 & 1;
        rc = launch_lstm_nb(NB, blob, p.att, w.prenet, Pn, w.ctx, c.memory_dim, w.dec_h[cur], Rd, w.att_h[cur], nullptr,
                            nullptr, p.I_att, Ra, s, &att_early);
        if (rc) return rc;
    }
    for (int step = step0; step < step0 + n_steps; ++step) {
        const int cur = step & 1, nxt = cur ^ 1;    // h ping-pong: read [cur], write [nxt]
        // attention RNN on [prenet | context | decoder hidden]   (model.py:707-717)
        rc = launch_lstm_nb(NB, blob, p.att, w.prenet, Pn, w.ctx, c.memory_dim, w.dec_h[cur], Rd,
                            w.att_h[cur], w.att_h[nxt], w.att_c, p.I_att, Ra, s, fuse ? &att_fresh : nullptr);
        if (rc) return rc;
        AttnArgs a{};
        a.Wq = blob + p.query_w; a.v = blob + p.v_w; a.Wloc = blob + p.loc_conv_w; a.Wd = blob + p.loc_dense_w;
        a.scalars = blob + p.scalars;
        a.att_h = w.att_h[nxt]; a.memory = w.memory; a.pm = w.pm;
        a.w = w.w; a.cum = w.cum; a.ctx = w.ctx; a.pos = w.pos; a.align_out = align_out; a.lengths = w.lengths;
        a.T = text_len; a.A = c.attention_dim; a.Ra = c.attention_rnn_dim; a.Dm = c.memory_dim;
        a.F = c.location_n_filters; a.K = c.location_kernel_size; a.R = c.window_range;
        a.step = step; a.max_steps = max_steps;
        if (fast) {
            const int gblocks = (c.attention_dim + 7) / 8;
            if (NB == 1) hipLaunchKernelGGL(gemv_rows_kernel<1>, dim3(gblocks), dim3(256), 0, s, a.Wq, a.att_h, w.qbuf, c.attention_dim, c.attention_rnn_dim);
            else if (NB == 2) hipLaunchKernelGGL(gemv_rows_kernel<2>, dim3(gblocks), dim3(256), 0, s, a.Wq, a.att_h, w.qbuf, c.attention_dim, c.attention_rnn_dim);
            else hipLaunchKernelGGL(gemv_rows_kernel<4>, dim3(gblocks), dim3(256), 0, s, a.Wq, a.att_h, w.qbuf, c.attention_dim, c.attention_rnn_dim);
            CTTS_CHECK_LAUNCH("gemv_rows");
            if (fuse) {
                const LstmCall e1 = make_call(blob, p.dec, w.att_h[nxt], Ra, w.ctx, c.memory_dim, nullptr, 0, w.dec_h[cur],
                                              p.I_dec, Rd, NB, dec_early);
                const LstmCall e2 = make_call(blob, p.d2, w.dec_h[nxt], Rd, nullptr, 0, nullptr, 0, w.d2_h[cur], p.I_d2, Rd2,
                                              NB, d2_early);
                const int n1 = Rd / R_DEC, n2 = Rd2 / R_D2;
                const dim3 grid(batch + n1 + n2);
                if (NB == 1) hipLaunchKernelGGL((attention_partials_kernel<1, R_DEC, R_D2>), grid, dim3(256), 0, s, a, w.qbuf, batch, e1, n1, e2);
                else if (NB == 2) hipLaunchKernelGGL((attention_partials_kernel<2, R_DEC, R_D2>), grid, dim3(256), 0, s, a, w.qbuf, batch, e1, n1, e2);
                else hipLaunchKernelGGL((attention_partials_kernel<4, R_DEC, R_D2>), grid, dim3(256), 0, s, a, w.qbuf, batch, e1, n1, e2);
            } else {
                hipLaunchKernelGGL(attention_window_kernel, dim3(batch), dim3(256), 0, s, a, w.qbuf);
            }
        } else if (c.location_n_filters <= 32) {
            hipLaunchKernelGGL(attention_step_kernel<32>, dim3(batch), dim3(256), 0, s, a);
        } else {
            hipLaunchKernelGGL(attention_step_kernel<64>, dim3(batch), dim3(256), 0, s, a);
        }
        CTTS_CHECK_LAUNCH("attention_step");
        // decoder RNN on [attention hidden | context], second decoder RNN on the first's output
        rc = launch_lstm_nb(NB, blob, p.dec, w.att_h[nxt], c.attention_rnn_dim, w.ctx, c.memory_dim, nullptr, 0,
                            w.dec_h[cur], w.dec_h[nxt], w.dec_c, p.I_dec, c.decoder_rnn_dim, s, fuse ? &dec_fresh : nullptr);
        if (rc) return rc;
        rc = launch_lstm_nb(NB, blob, p.d2, w.dec_h[nxt], c.decoder_rnn_dim, nullptr, 0, nullptr, 0, w.d2_h[cur],
                            w.d2_h[nxt], w.d2_c, p.I_d2, c.second_decoder_rnn_dim, s, fuse ? &d2_fresh : nullptr);
        if (rc) return rc;
        ProjArgs q{};
        q.Wp = blob + p.proj_w; q.bp = blob + p.proj_b; q.W1T = blob + p.prenet_w1; q.W2T = blob + p.prenet_w2;
        q.dec_h = w.dec_h[nxt]; q.d2_h = w.d2_h[nxt]; q.ctx = w.ctx;
        q.keep = step + 1 < max_steps ? keep_masks + (size_t)(step + 1) * 2 * batch * c.prenet_dim : nullptr;
        q.mel_out = mel_out; q.gate_out = gate_out; q.prenet_out = w.prenet;
        q.n_mel = c.n_mel_channels; q.Rd = c.second_decoder_rnn_dim; q.Dm = c.memory_dim; q.P = c.prenet_dim;
        q.B = batch; q.step = step; q.max_steps = max_steps;
        // (splitting this stage into three multi-workgroup GEMV launches was measured: 12.8 + 5.2 + 5.2 us vs 22 us
        // here - every dependent launch costs ~5 us before its first useful byte, so fewer launches win)
        if (fuse && step + 1 < step0 + n_steps) {
            // ... and behind it, the early part of the NEXT step's attention RNN
            const LstmCall e = make_call(blob, p.att, w.prenet, Pn, w.ctx, c.memory_dim, w.dec_h[nxt], Rd, w.att_h[nxt],
                                         p.I_att, Ra, NB, att_early);
            const dim3 grid(batch + Ra / R_ATT);
            if (NB == 1) hipLaunchKernelGGL((project_partial_kernel<1, R_ATT>), grid, dim3(256), 0, s, q, batch, e);
            else if (NB == 2) hipLaunchKernelGGL((project_partial_kernel<2, R_ATT>), grid, dim3(256), 0, s, q, batch, e);
            else hipLaunchKernelGGL((project_partial_kernel<4, R_ATT>), grid, dim3(256), 0, s, q, batch, e);
        } else {
            hipLaunchKernelGGL(project_prenet_kernel, dim3(batch), dim3(256), 0, s, q);
        }
        CTTS_CHECK_LAUNCH("project_prenet");
        if (hidden_out) {
            hipLaunchKernelGGL(hidden_state_kernel, dim3(batch), dim3(256), 0, s, w.dec_h[nxt], w.d2_h[nxt], w.ctx, hidden_out, Rd2,
                               c.memory_dim, step, max_steps);
            CTTS_CHECK_LAUNCH("hidden_state");
        }
    }
    return CTTS_OK;
}

// ---- packed-sequence LSTM (encoder BiLSTM, model.py:299-309) ---------------------------------------
namespace {
struct SeqPlan { int I, H, mb, nch; size_t A, bias, whh, total; BgMat bg; bool bg_ok; };   // bg: W_hh as MFMA tiles, gate-interleaved (batch > 4)
int make_seq_plan(int I, int H, SeqPlan& p) {
    CTTS_CHECK_ARG(I >= 16 && I % 16 == 0 && H >= 4 && H % 4 == 0, "lstm_seq: input_size=%d hidden_size=%d", I, H);
    p.I = I; p.H = H;
    p.mb = (4 * H + 255) / 256; p.nch = I / 16;
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o = align_up(o + n); return r; };
    p.A = take((size_t)p.mb * p.nch * 16 * 256);
    p.bias = take((size_t)p.mb * 256);
    p.whh = take((size_t)4 * H * H);
    p.bg_ok = H % 64 == 0;
    p.bg = BgMat{0, 0, 0, 0};
    if (p.bg_ok) {
        p.bg.rows = 4 * H; p.bg.tiles = 4 * H / BG_MT; p.bg.nchunks = H / BG_KC;
        p.bg.off = take((size_t)p.bg.tiles * p.bg.nchunks * 256);
    }
    p.total = o;
    return CTTS_OK;
}
}  // namespace

size_t ctts_lstm_seq_packed_bytes(int32_t input_size, int32_t hidden_size) {
    SeqPlan p;
    if (make_seq_plan(input_size, hidden_size, p)) return 0;
    return p.total * sizeof(float);
}

int ctts_lstm_seq_pack_f32(const ctts_lstm_weights* w, int32_t input_size, int32_t hidden_size, void* packed, void* stream) {
    SeqPlan p;
    int rc = make_seq_plan(input_size, hidden_size, p); if (rc) return rc;
    CTTS_CHECK_ARG(w && w->w_ih && w->w_hh && w->b_ih && w->b_hh && packed, "lstm_seq_pack: NULL pointer");
    hipStream_t s = as_stream(stream);
    float* blob = static_cast<float*>(packed);
    rc = launch_pack_a(blob + p.A, w->w_ih, 256, p.mb, p.nch, 0, p.I, GEMM_EPI_SPLIT, 0, 4 * p.H, 0, p.I, 1, s);
    if (rc) return rc;
    rc = launch_pack_bias(blob + p.bias, 256, p.mb, w->b_ih, 0, w->b_hh, 0, GEMM_EPI_SPLIT, 0, 4 * p.H, s);
    if (rc) return rc;
    CTTS_CHECK_HIP(hipMemcpyAsync(blob + p.whh, w->w_hh, (size_t)4 * p.H * p.H * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (p.bg_ok) {
        const BgSeg seg[1] = {{blob + p.whh, p.H, 0, p.H}};
        if ((rc = bg_pack(blob, p.bg, seg, 1, p.H, s))) return rc;
    }
    return CTTS_OK;
}

size_t ctts_lstm_seq_workspace_bytes(int32_t hidden_size, int32_t batch, int32_t ld) {
    // batch <= 4: the VALU step kernels; up to MAX_BATCH where the recurrent GEMM runs on the batched MFMA form (H % 64 == 0)
    if (hidden_size < 4 || batch < 1 || batch > (hidden_size % 64 == 0 ? MAX_BATCH : MAX_NB) || ld < 4) return 0;
    const size_t NB = pad_batch(batch);
    return (align_up((size_t)batch * 4 * hidden_size * ld) + 3 * align_up(NB * hidden_size)) * sizeof(float);
}

namespace {
// One direction of a packed-sequence LSTM, ready to step: workspace carved and zeroed, input projection launched.
struct SeqDir {
    const float* whh; float *h0, *h1, *c; LstmSeq sq; int H, NB;
    const float* bg_w; BgMat bg;       // batched form (batch > 4)
};

BgArgs seq_bg_args(const SeqDir& d, const float* hold, float* hnew, int batch) {
    BgArgs a{};
    a.W = d.bg_w; a.rows = d.bg.rows; a.batch = batch;
    bg_set_x(a, d.bg, hold, d.H, nullptr, 0, nullptr, 0, nullptr, 0);
    a.c = d.c; a.h_new = hnew; a.h_old = hold; a.H = d.H; a.sq = d.sq;
    return a;
}

int seq_prepare(const void* packed, const float* x, const int32_t* lengths, int32_t reverse, float* out, int64_t out_bstride,
                int32_t out_tstride, int32_t out_col, float* hn, int32_t hn_stride, int32_t hn_col, int32_t batch, int32_t T,
                int32_t input_size, int32_t hidden_size, int32_t ld, int32_t pad, void* workspace, size_t workspace_bytes,
                hipStream_t s, SeqDir& d) {
    SeqPlan p;
    int rc = make_seq_plan(input_size, hidden_size, p); if (rc) return rc;
    CTTS_CHECK_ARG(packed && x && lengths && out && hn && workspace, "lstm_seq: NULL pointer");
    const int max_batch = p.bg_ok ? MAX_BATCH : MAX_NB;
    CTTS_CHECK_ARG(batch >= 1 && batch <= max_batch && T >= 1, "lstm_seq: batch=%d (1..%d) T=%d", batch, max_batch, T);
    const int ntiles = (T + 127) / 128;
    CTTS_CHECK_ARG(ld % 4 == 0 && ntiles * 128 + 2 * pad <= ld, "lstm_seq: geometry T=%d ld=%d pad=%d", T, ld, pad);
    const size_t need = ctts_lstm_seq_workspace_bytes(hidden_size, batch, ld);
    if (need > workspace_bytes) { set_error("lstm_seq: workspace %zu bytes < required %zu", workspace_bytes, need); return CTTS_E_WORKSPACE; }
    const float* blob = static_cast<const float*>(packed);
    const int H = p.H, NB = pad_batch(batch);
    float* xp = static_cast<float*>(workspace);
    d.h0 = xp + align_up((size_t)batch * 4 * H * ld);
    d.h1 = d.h0 + align_up((size_t)NB * H);
    d.c = d.h1 + align_up((size_t)NB * H);
    d.whh = blob + p.whh; d.H = H; d.NB = NB;
    d.bg_w = blob + p.bg.off; d.bg = p.bg;
    CTTS_CHECK_HIP(hipMemsetAsync(d.h0, 0, 3 * align_up((size_t)NB * H) * sizeof(float), s));
    // input projection for every time step: Xp[b][4H][t] = W_ih x_t + b_ih + b_hh
    GemmArgs a{};
    a.ld = ld; a.pad = pad; a.L = T; a.ntiles = ntiles; a.batch = batch; a.dst_ld = ld; a.dst_pad = pad;
    a.A = blob + p.A; a.bias = blob + p.bias;
    a.nseg = 1; a.nch_total = p.nch; a.MB = p.mb; a.M = 4 * H;
    a.seg[0] = {x, (long long)p.I * ld, p.nch, 0, 0, 0};
    a.dst0 = xp; a.dst0_bstride = (long long)4 * H * ld; a.dst1 = xp; a.dst1_bstride = a.dst0_bstride;
    a.split = p.mb * 256;
    rc = launch_gemm_f32(GEMM_EPI_SPLIT, a, s);
    if (rc) return rc;
    d.sq = LstmSeq{};
    d.sq.gadd = xp; d.sq.ga_bstride = (long long)4 * H * ld; d.sq.ga_ld = ld; d.sq.ga_pad = pad;
    d.sq.lengths = lengths; d.sq.reverse = reverse;
    d.sq.out = out; d.sq.out_bstride = out_bstride; d.sq.out_tstride = out_tstride; d.sq.out_col = out_col;
    d.sq.hn = hn; d.sq.hn_stride = hn_stride; d.sq.hn_col = hn_col;
    return CTTS_OK;
}
}  // namespace

int ctts_lstm_seq_f32(const void* packed, const float* x, const int32_t* lengths, int32_t reverse, float* out,
                      int64_t out_bstride, int32_t out_tstride, int32_t out_col, float* hn, int32_t hn_stride,
                      int32_t hn_col, int32_t batch, int32_t T, int32_t input_size, int32_t hidden_size, int32_t ld,
                      int32_t pad, void* workspace, size_t workspace_bytes, void* stream) {
    hipStream_t s = as_stream(stream);
    SeqDir d;
    int rc = seq_prepare(packed, x, lengths, reverse, out, out_bstride, out_tstride, out_col, hn, hn_stride, hn_col, batch, T,
                         input_size, hidden_size, ld, pad, workspace, workspace_bytes, s, d);
    if (rc) return rc;
    for (int step = 0; step < T; ++step) {
        d.sq.step = step;
        float* hold = (step & 1) ? d.h1 : d.h0;
        float* hnew = (step & 1) ? d.h0 : d.h1;
        if (batch > MAX_NB) {
            const BgArgs a = seq_bg_args(d, hold, hnew, batch);
            if ((rc = bg_launch_seq(a, a, 1, d.NB, s))) return rc;
            continue;
        }
        rc = launch_lstm_raw(d.NB, nullptr, d.whh, nullptr, nullptr, nullptr, 0, nullptr, 0, nullptr, 0, hold, hnew, d.c,
                             0, d.H, batch, d.sq, s);
        if (rc) return rc;
    }
    return CTTS_OK;
}

int ctts_lstm_biseq_f32(const void* packed_fwd, const void* packed_bwd, const float* x, const int32_t* lengths, float* out,
                        int64_t out_bstride, int32_t out_tstride, int32_t out_col_fwd, int32_t out_col_bwd, float* hn,
                        int32_t hn_stride, int32_t hn_col_fwd, int32_t hn_col_bwd, int32_t batch, int32_t T, int32_t input_size,
                        int32_t hidden_size, int32_t ld, int32_t pad, void* workspace_fwd, void* workspace_bwd,
                        size_t workspace_bytes, void* stream) {
    hipStream_t s = as_stream(stream);
    CTTS_CHECK_ARG(workspace_fwd != workspace_bwd, "lstm_biseq: the two directions need their own workspaces");
    SeqDir d[2];
    int rc = seq_prepare(packed_fwd, x, lengths, 0, out, out_bstride, out_tstride, out_col_fwd, hn, hn_stride, hn_col_fwd, batch,
                         T, input_size, hidden_size, ld, pad, workspace_fwd, workspace_bytes, s, d[0]);
    if (rc) return rc;
    rc = seq_prepare(packed_bwd, x, lengths, 1, out, out_bstride, out_tstride, out_col_bwd, hn, hn_stride, hn_col_bwd, batch, T,
                     input_size, hidden_size, ld, pad, workspace_bwd, workspace_bytes, s, d[1]);
    if (rc) return rc;
    const LstmPart whole{nullptr, nullptr, 0, 0, 1};
    for (int step = 0; step < T && batch > MAX_NB; ++step) {      // batched form: the recurrent product as an MFMA GEMM over the batch
        BgArgs a[2];
        for (int k = 0; k < 2; ++k) {
            d[k].sq.step = step;
            a[k] = seq_bg_args(d[k], (step & 1) ? d[k].h1 : d[k].h0, (step & 1) ? d[k].h0 : d[k].h1, batch);
        }
        if ((rc = bg_launch_seq(a[0], a[1], 2, d[0].NB, s))) return rc;
    }
    for (int step = 0; step < T && batch <= MAX_NB; ++step) {
        LstmCall q[2];
        for (int k = 0; k < 2; ++k) {
            d[k].sq.step = step;
            float* hold = (step & 1) ? d[k].h1 : d[k].h0;
            float* hnew = (step & 1) ? d[k].h0 : d[k].h1;
            q[k] = LstmCall{nullptr, d[k].whh, nullptr, nullptr, nullptr, 0, nullptr, 0, nullptr, 0, hold, hnew, d[k].c,
                            d[k].sq, whole, 0, d[k].H, batch};
        }
        switch (d[0].NB) {
            case 1: rc = launch_lstm2<1>(q[0], q[1], s); break;
            case 2: rc = launch_lstm2<2>(q[0], q[1], s); break;
            default: rc = launch_lstm2<4>(q[0], q[1], s); break;
        }
        if (rc) return rc;
    }
    return CTTS_OK;
}

}  // extern "C"
