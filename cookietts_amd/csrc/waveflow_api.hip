// WaveFlow (ax core, waveflow=True) on the fp32 MFMA conv-GEMM.
//
// The height axis (n_group rows) is autoregressive (efficient_modules.py:42-65): row r+1 of a flow
// needs the WN_2d output for row r, so a flow is n_group-1 sequential row passes, each n_layers x
// {in-layer GEMM + gate, res/skip GEMM}.  The (k_h, k_w) Conv2d with dilation (d_h, 2^i) and no
// height padding (glow_ax.py:518-523) is a GEMM whose K axis is [height tap][width tap][channel]:
// height taps read earlier rows' layer inputs from a ring of k_h slots per layer (the reference's
// audio_queues, glow_ax.py:597-602), width taps are column shifts in the padded layout.  Rows
// 0..k_h-2 simply skip the leading (all-zero) height taps via the A-chunk window - no queue
// zero-fill.  Conditioning: interp(Wc*mel + bc) == Wc*interp(mel) + bc (linear interpolation
// commutes with the 1x1 conv), so mel is interpolated ONCE to L steps and the cond layer becomes
// extra K of every in-layer GEMM (the [B, 2*C*n_layers, L] tensor is never materialised).
// PermuteHeight (efficient_modules.py:360-403) is folded into a logical->physical row map.
#include <cstdlib>
#include <algorithm>
#include <vector>

#include "gemm_f32.h"
#include "tuning.h"
#include "waveglow_kernels.h"
#include "waveflow_sep.h"
#include "waveflow_tail.h"

namespace ctts {
namespace {

constexpr size_t ALIGN_F = 64;
inline size_t align_up(size_t v) { return (v + ALIGN_F - 1) / ALIGN_F * ALIGN_F; }
constexpr int WF_BM = 128;

struct WfPlan {
    ctts_waveflow_config c;
    int C, kmel, nch_in, nch_c;      // nch_c = chunks per (tap) segment
    bool sep, precond;               // separable in-layers; conditioning handed over per flow at frame rate
    int taps;                        // GEMM taps: kh*kw, or 1 behind the depthwise stage
    std::vector<int> n_rem;          // active latent rows per flow (early outputs, ax:170-189)
    struct Flow {
        size_t start_w, start_b, end_w, end_b, winv;
        std::vector<size_t> in_A, in_b, rs_A, rs_b, rs_T, rs_Tb, dw_w, dw_b, sA1, sb1, sA2, sb2;
    };
    std::vector<Flow> fl;
    size_t total;
    int ring;                        // ring slots per layer: (kh - 1) * max height dilation + 1 rows of history
    int dh(int i) const { return c.dilation_h_l[i] > 0 ? c.dilation_h_l[i] : c.dilation_h; }
    int rs_rows(int i) const { return (i < c.n_layers - 1 && !c.merge_res_skip) ? 2 * C : C; }
    int rs_mb(int i) const { return (rs_rows(i) + WF_BM - 1) / WF_BM; }
    int in_mb() const { return (C + 63) / 64; }
    // one 128-row block holds every gate pair: the res/skip GEMM runs inside the in-layer kernel (GEMM_EPI_GATE_RS)
    bool fused() const { return C == 64 && c.gated_unit == 0; }
    // both 1x1 stages of a separable layer in one launch (waveflow_sep.hip): pointwise/gate + res/skip, C = 128
    bool sep_fused() const {
        return sep && precond && wf_sep_supported(C);
    }
};

int make_wf_plan(const ctts_waveflow_config* cfg, WfPlan& p) {
    CTTS_CHECK_ARG(cfg != nullptr, "waveflow config is NULL");
    p.c = *cfg;
    const auto& c = p.c;
    CTTS_CHECK_ARG(c.n_flows >= 1 && c.n_layers >= 1 && c.n_layers <= 12, "n_flows=%d n_layers=%d", c.n_flows, c.n_layers);
    CTTS_CHECK_ARG(gemm_mode_valid(c.f32_gemm_mode), "f32_gemm_mode=%d (CTTS_GEMM_DEFAULT / _F32 / _BF16X3 / _BF16X6)", c.f32_gemm_mode);
    CTTS_CHECK_ARG(c.n_group >= 2 && c.n_group <= 64, "n_group=%d", c.n_group);
    CTTS_CHECK_ARG(c.n_channels >= 64 && c.n_channels % 64 == 0, "n_channels=%d (multiple of 64)", c.n_channels);
    CTTS_CHECK_ARG(c.kernel_size_w % 2 == 1 && c.kernel_size_w >= 1 && c.kernel_size_h >= 1, "kernel %dx%d",
                   c.kernel_size_h, c.kernel_size_w);
    CTTS_CHECK_ARG(c.dilation_h >= 1, "dilation_h=%d", c.dilation_h);
    p.ring = 1;
    for (int i = 0; i < c.n_layers; ++i) {
        CTTS_CHECK_ARG(c.dilation_h_l[i] >= 0 && c.dilation_w[i] >= 0, "negative dilation (layer %d)", i);
        p.ring = std::max(p.ring, (c.kernel_size_h - 1) * p.dh(i) + 1);
    }
    CTTS_CHECK_ARG(c.gated_unit >= 0 && c.gated_unit < GATE_KINDS && (c.merge_res_skip == 0 || c.merge_res_skip == 1),
                   "gated_unit=%d merge_res_skip=%d", c.gated_unit, c.merge_res_skip);
    CTTS_CHECK_ARG((c.mixing == CTTS_MIX_PERMUTE || c.mixing == CTTS_MIX_CONV1X1) && (c.mix_first == 0 || c.mix_first == 1) &&
                   c.n_early_every >= 0 && c.n_early_size >= 0, "mixing=%d mix_first=%d early %d/%d", c.mixing, c.mix_first,
                   c.n_early_every, c.n_early_size);
    CTTS_CHECK_ARG(c.mixing != CTTS_MIX_PERMUTE || c.n_flows % 2 == 0 || c.n_flows == 1, "PermuteHeight requires even n_flows");
    {
        int n_rem = c.n_group;
        p.n_rem.clear();
        for (int k = 0; k < c.n_flows; ++k) {
            if (c.n_early_every > 0 && k % c.n_early_every == 0 && k > 0) n_rem -= c.n_early_size;
            CTTS_CHECK_ARG(n_rem >= 2, "flow %d has %d remaining rows (increase n_group or decrease n_early_every/n_early_size)", k, n_rem);
            p.n_rem.push_back(n_rem);
        }
    }
    p.sep = c.seperable_conv != 0 && !(c.kernel_size_h == 1 && c.kernel_size_w == 1);   // glow_ax.py:521
    p.precond = c.cond_precomputed != 0;
    p.taps = p.sep ? 1 : c.kernel_size_h * c.kernel_size_w;
    CTTS_CHECK_ARG(p.taps + 1 <= GEMM_MAX_SEG, "dense kernel %dx%d needs more than %d segments (use seperable_conv)",
                   c.kernel_size_h, c.kernel_size_w, GEMM_MAX_SEG);
    CTTS_CHECK_ARG(c.kernel_size_h <= WF_MAX_KH, "kernel_size_h=%d (<= %d)", c.kernel_size_h, WF_MAX_KH);
    CTTS_CHECK_ARG(p.precond || c.n_mel_channels >= 1, "n_mel_channels=%d", c.n_mel_channels);
    p.C = c.n_channels;
    p.kmel = p.precond ? 0 : round_up(c.n_mel_channels, GEMM_KC);
    p.nch_c = p.C / GEMM_KC;
    p.nch_in = p.taps * p.nch_c + p.kmel / GEMM_KC;
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o = align_up(o + n); return r; };
    p.fl.resize(c.n_flows);
    for (int k = 0; k < c.n_flows; ++k) {
        auto& f = p.fl[k];
        f.start_w = take(p.C); f.start_b = take(p.C);
        f.end_w = take(2 * p.C); f.end_b = take(2);
        f.winv = take(c.mixing == CTTS_MIX_CONV1X1 ? (size_t)p.n_rem[k] * p.n_rem[k] : 0);
        for (int i = 0; i < c.n_layers; ++i) {
            f.in_A.push_back(take((size_t)p.in_mb() * p.nch_in * GEMM_KC * WF_BM));
            f.in_b.push_back(take((size_t)p.in_mb() * WF_BM));
            f.rs_A.push_back(take((size_t)p.rs_mb(i) * p.nch_c * GEMM_KC * WF_BM));
            f.rs_b.push_back(take((size_t)p.rs_mb(i) * WF_BM));
            f.rs_T.push_back(take(p.fused() ? 64 * 128 : 0));
            f.rs_Tb.push_back(take(p.fused() ? 128 : 0));
            f.dw_w.push_back(take(p.sep ? (size_t)p.C * c.kernel_size_h * c.kernel_size_w : 0));
            f.dw_b.push_back(take(p.sep ? p.C : 0));
            f.sA1.push_back(take(p.sep_fused() ? 128 * 256 : 0));
            f.sb1.push_back(take(p.sep_fused() ? 256 : 0));
            f.sA2.push_back(take(p.sep_fused() ? 128 * 256 : 0));
            f.sb2.push_back(take(p.sep_fused() ? 256 : 0));
        }
    }
    p.total = o;
    return CTTS_OK;
}

// res/skip weight [rows][64] -> transposed, row-padded [64][128] (+ bias [128]) for the fused epilogue
__global__ void wf_pack_rs_t_kernel(const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ wT,
                                    float* __restrict__ bT, int rows) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < 64 * 128) {
        const int ch = i >> 7, row = i & 127;
        wT[i] = row < rows ? w[row * 64 + ch] : 0.0f;
    } else if (i < 64 * 128 + 128) {
        const int row = i - 64 * 128;
        bT[row] = row < rows ? b[row] : 0.0f;
    }
}

struct WfGeom { int L, Lr, ld, pad, ntiles; };   // Lr: row stride of the dense `rows` buffer (16-byte rows)

int make_wf_geom(const WfPlan& p, int samples, WfGeom& g) {
    CTTS_CHECK_ARG(samples >= p.c.n_group && samples % p.c.n_group == 0, "samples=%d not a multiple of n_group=%d",
                   samples, p.c.n_group);
    g.L = samples / p.c.n_group;
    g.Lr = round_up(g.L, 4);
    int maxshift = 0;
    for (int i = 0; i < p.c.n_layers; ++i)
        maxshift = std::max(maxshift, (p.c.kernel_size_w / 2) * (p.c.dilation_w[i] > 0 ? p.c.dilation_w[i] : 1 << i));
    g.pad = round_up(maxshift > 128 ? maxshift : 128, 32);
    const int bn = gemm_bn(WF_BM);
    g.ntiles = (g.L + bn - 1) / bn;
    g.ld = g.ntiles * bn + 2 * g.pad;
    return CTTS_OK;
}

// Row queue (wf_row_persistent_kernel): control words + the layer descriptors of one flow, inside the caller's workspace
struct WfQueueWs {
    unsigned int *abort_word, *counters, *flags;     // [1] | [n_flows * n_group] | [n_group * (n_layers + 1) stages][batch][tiles of 64 columns]
    unsigned int* status;                            // sticky: WF_ABORT_MAGIC once a call on this workspace aborted, until it is reported
    GemmArgs* layers;                                // [n_group][n_layers] of the flow being run
    WfTailDesc* tails;                               // [n_group]
    size_t control_bytes;                            // abort word .. end of the flags: zeroed at the start of every call
};
struct WfWs { float *rows, *mel_up, *cond_up, *dwout, *X, *act, *out; size_t total, xslot, cond_slot; WfQueueWs q; };

void wf_carve(const WfPlan& p, const WfGeom& g, int batch, float* base, WfWs& w) {
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o = align_up(o + n); return base ? base + r : nullptr; };
    const size_t B = batch;
    w.rows = take(B * p.c.n_group * g.Lr);
    w.mel_up = take(B * p.kmel * g.ld);
    w.cond_slot = align_up(B * 2 * p.C * g.ld);              // upsampled conditioning of one layer
    w.cond_up = take(p.precond ? w.cond_slot * p.c.n_layers : 0);
    w.dwout = take(p.sep ? B * p.C * g.ld : 0);
    w.xslot = align_up(B * p.C * g.ld);
    w.X = take(w.xslot * p.c.n_layers * p.ring);
    w.act = take(B * p.C * g.ld);
    w.out = take(B * p.C * g.ld);
    {
        const size_t n_count = (size_t)p.c.n_flows * p.c.n_group, n_flags = (size_t)p.c.n_group * (p.c.n_layers + 1) * B * wf_row_tiles(g.L, 1);
        const size_t control = align_up(ALIGN_F + n_count + n_flags);                       // in 4-byte words
        float* c = take(control);
        w.q.abort_word = reinterpret_cast<unsigned int*>(c);
        w.q.counters = c ? w.q.abort_word + ALIGN_F : nullptr;
        w.q.flags = c ? w.q.counters + n_count : nullptr;
        w.q.control_bytes = control * sizeof(float);
        w.q.layers = reinterpret_cast<GemmArgs*>(take(((size_t)p.c.n_group * p.c.n_layers * sizeof(GemmArgs) + 3) / 4));
        w.q.tails = reinterpret_cast<WfTailDesc*>(take(((size_t)p.c.n_group * sizeof(WfTailDesc) + 3) / 4));
        w.q.status = reinterpret_cast<unsigned int*>(take(ALIGN_F));   // outside the per-call memset
    }
    w.total = o;
}

// rows[b][g][l] = z[b][G*l + g]   (efficient_model_ax.py:310)
__global__ __launch_bounds__(256) void wf_squeeze_kernel(const float* __restrict__ z, float* __restrict__ rows,
                                                         int G, int L, int Lr) {
    const int l = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (l >= Lr) return;
    const float* zb = z + (size_t)b * G * L + (size_t)l * G;
    float* rb = rows + (size_t)b * G * Lr + l;
    for (int g = 0; g < G; ++g) rb[(size_t)g * Lr] = l < L ? zb[g] : 0.f;   // row tail (l >= L) stays zero
}

struct RowMap { int phys[64]; };

// audio[b][G*l + g] = rows[b][phys[g]][l]   (ax:346, with the accumulated PermuteHeight map)
__global__ __launch_bounds__(256) void wf_unsqueeze_kernel(const float* __restrict__ rows, float* __restrict__ audio,
                                                           int G, int L, int Lr, RowMap map) {
    const int l = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (l >= L) return;
    const float* rb = rows + (size_t)b * G * Lr + l;
    float* ab = audio + (size_t)b * G * L + (size_t)l * G;
    for (int g = 0; g < G; ++g) ab[g] = rb[(size_t)map.phys[g] * Lr];
}

// rows: NaN -> 0 in place   (ax:13-16, 333-334)
__global__ __launch_bounds__(256) void wf_nan_to_zero_kernel(float* __restrict__ rows, size_t n4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    float4 v = reinterpret_cast<float4*>(rows)[i];
    v.x = (v.x != v.x) ? 0.f : v.x; v.y = (v.y != v.y) ? 0.f : v.y;
    v.z = (v.z != v.z) ? 0.f : v.z; v.w = (v.w != v.w) ? 0.f : v.w;
    reinterpret_cast<float4*>(rows)[i] = v;
}

// mel_up[b][m][pad + l] = linear interpolation (align_corners=True) of mel[b][m][:] to L steps,
// fp32 exactly like ATen's upsample_linear1d (glow_ax.py:545-554)
__global__ __launch_bounds__(256) void wf_interp_kernel(const float* __restrict__ mel, float* __restrict__ up,
                                                        int n_mel, int kmel, int F, int L, int ld, int pad) {
    const int l = blockIdx.x * 256 + threadIdx.x;
    const int m = blockIdx.y, b = blockIdx.z;
    if (l >= L) return;
    float v;
    const float* src = mel + ((size_t)b * n_mel + m) * F;
    if (F == L) {
        v = src[l];
    } else {
        const float scale = L > 1 ? (float)(F - 1) / (float)(L - 1) : 0.f;
        const float real = scale * (float)l;
        const int i0 = (int)real;
        const int i1 = i0 + 1 < F ? i0 + 1 : F - 1;
        const float l1 = real - (float)i0;
        const float l0 = 1.0f - l1;
        v = gemm_lerp(l0, src[i0], l1, src[i1]);   // explicit contraction: every interpolation in the library rounds the same way
    }
    up[((size_t)b * kmel + m) * ld + pad + l] = v;
}

// X0[b][c][pad + l] = ws[c] * rows[b][row][l] + bs[c]   (Conv2d(1->C, 1x1), glow_ax.py:558)
__global__ __launch_bounds__(256) void wf_start_kernel(const float* __restrict__ rows, const float* __restrict__ ws,
                                                       const float* __restrict__ bs, float* __restrict__ x, int C,
                                                       int G, int row, int L, int Lr, int ld, int pad) {
    const int n = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int b = blockIdx.z;
    if (n >= L) return;
    const float4 a = *reinterpret_cast<const float4*>(rows + ((size_t)b * G + row) * Lr + n);
    const int c0 = blockIdx.y * 16;
    float* xb = x + (size_t)b * C * ld + pad + n;
    const bool k1 = n + 1 < L, k2 = n + 2 < L, k3 = n + 3 < L;      // columns >= L are halo: keep them zero
    for (int c = c0; c < c0 + 16 && c < C; ++c) {
        const float w = ws[c], bias = bs[c];
        float4 v;
        v.x = wf_start_value(w, a.x, bias);
        v.y = k1 ? wf_start_value(w, a.y, bias) : 0.f; v.z = k2 ? wf_start_value(w, a.z, bias) : 0.f; v.w = k3 ? wf_start_value(w, a.w, bias) : 0.f;
        *reinterpret_cast<float4*>(xb + (size_t)c * ld) = v;
    }
}

// e = Wend * out + bend; log_s = e[0], t = e[1]; rows[b][row][l] = (rows[b][row][l] - t) / exp(log_s)
// (glow_ax.py:628, efficient_modules.py:61-62).  4 waves x 256 steps, each wave reduces C/4 channels.
__global__ __launch_bounds__(256) void wf_tail_kernel(const float* __restrict__ out, float* __restrict__ rows,
                                                      const float* __restrict__ Wend, const float* __restrict__ bend,
                                                      int C, int G, int row, int L, int Lr, int ld, int pad) {
    __shared__ __attribute__((aligned(16))) float part[3][2][256];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int n = blockIdx.x * 256 + lane * 4;
    float4 e0 = make_float4(0.f, 0.f, 0.f, 0.f), e1 = e0;
    const int cq = C / 4;
    const int cbeg = __builtin_amdgcn_readfirstlane(wv * cq);
    const float* ob = out + (size_t)b * C * ld + pad + n;
    for (int c = cbeg; c < cbeg + cq; ++c) {
        const float4 v = *reinterpret_cast<const float4*>(ob + (size_t)c * ld);
        const float w0 = Wend[c], w1 = Wend[C + c];
        e0.x = wf_end_fma(w0, v.x, e0.x); e0.y = wf_end_fma(w0, v.y, e0.y); e0.z = wf_end_fma(w0, v.z, e0.z); e0.w = wf_end_fma(w0, v.w, e0.w);
        e1.x = wf_end_fma(w1, v.x, e1.x); e1.y = wf_end_fma(w1, v.y, e1.y); e1.z = wf_end_fma(w1, v.z, e1.z); e1.w = wf_end_fma(w1, v.w, e1.w);
    }
    if (wv > 0) {
        *reinterpret_cast<float4*>(&part[wv - 1][0][lane * 4]) = e0;
        *reinterpret_cast<float4*>(&part[wv - 1][1][lane * 4]) = e1;
    }
    __syncthreads();
    if (wv != 0 || n >= L) return;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const float4 p0 = *reinterpret_cast<const float4*>(&part[q][0][lane * 4]);
        const float4 p1 = *reinterpret_cast<const float4*>(&part[q][1][lane * 4]);
        e0.x += p0.x; e0.y += p0.y; e0.z += p0.z; e0.w += p0.w;
        e1.x += p1.x; e1.y += p1.y; e1.z += p1.z; e1.w += p1.w;
    }
    const float b0 = bend[0], b1 = bend[1];
    float* rp = rows + ((size_t)b * G + row) * Lr + n;
    float4 a = *reinterpret_cast<const float4*>(rp);
    a.x = wf_row_update(a.x, e0.x, e1.x, b0, b1);
    a.y = n + 1 < L ? wf_row_update(a.y, e0.y, e1.y, b0, b1) : 0.f;       // row tail stays zero
    a.z = n + 2 < L ? wf_row_update(a.z, e0.z, e1.z, b0, b1) : 0.f;
    a.w = n + 3 < L ? wf_row_update(a.w, e0.w, e1.w, b0, b1) : 0.f;
    *reinterpret_cast<float4*>(rp) = a;
}

// Upsampled conditioning of every layer of one flow (glow_ax.py:545-554, 564-592): frames [B][2C*n_layers] rows of a
// padded frame-rate tensor -> up[layer][b][2C][pad + l], linear interpolation with align_corners=True, fp32 like ATen.
__global__ __launch_bounds__(256) void wf_interp_cond_kernel(const float* __restrict__ frames, float* __restrict__ up,
                                                             int rows2c, int n_layers, int F, int f_ld, int f_pad,
                                                             int L, int ld, int pad, size_t slot) {
    const int l = blockIdx.x * 256 + threadIdx.x;
    const int ch = blockIdx.y, b = blockIdx.z;                 // ch in [0, 2C*n_layers)
    if (l >= L) return;
    const float* src = frames + ((size_t)b * rows2c * n_layers + ch) * f_ld + f_pad;
    float v;
    if (F == L) {
        v = src[l];
    } else {
        const float scale = L > 1 ? (float)(F - 1) / (float)(L - 1) : 0.f;
        const float real = scale * (float)l;
        const int i0 = (int)real;
        const int i1 = i0 + 1 < F ? i0 + 1 : F - 1;
        const float l1 = real - (float)i0;
        const float l0 = 1.0f - l1;
        v = gemm_lerp(l0, src[i0], l1, src[i1]);   // explicit contraction: every interpolation in the library rounds the same way
    }
    const int layer = ch / rows2c, r = ch % rows2c;
    up[(size_t)layer * slot + ((size_t)b * rows2c + r) * ld + pad + l] = v;
}


// Four time steps per thread, one 16-byte store (pad and ld multiples of 4): the scalar form writes its 0.8 GB per flow of the
// author's model with 4-byte stores at 1.6 TB/s.  Same arithmetic per element.
__global__ __launch_bounds__(256) void wf_interp_cond_vec_kernel(const float* __restrict__ frames, float* __restrict__ up,
                                                                 int rows2c, int n_layers, int F, int f_ld, int f_pad,
                                                                 int L, int ld, int pad, size_t slot) {
    const int l4 = 4 * (blockIdx.x * 256 + threadIdx.x);
    const int ch = blockIdx.y, b = blockIdx.z;
    if (l4 >= L) return;
    const float* src = frames + ((size_t)b * rows2c * n_layers + ch) * f_ld + f_pad;
    const float scale = L > 1 ? (float)(F - 1) / (float)(L - 1) : 0.f;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int l = min(l4 + j, L - 1);
        if (F == L) {
            v[j] = src[l];
        } else {
            const float real = scale * (float)l;
            const int i0 = (int)real;
            const int i1 = i0 + 1 < F ? i0 + 1 : F - 1;
            const float l1 = real - (float)i0;
            const float l0 = 1.0f - l1;
            v[j] = gemm_lerp(l0, src[i0], l1, src[i1]);
        }
    }
    const int layer = ch / rows2c, r = ch % rows2c;
    float* dst = up + (size_t)layer * slot + ((size_t)b * rows2c + r) * ld + pad + l4;
    if (l4 + 3 < L) {
        *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (l4 + j < L) dst[j] = v[j];
    }
}


// Depthwise stage of a separable in-layer (glow_ax.py:525-527: Conv2d(C, C, (kh, kw), groups=C), width dilation dw,
// causal in height): y[b][c][l] = bias[c] + sum_{a >= a_min} sum_j w[c][a][j] * x_a[b][c][l + (j - kw/2) * dw], where
// x_a is the ring slot of height tap a (taps a < a_min reach above the first row: zeros, skipped).
__global__ __launch_bounds__(256) void wf_depthwise_kernel(WfSlots x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ y, int C,
                                                           int kh, int kw, int dw, int a_min, int L, int ld, int pad) {
    const int l = blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y, b = blockIdx.z;
    if (l >= L) return;
    const size_t row = ((size_t)b * C + c) * ld + pad + l;
    const float* wc = w + (size_t)c * kh * kw;
    float acc = bias[c];
    for (int a = a_min; a < kh; ++a) {
        const float* xa = x.p[a] + row;
        for (int j = 0; j < kw; ++j) acc = fmaf(wc[a * kw + j], xa[(j - kw / 2) * dw], acc);
    }
    y[row] = acc;
}

// x[b][row0 + e][pad + t] = table[ids[b]][e]   (speaker embedding repeated over frames and concatenated, ax:286-291)
__global__ __launch_bounds__(256) void embed_rows_kernel(const float* __restrict__ table, const int64_t* __restrict__ ids,
                                                         float* __restrict__ x, int row0, int E, int C, int T, int ld,
                                                         int pad) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int e = blockIdx.y, b = blockIdx.z;
    if (t >= T) return;
    const long long id = ids[b];          // an id outside the table poisons the utterance (NaN) instead of reading out of bounds
    x[((size_t)b * C + row0 + e) * ld + pad + t] = (id >= 0 && id < CTTS_N_SPEAKERS) ? table[(size_t)id * E + e] : __builtin_nanf("");
}

// y = alpha * x + r on the valid columns (rezero + residual of the model-level cond stack, ax:299-307)
__global__ __launch_bounds__(256) void scale_add_rows_kernel(const float* __restrict__ x, const float* __restrict__ alpha,
                                                             const float* __restrict__ r, float* __restrict__ y, int C,
                                                             int T, int ld, int pad) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y, b = blockIdx.z;
    if (t >= T) return;
    const size_t i = ((size_t)b * C + c) * ld + pad + t;
    const float a = alpha ? alpha[0] : 1.0f;
    y[i] = r ? a * x[i] + r[i] : a * x[i];
}

// x = (x + shift) * scale on the valid columns (shift_spect / scale_spect, ax:206-209, 281-284); the halo stays zero
__global__ __launch_bounds__(256) void affine_rows_kernel(float* __restrict__ x, int C, int T, int ld, int pad, float shift,
                                                          float scale) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y, b = blockIdx.z;
    if (t >= T) return;
    const size_t i = ((size_t)b * C + c) * ld + pad + t;
    x[i] = (x[i] + shift) * scale;
}

// F.interpolate along time on padded rows, the arithmetic of ATen's upsample kernels:
//   mode 0 = 'linear', align_corners=True  (glow_ax.py:365 / ax:174): src = n * (Tin - 1) / (Tout - 1)
//   mode 1 = 'linear', align_corners=False (TransposedUpsampleNet residual, glow_ax.py:231): src = max((n + .5) * s - .5, 0)
//   mode 2 = 'nearest'                                                          : src = min(floor(n * s), Tin - 1)
// with s = 1 / scale_factor when a scale factor was given (ATen uses the reciprocal of the given factor), else Tin / Tout.
__global__ __launch_bounds__(256) void resample_rows_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int Tin,
                                                            int ld_in, int pad_in, int Tout, int ld_out, int pad_out,
                                                            int mode, float s) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y, b = blockIdx.z;
    if (n >= Tout) return;
    const float* src = x + ((size_t)b * C + c) * ld_in + pad_in;
    float v;
    if (mode == 2) {
        const int i = min((int)floorf((float)n * s), Tin - 1);
        v = src[i];
    } else {
        float real;
        if (mode == 0) real = (Tout > 1 ? (float)(Tin - 1) / (float)(Tout - 1) : 0.f) * (float)n;
        else real = fmaxf(s * ((float)n + 0.5f) - 0.5f, 0.f);
        const int i0 = (int)real;
        const int i1 = i0 + 1 < Tin ? i0 + 1 : Tin - 1;
        const float l1 = real - (float)i0;
        const float l0 = 1.0f - l1;
        v = gemm_lerp(l0, src[i0], l1, src[i1]);   // explicit contraction: every interpolation in the library rounds the same way
    }
    y[((size_t)b * C + c) * ld_out + pad_out + n] = v;
}

// ConvTranspose1d(stride s, padding p) as s stride-1 convolutions (one per output residue r = (n + p) mod s, evaluated
// by ctts_conv1d_f32 over the INPUT positions m): out[n] = phase[r][(n + p) / s].  phases: [s][B][C][ld_in].
__global__ __launch_bounds__(256) void interleave_phases_kernel(const float* __restrict__ ph, float* __restrict__ y, int C,
                                                                int s, int p, int Tin, int ld_in, int pad_in, int Tout,
                                                                int ld_out, int pad_out, size_t phase_stride) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y, b = blockIdx.z;
    if (n >= Tout) return;
    const int q = n + p, r = q % s, m = q / s;
    float v = 0.f;
    if (m < Tin) v = ph[(size_t)r * phase_stride + ((size_t)b * C + c) * ld_in + pad_in + m];
    y[((size_t)b * C + c) * ld_out + pad_out + n] = v;
}

// inverse of the "perceived volume" companding (efficient_model_ax.py:342-344): z > 0 -> 10^log2(z), z < 0 -> -(10^log2(-z))
__global__ __launch_bounds__(256) void vol_unscale_kernel(float* __restrict__ x, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    if (v > 0.f) x[i] = powf(10.0f, log2f(v));
    else if (v < 0.f) x[i] = -powf(10.0f, log2f(-v));
}

// y[n] = x[n] + p * y[n-1], one workgroup per utterance, fp64 state (scipy.signal.lfilter([1],[1,-p]) runs in
// float64 on the reference's CPU path, ax:351-355).  Thread t owns one contiguous span: pass 1 runs the recurrence
// from a zero state to get the span's own contribution, thread 0 chains the 256 span ends
// (carry_out = end + p^len * carry_in), pass 2 re-runs each span from its true incoming state.
// x and y may alias (the host wrapper filters in place): no __restrict__; every thread reads its whole span in pass 1
// before any thread writes (barrier), and pass 2 reads element n before writing element n of its own span only.
__global__ __launch_bounds__(256) void deemphasis_kernel(const float* x, float* y, int T, double p) {
    __shared__ double ends[256], gains[256], carry[256];
    const int b = blockIdx.x, t = threadIdx.x;
    const int span = (T + 255) / 256;
    const int n0 = min(t * span, T), n1 = min(n0 + span, T);
    const float* xb = x + (size_t)b * T;
    float* yb = y + (size_t)b * T;
    double s = 0.0, gpow = 1.0;
    for (int n = n0; n < n1; ++n) { s = (double)xb[n] + p * s; gpow *= p; }
    ends[t] = s; gains[t] = gpow;
    __syncthreads();
    if (t == 0) {
        double c = 0.0;
        for (int i = 0; i < 256; ++i) { carry[i] = c; c = ends[i] + gains[i] * c; }
    }
    __syncthreads();
    s = carry[t];
    for (int n = n0; n < n1; ++n) { s = (double)xb[n] + p * s; yb[n] = (float)s; }
}

// Vector form: a thread owns 4 consecutive columns (16-byte aligned) of one channel.  DW = 0 stands for any width
// dilation that is a multiple of 4: every tap is then ONE aligned 16-byte load for the 4 columns; for DW = 1, 2 the
// taps of the 4 columns overlap, so an aligned window [l - 4*HW, l + 4 + 4*HW) is loaded once per height tap and the
// taps are picked out of registers (compile-time indices).
template <int KW, int DW>
__global__ __launch_bounds__(256) void wf_depthwise_vec_kernel(WfSlots x, const float* __restrict__ w,
                                                               const float* __restrict__ bias, float* __restrict__ y,
                                                               int C, int kh, int dw, int a_min, int L, int ld, int pad) {
    const int l = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int c = blockIdx.y, b = blockIdx.z;
    if (l >= L) return;
    const size_t row = ((size_t)b * C + c) * ld + pad + l;
    const float* wc = w + (size_t)c * kh * KW;
    const float bc = bias[c];
    float acc[4] = {bc, bc, bc, bc};
    for (int a = a_min; a < kh; ++a) {
        const float* xa = x.p[a] + row;
        if constexpr (DW == 0) {
#pragma unroll
            for (int j = 0; j < KW; ++j) {
                const float4 v = *reinterpret_cast<const float4*>(xa + (j - KW / 2) * dw);
                const float wj = wc[a * KW + j];
                acc[0] = fmaf(wj, v.x, acc[0]); acc[1] = fmaf(wj, v.y, acc[1]);
                acc[2] = fmaf(wj, v.z, acc[2]); acc[3] = fmaf(wj, v.w, acc[3]);
            }
        } else {
            constexpr int HW = ((KW / 2) * DW + 3) / 4;          // float4s on each side of the centre one
            float win[4 * (2 * HW + 1)];
#pragma unroll
            for (int q = 0; q < 2 * HW + 1; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(xa + 4 * (q - HW));
                win[4 * q] = v.x; win[4 * q + 1] = v.y; win[4 * q + 2] = v.z; win[4 * q + 3] = v.w;
            }
#pragma unroll
            for (int j = 0; j < KW; ++j) {
                const float wj = wc[a * KW + j];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = fmaf(wj, win[4 * HW + e + (j - KW / 2) * DW], acc[e]);
            }
        }
    }
    // columns >= L are halo and must stay zero
    float4 o;
    o.x = acc[0];
    o.y = l + 1 < L ? acc[1] : 0.f; o.z = l + 2 < L ? acc[2] : 0.f; o.w = l + 3 < L ? acc[3] : 0.f;
    *reinterpret_cast<float4*>(y + row) = o;
}

template <int KW>
void launch_depthwise_vec(dim3 grid, hipStream_t s, const WfSlots& xs, const float* w, const float* b, float* y, int C,
                          int kh, int dw, int a_min, int L, int ld, int pad) {
    if (dw == 1) hipLaunchKernelGGL((wf_depthwise_vec_kernel<KW, 1>), grid, dim3(256), 0, s, xs, w, b, y, C, kh, dw, a_min, L, ld, pad);
    else if (dw == 2) hipLaunchKernelGGL((wf_depthwise_vec_kernel<KW, 2>), grid, dim3(256), 0, s, xs, w, b, y, C, kh, dw, a_min, L, ld, pad);
    else hipLaunchKernelGGL((wf_depthwise_vec_kernel<KW, 0>), grid, dim3(256), 0, s, xs, w, b, y, C, kh, dw, a_min, L, ld, pad);
}

// InvertibleConv1x1.inverse over the latent rows (efficient_modules.py:269-286): new[i] = sum_j Winv[i][j] old[j] for
// the n active logical rows, in place (a thread owns its columns), rows addressed through the logical -> physical map
__global__ __launch_bounds__(256) void wf_mix_kernel(float* __restrict__ rows, const float* __restrict__ Winv, RowMap map,
                                                     int n, int G, int Lr) {
    __shared__ float sW[64 * 64];
    for (int i = threadIdx.x; i < n * n; i += 256) sW[i] = Winv[i];
    __syncthreads();
    const int l = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (l >= Lr) return;
    float v[64];
    float* base = rows + (size_t)b * G * Lr + l;
#pragma unroll 1
    for (int j = 0; j < n; ++j) v[j] = base[(size_t)map.phys[j] * Lr];
#pragma unroll 1
    for (int i = 0; i < n; ++i) {
        float acc = 0.f;
#pragma unroll 1
        for (int j = 0; j < n; ++j) acc = fmaf(sW[i * n + j], v[j], acc);
        base[(size_t)map.phys[i] * Lr] = acc;
    }
}

void wf_permutation(int k, int G, int* perm) {
    for (int g = 0; g < G; ++g) perm[g] = G - 1 - g;                       // reverse (k % 4 in {0,1})
    if (k % 4 == 2 || k % 4 == 3) {                                        // reverse each half separately
        const int half = G / 2;
        for (int g = 0; g < half; ++g) perm[g] = half - 1 - g;
        for (int g = half; g < G; ++g) perm[g] = G - 1 - (g - half);
    }
}

}  // namespace
}  // namespace ctts

using namespace ctts;

namespace {

// Shared body of the two entry points.  cond = mel [B][n_mel][frames] (dense) when the single linear cond layer is
// folded into the GEMM, or the per-flow frame-rate conditioning [n_flows][B][2C*n_layers][cond_ld] (padded rows).
// ---- region split of a fused-layer launch (see the runner) ----------------------------------------------------------
constexpr int WF_TILE = 256;          // columns per region unit: every launch shape's tile width divides it
#ifndef WF_SPLIT_MIN_BLOCKS
#define WF_SPLIT_MIN_BLOCKS 256
#endif
#ifndef WF_NBIG
#define WF_NBIG 2                     // big regions per layer (2: A | M | B on three streams)
#endif
constexpr int WF_NREG = 2 * WF_NBIG - 1;     // regions incl. the one-tile separators: big 0, sep 0, big 1, sep 1, ...
// The row queue is taken from this many 128-column items per layer on: where the per-layer launches stop being the eight-wave
// split-K tile (gemm_f32_small.hip gate_rs_splitk8_tile, up to GATE_RS_SPLITK_MAX_BLOCKS = 256 items).  Config 4, round 5: batch 1
// (113 items) 30.9 ms per layer against 40.6 queued; batch 2 (226 items) 52.0 against 55.7 (until the eight-wave tile the queue was
// 2 % ahead there: 56.9 against 58.3 ms, and the bound was 200); batch 3 (339 items) 74.4 queued against 92.9.
#ifndef WF_ROW_QUEUE_MIN_ITEMS
#define WF_ROW_QUEUE_MIN_ITEMS 257
#endif
#ifndef WF_ROW_QUEUE_MAX_ITEMS
#define WF_ROW_QUEUE_MAX_ITEMS 1250
#endif
#ifndef WF_ROW_QUEUE_SPLITK_BELOW
#define WF_ROW_QUEUE_SPLITK_BELOW 400
#endif
// audio[b][:] = NaN when the row queue's abort word is set, and the workspace's sticky status word says so until a later
// call reports it (CTTS_E_ABORT): the NaN is the in-stream marker, the status the one a caller can act on
constexpr unsigned int WF_ABORT_MAGIC = 0xAB0F7001u;       // a value, not a flag: the workspace may start out as anything
__global__ void wf_abort_poison_kernel(const unsigned int* __restrict__ abort_word, float* __restrict__ audio, long long n,
                                       unsigned int* __restrict__ status) {
    if (*abort_word == 0) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) *status = WF_ABORT_MAGIC;
    float* a = audio + (size_t)blockIdx.x * n;
    for (long long i = threadIdx.x; i < n; i += 256) a[i] = __builtin_nanf("");
}

// Host side of the row queue's descriptors: a PINNED buffer per host thread, so that the per-flow copies are ordinary stream-ordered
// DMA whatever the runtime does with pageable sources; the buffer is rewritten by the thread's next call only after the event
// recorded behind this call's last copy (normally long past: the wait is a formality).  Never freed (as the helper streams below).
struct WfDescHost {
    char* host = nullptr;                                  // bytes
    size_t cap = 0;
    hipEvent_t done = nullptr;
    bool pending = false, in_call = false;
    int begin(size_t n) {
        if (in_call) CTTS_CHECK_HIP(hipDeviceSynchronize());     // the previous call failed half-way: its copies may still be queued
        in_call = true;
        if (pending) { CTTS_CHECK_HIP(hipEventSynchronize(done)); pending = false; }
        if (!done) CTTS_CHECK_HIP(hipEventCreateWithFlags(&done, hipEventDisableTiming));
        if (n > cap) {
            if (host) CTTS_CHECK_HIP(hipHostFree(host));
            host = nullptr; cap = 0;
            CTTS_CHECK_HIP(hipHostMalloc(reinterpret_cast<void**>(&host), n, hipHostMallocDefault));
            cap = n;
        }
        return CTTS_OK;
    }
    int end(hipStream_t s) {
        CTTS_CHECK_HIP(hipEventRecord(done, s));
        pending = true;
        in_call = false;
        return CTTS_OK;
    }
};
thread_local WfDescHost t_wf_desc;

struct WfRegionStreams {              // per host thread: one helper stream per region but the first + double-buffered events
    hipStream_t st[WF_NREG] = {};
    hipEvent_t fork = nullptr, ev[WF_NREG][2] = {};
    bool ready = false;
    int init() {
        if (ready) return CTTS_OK;
        for (int k = 1; k < WF_NREG; ++k) CTTS_CHECK_HIP(hipStreamCreateWithFlags(&st[k], hipStreamNonBlocking));
        CTTS_CHECK_HIP(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
        for (int k = 0; k < WF_NREG; ++k)
            for (int q = 0; q < 2; ++q) CTTS_CHECK_HIP(hipEventCreateWithFlags(&ev[k][q], hipEventDisableTiming));
        ready = true;
        return CTTS_OK;
    }
};
thread_local WfRegionStreams t_wf_streams;

struct WfRegionSplit {
    bool on = false;
    int ntiles = 0, L = 0;
    int tile0[WF_NREG + 1] = {};
    void plan() {       // big regions of (ntiles - separators) / WF_NBIG tiles, one-tile separators between them
        const int big = (ntiles - (WF_NBIG - 1)) / WF_NBIG;
        int t = 0;
        for (int k = 0; k < WF_NREG; ++k) { tile0[k] = t; t += (k & 1) ? 1 : big; }
        tile0[WF_NREG] = ntiles;          // the last big region takes the remainder
    }
    static GemmArgs shifted(const GemmArgs& a, int tile0, int nt, int L) {
        GemmArgs r = a;
        const long long co = (long long)tile0 * WF_TILE;
        for (int j = 0; j < r.nseg; ++j) r.seg[j].base += co;
        if (r.dst0) r.dst0 += co;
        if (r.dst1) r.dst1 += co;
        if (r.src0) r.src0 += co;
        if (r.addend) r.addend += co;
        const int cols = L - tile0 * WF_TILE;
        r.L = cols < nt * WF_TILE ? cols : nt * WF_TILE;
        r.ntiles = nt;
        r.shape_blocks = 255;       // every region on the 128 x 128 shape (same K order as the 128 x 256 one; never the split-K shape)
        return r;
    }
    int begin_row(hipStream_t s) {
        auto& t = t_wf_streams;
        int rc = t.init(); if (rc) return rc;
        plan();
        CTTS_CHECK_HIP(hipEventRecord(t.fork, s));                      // the row's start kernel has written every column
        for (int k = 1; k < WF_NREG; ++k) CTTS_CHECK_HIP(hipStreamWaitEvent(t.st[k], t.fork, 0));
        return CTTS_OK;
    }
    // region k of layer i waits for its neighbours' layer i - 1 (its own layer i - 1 precedes it on its stream)
    int layer(const GemmArgs& a, int i, hipStream_t s) {
        auto& t = t_wf_streams;
        const int par = i & 1, prev = par ^ 1;
        for (int k = 0; k < WF_NREG; ++k) {
            hipStream_t st = k == 0 ? s : t.st[k];
            if (i > 0) {
                if (k > 0) CTTS_CHECK_HIP(hipStreamWaitEvent(st, t.ev[k - 1][prev], 0));
                if (k + 1 < WF_NREG) CTTS_CHECK_HIP(hipStreamWaitEvent(st, t.ev[k + 1][prev], 0));
            }
            const int rc = launch_gemm_f32(GEMM_EPI_GATE_RS, shifted(a, tile0[k], tile0[k + 1] - tile0[k], L), st);
            if (rc) return rc;
            CTTS_CHECK_HIP(hipEventRecord(t.ev[k][par], st));
        }
        last_par = par;
        return CTTS_OK;
    }
    int end_row(hipStream_t s) {                                        // the row's tail kernel reads every column
        auto& t = t_wf_streams;
        for (int k = 1; k < WF_NREG; ++k) CTTS_CHECK_HIP(hipStreamWaitEvent(s, t.ev[k][last_par], 0));
        return CTTS_OK;
    }
    int last_par = 0;
};


// Row-queue calls of this thread whose status word has not been looked at yet: (workspace, event recorded behind the call's
// last kernel).  A SET, not the last one: a thread that alternates two workspaces (two batch sizes, two models) must find
// the first one's abort at its next call on THAT workspace; and the event, not the new call's stream, is what is waited for -
// the earlier call may have run on another stream.
struct WfPending {
    static constexpr int N = 16;
    const void* ws[N] = {};
    unsigned int* st[N] = {};                               // the workspace's status word
    hipEvent_t ev[N] = {};
    unsigned long long age[N] = {};
    unsigned long long clock = 0;
    int find(const void* w) const {
        for (int i = 0; i < N; ++i)
            if (ws[i] == w) return i;
        return -1;
    }
};
thread_local WfPending t_wf_pending;

// Reads the workspace's sticky status word and clears it: CTTS_E_ABORT (+ ctts_last_error) if a row-queue call on this
// workspace gave up its bounded wait since the last report.  `done`: the event behind that call (waited for), or NULL (then
// stream `s` is synchronised: the caller says the call ran there).
int wf_report_abort(unsigned int* status, hipEvent_t done, hipStream_t s) {
    unsigned int host = 0;
    if (done) CTTS_CHECK_HIP(hipEventSynchronize(done));
    CTTS_CHECK_HIP(hipMemcpyAsync(&host, status, sizeof host, hipMemcpyDeviceToHost, s));
    CTTS_CHECK_HIP(hipStreamSynchronize(s));
    if (host != WF_ABORT_MAGIC) return CTTS_OK;
    CTTS_CHECK_HIP(hipMemsetAsync(status, 0, sizeof host, s));
    set_error("waveflow: the row queue of an earlier call on this workspace aborted (a bounded wait between workgroups "
              "expired); that call's audio is NaN.  Repeat it - with CTTS_WF_NO_ROW_QUEUE=1 (one launch per layer) if it "
              "happens again");
    return CTTS_E_ABORT;
}

// the pending entry of `workspace`, if any, is checked and removed
int wf_check_pending(const void* workspace, unsigned int* status, hipStream_t s) {
    auto& pd = t_wf_pending;
    const int i = pd.find(workspace);
    if (i < 0) return CTTS_OK;
    pd.ws[i] = nullptr;
    return wf_report_abort(status, pd.ev[i], s);
}

// remember that `workspace` ran the row queue on `s`.  A full table gives up its oldest entry: that workspace's status is read
// now (its call is long done) and an abort found there is reported by THIS call - late and against the wrong call, but not lost.
int wf_note_pending(const void* workspace, unsigned int* status, hipStream_t s) {
    auto& pd = t_wf_pending;
    int i = pd.find(workspace);
    int rc = CTTS_OK;
    if (i < 0) i = pd.find(nullptr);
    if (i < 0) {
        i = 0;
        for (int k = 1; k < WfPending::N; ++k)
            if (pd.age[k] < pd.age[i]) i = k;
        pd.ws[i] = nullptr;
        rc = wf_report_abort(pd.st[i], pd.ev[i], s);
    }
    if (!pd.ev[i]) CTTS_CHECK_HIP(hipEventCreateWithFlags(&pd.ev[i], hipEventDisableTiming));
    CTTS_CHECK_HIP(hipEventRecord(pd.ev[i], s));
    pd.ws[i] = workspace;
    pd.st[i] = status;
    pd.age[i] = ++pd.clock;
    return rc;
}

int wf_inverse(const ctts_waveflow_config* cfg, const void* packed, const float* z, const float* cond, int cond_ld,
               int cond_pad, float* audio, int batch, int samples, int frames, void* workspace, size_t workspace_bytes,
               void* stream) {
    WfPlan p; WfGeom g; WfWs w;
    int rc = make_wf_plan(cfg, p); if (rc) return rc;
    rc = make_wf_geom(p, samples, g); if (rc) return rc;
    CTTS_CHECK_ARG(packed && z && cond && audio && workspace && batch >= 1 && frames >= 1, "waveflow inverse: bad argument");
    wf_carve(p, g, batch, static_cast<float*>(workspace), w);
    if (w.total * sizeof(float) > workspace_bytes) {
        set_error("waveflow inverse: workspace %zu bytes < required %zu", workspace_bytes, w.total * sizeof(float));
        return CTTS_E_WORKSPACE;
    }
    hipStream_t s = as_stream(stream);
    // the previous call of this thread on this workspace used the row queue: if it aborted, say so now instead of computing
    // on top of it (one stream synchronisation + 4 bytes; the caller sees CTTS_E_ABORT once, the call after that runs)
    if ((rc = wf_check_pending(workspace, w.q.status, s))) return rc;
    const float* blob = static_cast<const float*>(packed);
    const int G = p.c.n_group, C = p.C, L = g.L, kh = p.c.kernel_size_h, kw = p.c.kernel_size_w;
    const int gkh = p.sep ? 1 : kh, gkw = p.sep ? 1 : kw;
    const long long cstride = (long long)C * g.ld;
    const dim3 lgrid((g.Lr + 255) / 256, batch);

    hipLaunchKernelGGL(wf_squeeze_kernel, lgrid, dim3(256), 0, s, z, w.rows, G, L, g.Lr);
    CTTS_CHECK_LAUNCH("wf_squeeze");
    if (!p.precond) {
        hipLaunchKernelGGL(wf_interp_kernel, dim3((L + 255) / 256, p.c.n_mel_channels, batch), dim3(256), 0, s, cond,
                           w.mel_up, p.c.n_mel_channels, p.kmel, frames, L, g.ld, g.pad);
        CTTS_CHECK_LAUNCH("wf_interp");
    }

    // Logical row g of the current flow lives in physical row phys[g] of w.rows.  With early outputs only the LAST
    // n_rem rows of the squeezed latent are active at first (the last split, ax:311-313); the earlier chunks re-join in
    // front, one per n_early_every flows (ax:340-341).
    int phys[64], perm[64], tmp[64];
    int Ga = p.n_rem[p.c.n_flows - 1];
    for (int i = 0; i < Ga; ++i) phys[i] = G - Ga + i;
    const int NS = p.ring;
    auto X = [&](int layer, int slot) { return w.X + ((size_t)layer * NS + slot) * w.xslot; };
    const bool no_fuse = tuning().wf_no_fuse;
    const bool fuse = p.fused() && !no_fuse;
    // Region split of the fused layer (round 4): every layer launch used to wait for the slowest of the previous layer's 456
    // workgroups although tile t of layer i + 1 needs only tiles t - 1, t, t + 1 of layer i (dilation <= 128 < one 256-column
    // tile).  The columns are cut into three regions A | M (one tile) | B launched on three streams: A(i+1) and B(i+1) wait
    // for M(i) only, M(i+1) waits for A(i) and B(i), so the two halves drift by up to a layer and each half's tail overlaps
    // the other's next layer.  Same kernel, same tiles: bit-identical to the single launch (CTTS_WF_NO_REGION_SPLIT).
    WfRegionSplit rsplit;
    // (only where the launch is the 128 x 256 shape: below 256 blocks - batch <= 4 at 900 frames - a layer lasts ~35-100 us and
    // three launches + six event operations per layer make the call host-bound: measured 38.7 -> 78 ms at batch 1)
    rsplit.on = fuse && !tuning().wf_no_region_split && g.ntiles >= 8 && (long long)g.ntiles * batch >= WF_SPLIT_MIN_BLOCKS && !tuning().f32_force_small;
    rsplit.ntiles = g.ntiles; rsplit.L = L;
    // Row queue (round 4, gemm_f32_small.hip wf_row_persistent_kernel): the fused layers of a row as ONE launch whose workgroups take
    // (layer, tile) items in order and wait for the three neighbouring tiles of the previous layer only.  Same tile body as the
    // per-layer 128 x 128 shape: bit-identical to it (and to the 128 x 256 shape).  Descriptors: the GemmArgs of a whole flow are
    // built on the host, copied once per flow, and a row's launch points at its slice.
    const int q_items = wf_row_tiles(L, 0) * batch;
    const int q_min = tuning().wf_row_queue_min >= 0 ? tuning().wf_row_queue_min : WF_ROW_QUEUE_MIN_ITEMS;
    // (upper bound: from ~1250 items per layer on - batch 10 at 900 frames: 204.0 ms queued against 207.6, batch 12: 245.1 against
    // 243.4, batch 16: 327 against 319 - the per-layer launches on the 128 x 256 shape with the region split are ahead again: more
    // work per workgroup, and a layer's tail is small against it)
    // (the queue's tile bodies are the fp32-MFMA DMA-staged ones: under the split-bf16 modes, CTTS_F32_NO_GLDS or CTTS_F32_NO_SMALL
    // the layers are launched one by one as before)
    bool queue_on = fuse && !p.sep && !tuning().wf_no_row_queue && q_items >= q_min &&
                    (q_items < WF_ROW_QUEUE_MAX_ITEMS || tuning().wf_row_queue_min >= 0) &&
                    !gemm_mode_is_split(p.c.f32_gemm_mode) && !tuning().f32_no_glds && !tuning().f32_no_small;
    // Which tile body (measured, profiles/r4_12): below 400 items of 128 columns per layer the split-K body (items of 128 x 64, half
    // the serial chain each: 97 -> 77 ms at batch 3 x 900 frames, 95 -> 66 at 8 x 300); the 128 x 128 body from there on
    int q_body = 0;
    if (!tuning().f32_no_splitk) q_body = q_items < WF_ROW_QUEUE_SPLITK_BELOW ? 1 : 0;
    if (tuning().wf_queue_debug & 16) q_body = 0;
    if (tuning().wf_queue_debug & 32) q_body = 1;
    for (int i = 0; i < p.c.n_layers && queue_on; ++i)      // a fresh segment may reach one 128-column tile to either side
        queue_on = (kw / 2) * (p.c.dilation_w[i] > 0 ? p.c.dilation_w[i] : 1 << i) <= 128;
    size_t q_host_off = 0;                                 // bytes of the pinned host buffer the flows queued so far occupy
    // whole-flow form: every row of a flow in ONE launch (tail stages between the rows); CTTS_WF_QUEUE_DEBUG=128: one launch per row
    const bool q_flow = queue_on && !(tuning().wf_queue_debug & 128);
    unsigned int q_launch = 0;
    if (queue_on) {
        if ((rc = t_wf_desc.begin((size_t)p.c.n_flows * G * (p.c.n_layers * sizeof(GemmArgs) + sizeof(WfTailDesc))))) return rc;
        CTTS_CHECK_HIP(hipMemsetAsync(w.q.abort_word, 0, w.q.control_bytes, s));
        if (tuning().wf_inject_abort) CTTS_CHECK_HIP(hipMemsetAsync(w.q.abort_word, 1, sizeof(unsigned int), s));   // (tests)
    }
    const bool sep_fuse = p.sep_fused() && !no_fuse;
    // un-mix of flow k on the active rows: PermuteHeight composes into the map, the 1x1 conv is a pass over the rows
    auto unmix = [&](int k) -> int {
        if (p.c.mixing == CTTS_MIX_PERMUTE) {
            wf_permutation(k, Ga, perm);
            for (int i = 0; i < Ga; ++i) tmp[i] = phys[perm[i]];
            for (int i = 0; i < Ga; ++i) phys[i] = tmp[i];
            return CTTS_OK;
        }
        RowMap m;
        for (int i = 0; i < 64; ++i) m.phys[i] = i < Ga ? phys[i] : 0;
        hipLaunchKernelGGL(wf_mix_kernel, lgrid, dim3(256), 0, s, w.rows, blob + p.fl[k].winv, m, Ga, G, g.Lr);
        CTTS_CHECK_LAUNCH("wf_mix");
        return CTTS_OK;
    };

    // the fused dense layer (no depthwise stage) of flow f, row r, layer i - the ONE place its launch arguments are made, for the
    // per-layer launches, the region split and the row queue alike
    // mark_fresh: 0 no marks (per-layer launches), 1 what the previous layer of the SAME row wrote (row queue, one launch per
    // row), 2 every X segment (whole-flow queue: earlier rows were written inside the same launch too)
    auto fused_args = [&](const WfPlan::Flow& f, int r, int i, int mark_fresh) {
        const int dw = p.c.dilation_w[i] > 0 ? p.c.dilation_w[i] : 1 << i;
        const bool merge = p.c.merge_res_skip != 0;
        const int si = merge ? 0 : i, slot = r % NS;
        const int dh = p.dh(i);
        const int a_min = std::max(0, kh - 1 - r / dh);
        GemmArgs a{};
        a.gate = p.c.gated_unit;
        a.gemm_mode = p.c.f32_gemm_mode;
        a.bm = WF_BM;
        a.ld = g.ld; a.pad = g.pad; a.L = L; a.ntiles = g.ntiles; a.batch = batch;
        a.dst_ld = g.ld; a.dst_pad = g.pad;
        a.A = blob + f.in_A[i]; a.bias = blob + f.in_b[i];
        a.a_nch_alloc = p.nch_in;
        a.MB = p.in_mb(); a.M = 2 * C; a.pairC = C;
        int ns = 0;
        a.a_ch_off = a_min * kw * p.nch_c;
        for (int ah = a_min; ah < gkh; ++ah) {
            const int src_row = r - (kh - 1 - ah) * dh;
            // the current row's slot of layer i > 0 is what layer i - 1 of this very row wrote
            const int fresh = mark_fresh == 2 || (mark_fresh == 1 && src_row == r && si > 0) ? 1 : 0;
            for (int j = 0; j < gkw; ++j)
                a.seg[ns++] = {X(si, src_row % NS), cstride, p.nch_c, (j - kw / 2) * dw, 0, fresh};
        }
        a.nch_total = (kh - a_min) * kw * p.nch_c;
        if (p.precond) {
            a.addend = w.cond_up + (size_t)i * w.cond_slot;
            a.addend_bstride = (long long)2 * C * g.ld;
        } else {
            a.seg[ns++] = {w.mel_up, (long long)p.kmel * g.ld, p.kmel / GEMM_KC, 0, 0, 0};
            a.nch_total += p.kmel / GEMM_KC;
        }
        a.nseg = ns;
        const bool last = i == p.c.n_layers - 1 || merge;       // all rows of the res/skip GEMM are skip rows
        // x_{i+1}[row r] = x_i[row r] + res -> layer i+1's ring slot;  skip rows (+)= into w.out
        a.rs_wT = blob + f.rs_T[i]; a.rs_bias = blob + f.rs_Tb[i]; a.rs_rows = p.rs_rows(i);
        a.dst0 = last ? w.out : X(i + 1, slot); a.dst0_bstride = cstride; a.acc0 = 1;
        a.src0 = X(si, slot); a.src0_bstride = cstride;
        a.dst1 = w.out; a.dst1_bstride = cstride; a.acc1 = i > 0 ? 1 : 0;
        a.split = last ? 0 : C;
        return a;
    };

    // one row of the recurrence as one launch per layer (per-layer GEMM shapes, region split, separable / unfused forms)
    auto per_layer_row = [&](const WfPlan::Flow& f, int r, int slot) -> int {
        int rc = CTTS_OK;
        if (rsplit.on && (rc = rsplit.begin_row(s))) return rc;
        for (int i = 0; i < p.c.n_layers; ++i) {
            const int dw = p.c.dilation_w[i] > 0 ? p.c.dilation_w[i] : 1 << i;
            // merge_res_skip (glow_ax.py:612-626): no residual into `audio`, so every layer's queue holds the
            // `start` outputs - layer i reads the ring of layer 0 and nothing is written to ring i+1
            const bool merge = p.c.merge_res_skip != 0;
            const int si = merge ? 0 : i;
            const int dh = p.dh(i);                                   // height tap ah reads row r - (kh-1-ah)*dh
            const int a_min = std::max(0, kh - 1 - r / dh);           // earlier rows do not exist: skip those taps
            GemmArgs a{};
            a.gate = p.c.gated_unit;
            a.gemm_mode = p.c.f32_gemm_mode;
            a.bm = WF_BM;
            a.ld = g.ld; a.pad = g.pad; a.L = L; a.ntiles = g.ntiles; a.batch = batch;
            a.dst_ld = g.ld; a.dst_pad = g.pad;
            a.A = blob + f.in_A[i]; a.bias = blob + f.in_b[i];
            a.a_nch_alloc = p.nch_in;
            a.MB = p.in_mb(); a.M = 2 * C; a.pairC = C;
            int ns = 0;
            if (p.sep) {
                WfSlots xs{};
                for (int ah = a_min; ah < kh; ++ah) xs.p[ah] = X(si, (r - (kh - 1 - ah) * dh) % NS);
                const dim3 vgrid(((L + 3) / 4 + 255) / 256, C, batch);
                const bool vec = g.pad % 4 == 0 && g.ld % 4 == 0 && (dw <= 2 || dw % 4 == 0);
                if (vec && kw == 7) launch_depthwise_vec<7>(vgrid, s, xs, blob + f.dw_w[i], blob + f.dw_b[i], w.dwout, C, kh, dw, a_min, L, g.ld, g.pad);
                else if (vec && kw == 5) launch_depthwise_vec<5>(vgrid, s, xs, blob + f.dw_w[i], blob + f.dw_b[i], w.dwout, C, kh, dw, a_min, L, g.ld, g.pad);
                else if (vec && kw == 3) launch_depthwise_vec<3>(vgrid, s, xs, blob + f.dw_w[i], blob + f.dw_b[i], w.dwout, C, kh, dw, a_min, L, g.ld, g.pad);
                else
                    hipLaunchKernelGGL(wf_depthwise_kernel, dim3((L + 255) / 256, C, batch), dim3(256), 0, s, xs,
                                       blob + f.dw_w[i], blob + f.dw_b[i], w.dwout, C, kh, kw, dw, a_min, L, g.ld, g.pad);
                CTTS_CHECK_LAUNCH("wf_depthwise");
                if (sep_fuse) {
                    WfSepArgs q{};
                    q.dwout = w.dwout;
                    q.A1 = blob + f.sA1[i]; q.b1 = blob + f.sb1[i]; q.A2 = blob + f.sA2[i]; q.b2 = blob + f.sb2[i];
                    q.cond = w.cond_up + (size_t)i * w.cond_slot;
                    q.xin = X(si, slot);
                    q.xout = (i == p.c.n_layers - 1 || merge) ? nullptr : X(i + 1, slot);
                    q.gate = p.c.gated_unit;
                    q.split_bf16 = gemm_split_level(p.c.f32_gemm_mode) == 3 ? 1 : 0;   // (x6: the fused separable layer stays on fp32 MFMA)
                    q.out = w.out; q.acc_out = i > 0 ? 1 : 0; q.rs_rows = p.rs_rows(i);
                    q.L = L; q.ld = g.ld; q.pad = g.pad; q.ntiles = (L + 63) / 64;
                    if ((rc = launch_wf_sep_layer(q, batch, s))) return rc;
                    note_gemm_loop(q.split_bf16 ? 3 : 0);
                    continue;
                }
                a.a_ch_off = 0;
                a.seg[ns++] = {w.dwout, cstride, p.nch_c, 0, 0, 0};
                a.nch_total = p.nch_c;
            } else {
                a.a_ch_off = a_min * kw * p.nch_c;
                for (int ah = a_min; ah < gkh; ++ah) {
                    const int src_row = r - (kh - 1 - ah) * dh;
                    for (int j = 0; j < gkw; ++j)
                        a.seg[ns++] = {X(si, src_row % NS), cstride, p.nch_c, (j - kw / 2) * dw, 0, 0};
                }
                a.nch_total = (kh - a_min) * kw * p.nch_c;
            }
            if (p.precond) {
                a.addend = w.cond_up + (size_t)i * w.cond_slot;
                a.addend_bstride = (long long)2 * C * g.ld;
            } else {
                a.seg[ns++] = {w.mel_up, (long long)p.kmel * g.ld, p.kmel / GEMM_KC, 0, 0, 0};
                a.nch_total += p.kmel / GEMM_KC;
            }
            a.nseg = ns;
            const bool last = i == p.c.n_layers - 1 || merge;       // all rows of the res/skip GEMM are skip rows
            if (fuse) {
                if (!p.sep) a = fused_args(f, r, i, 0);
                else {
                    // x_{i+1}[row r] = x_i[row r] + res -> layer i+1's ring slot;  skip rows (+)= into w.out
                    a.rs_wT = blob + f.rs_T[i]; a.rs_bias = blob + f.rs_Tb[i]; a.rs_rows = p.rs_rows(i);
                    a.dst0 = last ? w.out : X(i + 1, slot); a.dst0_bstride = cstride; a.acc0 = 1;
                    a.src0 = X(si, slot); a.src0_bstride = cstride;
                    a.dst1 = w.out; a.dst1_bstride = cstride; a.acc1 = i > 0 ? 1 : 0;
                    a.split = last ? 0 : C;
                }
                if (rsplit.on) {
                    if ((rc = rsplit.layer(a, i, s))) return rc;
                    continue;
                }
                if ((rc = launch_gemm_f32(GEMM_EPI_GATE_RS, a, s))) return rc;
                continue;
            }
            a.dst0 = w.act; a.dst0_bstride = cstride;
            if ((rc = launch_gemm_f32(GEMM_EPI_GATE, a, s))) return rc;

            GemmArgs q{};
            q.gemm_mode = p.c.f32_gemm_mode;
            q.bm = WF_BM;
            q.ld = g.ld; q.pad = g.pad; q.L = L; q.ntiles = g.ntiles; q.batch = batch;
            q.dst_ld = g.ld; q.dst_pad = g.pad;
            q.A = blob + f.rs_A[i]; q.bias = blob + f.rs_b[i];
            q.nseg = 1; q.nch_total = p.nch_c; q.MB = p.rs_mb(i); q.M = p.rs_rows(i);
            q.seg[0] = {w.act, cstride, p.nch_c, 0, 0, 0};
            // x_{i+1}[row r] = x_i[row r] + res : written into layer i+1's ring slot (its queue entry)
            q.dst0 = last ? w.out : X(i + 1, slot); q.dst0_bstride = cstride; q.acc0 = 1;
            q.src0 = X(si, slot); q.src0_bstride = cstride;
            q.dst1 = w.out; q.dst1_bstride = cstride; q.acc1 = i > 0 ? 1 : 0;
            q.split = last ? 0 : C;
            if ((rc = launch_gemm_f32(GEMM_EPI_SPLIT, q, s))) return rc;
        }
        if (rsplit.on && (rc = rsplit.end_row(s))) return rc;
        return CTTS_OK;
    };

    for (int k = p.c.n_flows - 1; k >= 0; --k) {
        const auto& f = p.fl[k];
        if (!p.c.mix_first && (rc = unmix(k))) return rc;                      // ax:324-325
        int q_max_nseg = 0;
        if (queue_on) {                                                        // this flow's descriptors, one copy
            GemmArgs* host = reinterpret_cast<GemmArgs*>(t_wf_desc.host + q_host_off);
            size_t n = 0;
            for (int r = 0; r < Ga - 1; ++r)
                for (int i = 0; i < p.c.n_layers; ++i) {
                    GemmArgs& a = host[n++] = fused_args(f, r, i, (tuning().wf_queue_debug & 4) ? 0 : q_flow ? 2 : 1);
                    gemm_apply_defaults(a);                     // (the tile body is launched without launch_gemm_f32:
                    if ((rc = gemm_check_args(GEMM_EPI_GATE_RS, a))) return rc;   //  its defaults and checks are applied here)
                    q_max_nseg = std::max(q_max_nseg, a.nseg);
                    CTTS_CHECK_ARG(wf_row_persistent_supported(a), "waveflow row queue: layer %d not supported by the tile body", i);
                }
            CTTS_CHECK_HIP(hipMemcpyAsync(w.q.layers, host, n * sizeof(GemmArgs), hipMemcpyHostToDevice, s));
            q_host_off += n * sizeof(GemmArgs);
            if (q_flow) {                                                      // the tail stage behind every row
                WfTailDesc* th = reinterpret_cast<WfTailDesc*>(t_wf_desc.host + q_host_off);
                for (int r = 0; r < Ga - 1; ++r) {
                    WfTailDesc d{};
                    d.out = w.out; d.out_bstride = cstride;
                    d.rows = w.rows; d.Wend = blob + f.end_w; d.bend = blob + f.end_b;
                    d.ws = blob + f.start_w; d.bs = blob + f.start_b;
                    d.x0 = r + 1 < Ga - 1 ? X(0, (r + 1) % NS) : nullptr; d.x0_bstride = cstride;
                    d.C = C; d.G = G; d.row = phys[r + 1]; d.L = L; d.Lr = g.Lr; d.ld = g.ld; d.pad = g.pad;
                    th[r] = d;
                }
                CTTS_CHECK_HIP(hipMemcpyAsync(w.q.tails, th, (size_t)(Ga - 1) * sizeof(WfTailDesc), hipMemcpyHostToDevice, s));
                q_host_off += (size_t)(Ga - 1) * sizeof(WfTailDesc);
            }
        }
        if (p.precond) {   // this flow's conditioning, upsampled once for all rows and layers
            const float* fr = cond + (size_t)k * batch * 2 * C * p.c.n_layers * cond_ld;
            if (g.pad % 4 == 0 && g.ld % 4 == 0 && w.cond_slot % 4 == 0 && !tuning().wf_no_vec_interp)
                hipLaunchKernelGGL(wf_interp_cond_vec_kernel, dim3(((L + 3) / 4 + 255) / 256, 2 * C * p.c.n_layers, batch), dim3(256), 0, s,
                                   fr, w.cond_up, 2 * C, p.c.n_layers, frames, cond_ld, cond_pad, L, g.ld, g.pad, w.cond_slot);
            else
                hipLaunchKernelGGL(wf_interp_cond_kernel, dim3((L + 255) / 256, 2 * C * p.c.n_layers, batch), dim3(256), 0, s,
                                   fr, w.cond_up, 2 * C, p.c.n_layers, frames, cond_ld, cond_pad, L, g.ld, g.pad, w.cond_slot);
            CTTS_CHECK_LAUNCH("wf_interp_cond");
        }
        if (q_flow) {                                       // start of row 0, then every row and tail of the flow in one launch
            hipLaunchKernelGGL(wf_start_kernel, dim3(((L + 3) / 4 + 255) / 256, (C + 15) / 16, batch), dim3(256), 0, s,
                               w.rows, blob + f.start_w, blob + f.start_b, X(0, 0), C, G, phys[0], L, g.Lr, g.ld, g.pad);
            CTTS_CHECK_LAUNCH("wf_start");
            const unsigned int li = q_launch++;
            if ((rc = launch_wf_row_persistent(w.q.layers, w.q.tails, Ga - 1, p.c.n_layers, q_max_nseg, L, batch, q_body,
                                               w.q.counters + li, w.q.flags, w.q.abort_word, li + 1, s)))
                return rc;
        }
        for (int r = 0; r < Ga - 1 && !q_flow; ++r) {
            const int slot = r % NS;
            hipLaunchKernelGGL(wf_start_kernel, dim3(((L + 3) / 4 + 255) / 256, (C + 15) / 16, batch), dim3(256), 0, s,
                               w.rows, blob + f.start_w, blob + f.start_b, X(0, slot), C, G, phys[r], L, g.Lr, g.ld, g.pad);
            CTTS_CHECK_LAUNCH("wf_start");
            if (queue_on) {
                const unsigned int li = q_launch++;
                if ((rc = launch_wf_row_persistent(w.q.layers + (size_t)r * p.c.n_layers, nullptr, 1, p.c.n_layers, q_max_nseg, L, batch, q_body,
                                                   w.q.counters + li, w.q.flags, w.q.abort_word, li + 1, s)))
                    return rc;
            } else if ((rc = per_layer_row(f, r, slot))) {
                return rc;
            }
            hipLaunchKernelGGL(wf_tail_kernel, dim3((g.Lr + 255) / 256, batch), dim3(256), 0, s, w.out, w.rows,
                               blob + f.end_w, blob + f.end_b, C, G, phys[r + 1], L, g.Lr, g.ld, g.pad);
            CTTS_CHECK_LAUNCH("wf_tail");
        }
        const size_t n4 = (size_t)batch * G * g.Lr / 4;
        hipLaunchKernelGGL(wf_nan_to_zero_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, w.rows, n4);
        CTTS_CHECK_LAUNCH("wf_nan_to_zero");
        if (p.c.mix_first && (rc = unmix(k))) return rc;                       // ax:337-338
        if (k > 0 && p.n_rem[k - 1] > Ga) {                                    // ax:340-341: the early chunk re-joins in front
            const int grow = p.n_rem[k - 1] - Ga;
            for (int i = Ga - 1; i >= 0; --i) phys[i + grow] = phys[i];
            for (int i = 0; i < grow; ++i) phys[i] = G - Ga - grow + i;
            Ga += grow;
        }
    }
    RowMap map;
    for (int i = 0; i < 64; ++i) map.phys[i] = i < G ? phys[i] : 0;
    hipLaunchKernelGGL(wf_unsqueeze_kernel, dim3((L + 255) / 256, batch), dim3(256), 0, s, w.rows, audio, G, L, g.Lr, map);
    CTTS_CHECK_LAUNCH("wf_unsqueeze");
    if (queue_on && (rc = t_wf_desc.end(s))) return rc;
    if (queue_on) {      // a bounded wait of the row queue expired (never, by construction): the audio is NaN, not plausible noise
        hipLaunchKernelGGL(wf_abort_poison_kernel, dim3(batch), dim3(256), 0, s, w.q.abort_word, audio, (long long)G * L,
                           w.q.status);
        CTTS_CHECK_LAUNCH("wf_abort_poison");
        if ((rc = wf_note_pending(workspace, w.q.status, s))) return rc;   // this thread's next call on this workspace looks at the status first
    }
    return CTTS_OK;
}

}  // namespace

extern "C" {

size_t ctts_waveflow_packed_bytes(const ctts_waveflow_config* cfg) {
    WfPlan p;
    if (make_wf_plan(cfg, p)) return 0;
    return p.total * sizeof(float);
}

int ctts_waveflow_pack_flow(const ctts_waveflow_config* cfg, int32_t k, const ctts_waveflow_flow_weights* w,
                            void* packed, void* stream) {
    WfPlan p;
    int rc = make_wf_plan(cfg, p); if (rc) return rc;
    CTTS_CHECK_ARG(k >= 0 && k < p.c.n_flows && w && packed, "waveflow pack_flow: bad argument");
    CTTS_CHECK_ARG(w->start_w && w->start_b && w->in_w && w->in_b && w->rs_w && w->rs_b && w->end_w && w->end_b,
                   "waveflow pack_flow: NULL weight pointer");
    CTTS_CHECK_ARG(p.precond || (w->cond_w && w->cond_b), "waveflow pack_flow: NULL cond layer");
    CTTS_CHECK_ARG(!p.sep || (w->dw_w && w->dw_b), "waveflow pack_flow: NULL depthwise weights");
    hipStream_t s = as_stream(stream);
    float* blob = static_cast<float*>(packed);
    const auto& f = p.fl[k];
    const int C = p.C, kh = p.c.kernel_size_h, kw = p.c.kernel_size_w, nm = p.c.n_mel_channels;
    const int gkh = p.sep ? 1 : kh, gkw = p.sep ? 1 : kw;      // taps of the GEMM stage
    CTTS_CHECK_HIP(hipMemcpyAsync(blob + f.start_w, w->start_w, C * sizeof(float), hipMemcpyDeviceToDevice, s));
    CTTS_CHECK_HIP(hipMemcpyAsync(blob + f.start_b, w->start_b, C * sizeof(float), hipMemcpyDeviceToDevice, s));
    CTTS_CHECK_HIP(hipMemcpyAsync(blob + f.end_w, w->end_w, 2 * C * sizeof(float), hipMemcpyDeviceToDevice, s));
    CTTS_CHECK_HIP(hipMemcpyAsync(blob + f.end_b, w->end_b, 2 * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (p.c.mixing == CTTS_MIX_CONV1X1) {
        CTTS_CHECK_ARG(w->w_inverse != nullptr, "waveflow pack_flow: 1x1-conv mixing needs w_inverse (flow %d)", k);
        CTTS_CHECK_HIP(hipMemcpyAsync(blob + f.winv, w->w_inverse, (size_t)p.n_rem[k] * p.n_rem[k] * sizeof(float),
                                      hipMemcpyDeviceToDevice, s));
    }
    for (int i = 0; i < p.c.n_layers; ++i) {
        CTTS_CHECK_ARG(w->in_w[i] && w->in_b[i] && w->rs_w[i] && w->rs_b[i], "waveflow pack_flow: NULL layer %d", i);
        CTTS_CHECK_HIP(hipMemsetAsync(blob + f.in_A[i], 0, (size_t)p.in_mb() * p.nch_in * GEMM_KC * WF_BM * sizeof(float), s));
        // K = [height tap a][width tap j][channel] then the cond rows;  in_w[i] is [2C][C][gkh][gkw]
        for (int a = 0; a < gkh; ++a)
            for (int j = 0; j < gkw; ++j)
                if ((rc = launch_pack_a(blob + f.in_A[i], w->in_w[i] + a * gkw + j, WF_BM, p.in_mb(), p.nch_in,
                                        (a * gkw + j) * C, C, GEMM_EPI_GATE, C, 2 * C, 0, (long long)C * gkh * gkw,
                                        gkh * gkw, s))) return rc;
        if (!p.precond)
            if ((rc = launch_pack_a(blob + f.in_A[i], w->cond_w, WF_BM, p.in_mb(), p.nch_in, gkh * gkw * C, nm,
                                    GEMM_EPI_GATE, C, 2 * C, (long long)2 * C * i, nm, 1, s))) return rc;
        if ((rc = launch_pack_bias(blob + f.in_b[i], WF_BM, p.in_mb(), w->in_b[i], 0, p.precond ? nullptr : w->cond_b,
                                   (long long)2 * C * i, GEMM_EPI_GATE, C, 2 * C, s))) return rc;
        if (p.sep) {
            CTTS_CHECK_ARG(w->dw_w[i] && w->dw_b[i], "waveflow pack_flow: NULL depthwise layer %d", i);
            CTTS_CHECK_HIP(hipMemcpyAsync(blob + f.dw_w[i], w->dw_w[i], (size_t)C * kh * kw * sizeof(float),
                                          hipMemcpyDeviceToDevice, s));
            CTTS_CHECK_HIP(hipMemcpyAsync(blob + f.dw_b[i], w->dw_b[i], C * sizeof(float), hipMemcpyDeviceToDevice, s));
        }
        const int rows = p.rs_rows(i);
        if ((rc = launch_pack_a(blob + f.rs_A[i], w->rs_w[i], WF_BM, p.rs_mb(i), p.nch_c, 0, C, GEMM_EPI_SPLIT, C, rows,
                                0, C, 1, s))) return rc;
        if ((rc = launch_pack_bias(blob + f.rs_b[i], WF_BM, p.rs_mb(i), w->rs_b[i], 0, nullptr, 0, GEMM_EPI_SPLIT, C,
                                   rows, s))) return rc;
        if (p.fused()) {
            hipLaunchKernelGGL(wf_pack_rs_t_kernel, dim3(33), dim3(256), 0, s, w->rs_w[i], w->rs_b[i], blob + f.rs_T[i],
                               blob + f.rs_Tb[i], rows);
            CTTS_CHECK_LAUNCH("wf_pack_rs_t");
        }
        if (p.sep_fused())
            if ((rc = launch_wf_sep_pack(w->in_w[i], w->in_b[i], w->rs_w[i], w->rs_b[i], blob + f.sA1[i], blob + f.sb1[i],
                                         blob + f.sA2[i], blob + f.sb2[i], rows, s))) return rc;
    }
    return CTTS_OK;
}

size_t ctts_waveflow_workspace_bytes(const ctts_waveflow_config* cfg, int32_t batch, int32_t samples) {
    WfPlan p; WfGeom g; WfWs w;
    if (make_wf_plan(cfg, p) || make_wf_geom(p, samples, g) || batch < 1) return 0;
    wf_carve(p, g, batch, nullptr, w);
    return w.total * sizeof(float);
}

int ctts_waveflow_inverse_f32(const ctts_waveflow_config* cfg, const void* packed, const float* z, const float* mel,
                              float* audio, int32_t batch, int32_t samples, int32_t frames, void* workspace,
                              size_t workspace_bytes, void* stream) {
    CTTS_CHECK_ARG(cfg && !cfg->cond_precomputed, "waveflow inverse: this model takes per-flow conditioning "
                                                  "(ctts_waveflow_inverse_cond_f32)");
    return wf_inverse(cfg, packed, z, mel, 0, 0, audio, batch, samples, frames, workspace, workspace_bytes, stream);
}

int ctts_waveflow_abort_status(const ctts_waveflow_config* cfg, int32_t batch, int32_t samples, void* workspace,
                               size_t workspace_bytes, void* stream) {
    WfPlan p; WfGeom g; WfWs w;
    int rc = make_wf_plan(cfg, p); if (rc) return rc;
    rc = make_wf_geom(p, samples, g); if (rc) return rc;
    CTTS_CHECK_ARG(workspace && batch >= 1, "waveflow abort_status: bad argument");
    wf_carve(p, g, batch, static_cast<float*>(workspace), w);
    if (w.total * sizeof(float) > workspace_bytes) {
        set_error("waveflow abort_status: workspace %zu bytes < required %zu", workspace_bytes, w.total * sizeof(float));
        return CTTS_E_WORKSPACE;
    }
    {   // the pending entry (if this thread has one) carries the event of the call in question; without one the caller's
        // stream is what the call ran on
        auto& pd = t_wf_pending;
        const int i = pd.find(workspace);
        hipEvent_t done = nullptr;
        if (i >= 0) { pd.ws[i] = nullptr; done = pd.ev[i]; }
        return wf_report_abort(w.q.status, done, as_stream(stream));
    }
}

int ctts_waveflow_inverse_cond_f32(const ctts_waveflow_config* cfg, const void* packed, const float* z,
                                   const float* cond, int32_t cond_ld, int32_t cond_pad, float* audio, int32_t batch,
                                   int32_t samples, int32_t frames, void* workspace, size_t workspace_bytes,
                                   void* stream) {
    CTTS_CHECK_ARG(cfg && cfg->cond_precomputed, "waveflow inverse_cond: the model folds its cond layer "
                                                 "(ctts_waveflow_inverse_f32)");
    CTTS_CHECK_ARG(cond_pad >= 0 && cond_ld >= cond_pad + frames, "waveflow inverse_cond: cond_ld=%d pad=%d frames=%d",
                   cond_ld, cond_pad, frames);
    return wf_inverse(cfg, packed, z, cond, cond_ld, cond_pad, audio, batch, samples, frames, workspace,
                      workspace_bytes, stream);
}

int ctts_embed_rows_f32(const float* table, const int64_t* ids, float* x, int32_t row0, int32_t embed_dim,
                        int32_t batch, int32_t C, int32_t T, int32_t ld, int32_t pad, void* stream) {
    CTTS_CHECK_ARG(table && ids && x && embed_dim >= 1 && row0 >= 0 && row0 + embed_dim <= C && batch >= 1 && T >= 1 &&
                   ld >= pad + T, "embed_rows: bad argument");
    hipLaunchKernelGGL(embed_rows_kernel, dim3((T + 255) / 256, embed_dim, batch), dim3(256), 0, as_stream(stream), table,
                       ids, x, row0, embed_dim, C, T, ld, pad);
    CTTS_CHECK_LAUNCH("embed_rows");
    return CTTS_OK;
}

int ctts_scale_add_rows_f32(const float* x, const float* alpha_dev, const float* r, float* y, int32_t batch, int32_t C,
                            int32_t T, int32_t ld, int32_t pad, void* stream) {
    CTTS_CHECK_ARG(x && y && batch >= 1 && C >= 1 && T >= 1 && ld >= pad + T, "scale_add_rows: bad argument");
    hipLaunchKernelGGL(scale_add_rows_kernel, dim3((T + 255) / 256, C, batch), dim3(256), 0, as_stream(stream), x,
                       alpha_dev, r, y, C, T, ld, pad);
    CTTS_CHECK_LAUNCH("scale_add_rows");
    return CTTS_OK;
}

int ctts_affine_rows_f32(float* x, int32_t batch, int32_t C, int32_t rows, int32_t T, int32_t ld, int32_t pad, float shift,
                         float scale, void* stream) {
    CTTS_CHECK_ARG(x && batch >= 1 && rows >= 1 && rows <= C && T >= 1 && ld >= pad + T, "affine_rows: bad argument");
    hipLaunchKernelGGL(affine_rows_kernel, dim3((T + 255) / 256, rows, batch), dim3(256), 0, as_stream(stream), x, C, T, ld,
                       pad, shift, scale);
    CTTS_CHECK_LAUNCH("affine_rows");
    return CTTS_OK;
}

int ctts_resample_rows_f32(const float* x, float* y, int32_t batch, int32_t C, int32_t T_in, int32_t ld_in, int32_t pad_in,
                           int32_t T_out, int32_t ld_out, int32_t pad_out, int32_t mode, float scale_factor, void* stream) {
    CTTS_CHECK_ARG(x && y && batch >= 1 && C >= 1 && T_in >= 1 && T_out >= 1 && ld_in >= pad_in + T_in &&
                   ld_out >= pad_out + T_out && mode >= 0 && mode <= 2, "resample_rows: bad argument");
    const float s = scale_factor > 0.f ? 1.0f / scale_factor : (float)T_in / (float)T_out;
    hipLaunchKernelGGL(resample_rows_kernel, dim3((T_out + 255) / 256, C, batch), dim3(256), 0, as_stream(stream), x, y, C,
                       T_in, ld_in, pad_in, T_out, ld_out, pad_out, mode, s);
    CTTS_CHECK_LAUNCH("resample_rows");
    return CTTS_OK;
}

int ctts_interleave_phases_f32(const float* phases, float* y, int32_t batch, int32_t C, int32_t stride, int32_t padding,
                               int32_t T_in, int32_t ld_in, int32_t pad_in, int32_t T_out, int32_t ld_out, int32_t pad_out,
                               void* stream) {
    CTTS_CHECK_ARG(phases && y && batch >= 1 && C >= 1 && stride >= 1 && padding >= 0 && T_in >= 1 && T_out >= 1 &&
                   ld_in >= pad_in + T_in && ld_out >= pad_out + T_out, "interleave_phases: bad argument");
    hipLaunchKernelGGL(interleave_phases_kernel, dim3((T_out + 255) / 256, C, batch), dim3(256), 0, as_stream(stream), phases,
                       y, C, stride, padding, T_in, ld_in, pad_in, T_out, ld_out, pad_out, (size_t)batch * C * ld_in);
    CTTS_CHECK_LAUNCH("interleave_phases");
    return CTTS_OK;
}

int ctts_vol_unscale_f32(float* x, int64_t n, void* stream) {
    CTTS_CHECK_ARG(x && n >= 1, "vol_unscale: bad argument");
    hipLaunchKernelGGL(vol_unscale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), x, (size_t)n);
    CTTS_CHECK_LAUNCH("vol_unscale");
    return CTTS_OK;
}

int ctts_deemphasis_f32(const float* x, float* y, int32_t batch, int32_t T, double p, void* stream) {
    CTTS_CHECK_ARG(x && y && batch >= 1 && T >= 1, "deemphasis: bad argument");
    hipLaunchKernelGGL(deemphasis_kernel, dim3(batch), dim3(256), 0, as_stream(stream), x, y, T, p);
    CTTS_CHECK_LAUNCH("deemphasis");
    return CTTS_OK;
}

}  // extern "C"
